"""ctypes wrapper of oracle/_build/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (mapper_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")
REF_DRIVER = os.path.join(_HERE, "_ref", "ref_driver")
DUMP_MAGIC = 0x48495444554D5031


def build(force=False):
    """Compile the C restatement (gcc only, no GPU)."""
    src = [os.path.join(_HERE, f) for f in ("fmoracle.c", "fmoracle.h")]
    if (not force and os.path.exists(_LIB)
            and all(os.path.getmtime(_LIB) >= os.path.getmtime(s) for s in src)):
        return _LIB
    subprocess.check_call(["make", "-s", "-C", _HERE, "_build/liboracle.so"])
    return _LIB


class OrcHit(C.Structure):
    _fields_ = [("fp", C.c_int64), ("rp", C.c_int64), ("sz", C.c_int64), ("dp", C.c_int64),
                ("score", C.c_int64), ("align", C.c_uint64 * 4)]


class _OrcBatch(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_hits", C.c_int64), ("n_locs", C.c_int64),
                ("status", C.POINTER(C.c_int64)), ("hits", C.POINTER(C.c_int64)),
                ("locs", C.POINTER(C.c_int64)), ("n_ext", C.c_int64), ("seconds", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.orc_index_open.restype = C.c_void_p
        L.orc_index_open.argtypes = [C.c_char_p]
        L.orc_index_close.argtypes = [C.c_void_p]
        L.orc_index_from_arrays.restype = C.c_void_p
        L.orc_index_from_arrays.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                            C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
        for f in ("orc_txt_length", "orc_seq_count", "orc_sar_bits", "orc_wildcard_count"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        L.orc_c_array.restype = C.POINTER(C.c_uint64)
        L.orc_c_array.argtypes = [C.c_void_p]
        L.orc_occ_all.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_occ.restype = C.c_int64
        L.orc_occ.argtypes = [C.c_void_p, C.c_int64, C.c_int]
        L.orc_extend_all.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                     C.POINTER(C.c_int64)]
        L.orc_prefix_all.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_ann_build.restype = C.c_int
        L.orc_ann_build.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    C.POINTER(C.c_int64)]
        L.orc_sar_get.restype = C.c_int64
        L.orc_sar_get.argtypes = [C.c_void_p, C.c_int64]
        L.orc_sar_get_range.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
        L.orc_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_int64, C.POINTER(C.c_uint8)]
        L.orc_pos_to_str.argtypes = [C.c_void_p, C.c_int64, C.c_char_p, C.c_size_t]
        L.orc_blocksearch.restype = C.c_int64
        L.orc_blocksearch.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int,
                                      C.POINTER(C.POINTER(OrcHit)), C.POINTER(C.c_int64)]
        L.orc_seqsearch.restype = C.c_int64
        L.orc_seqsearch.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcHit), C.POINTER(C.c_uint8),
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.POINTER(OrcHit))]
        L.orc_map_batch.restype = C.POINTER(_OrcBatch)
        L.orc_map_batch.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_int64),
                                    C.c_int64, C.c_int, C.c_int]
        L.orc_blocksc_batch.restype = C.POINTER(_OrcBatch)
        L.orc_blocksc_batch.argtypes = L.orc_map_batch.argtypes
        L.orc_blocksc.restype = C.c_int64
        L.orc_blocksc.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int,
                                  C.POINTER(C.POINTER(OrcHit)), C.POINTER(C.c_int64)]
        L.orc_scsearch.restype = C.c_int64
        L.orc_scsearch.argtypes = [C.c_void_p, C.POINTER(OrcHit), C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.POINTER(C.POINTER(OrcHit))]
        L.orc_batch_free.argtypes = [C.POINTER(_OrcBatch)]
        L._libc = C.CDLL(None)
        L._libc.free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


class HitDump:
    """The hit dump shared by ref_driver, the oracle and the HIP path (see ref_driver.c)."""

    def __init__(self, status, hits, locs, n_ext=None, seconds=None):
        self.status = np.asarray(status, dtype=np.int64)
        self.hits = np.asarray(hits, dtype=np.int64).reshape(-1, 11)
        self.locs = np.asarray(locs, dtype=np.int64)
        self.n_ext = n_ext
        self.seconds = seconds

    @classmethod
    def load(cls, path):
        a = np.fromfile(path, dtype=np.int64)
        assert int(a[0]) == DUMP_MAGIC, "bad dump magic"
        nr, nh, nl = int(a[1]), int(a[2]), int(a[3])
        o = 4
        status = a[o:o + nr]; o += nr
        hits = a[o:o + 11 * nh]; o += 11 * nh
        locs = a[o:o + nl]
        return cls(status, hits, locs)

    def save(self, path):
        hdr = np.array([DUMP_MAGIC, len(self.status), len(self.hits), len(self.locs)], dtype=np.int64)
        with open(path, "wb") as f:
            for arr in (hdr, self.status, self.hits.reshape(-1), self.locs):
                np.ascontiguousarray(arr, dtype=np.int64).tofile(f)

    def located(self):
        """Canonical mapped form: int64 [n_locs][3] = (read, text_pos, score), sorted."""
        h = self.hits
        if len(h) == 0:
            return np.zeros((0, 3), dtype=np.int64)
        rep = h[:, 3]
        out = np.empty((int(rep.sum()), 3), dtype=np.int64)
        out[:, 0] = np.repeat(h[:, 0], rep)
        out[:, 2] = np.repeat(h[:, 5], rep)
        # locs are stored hit after hit starting at first_loc
        idx = np.repeat(h[:, 10] - np.concatenate(([0], np.cumsum(rep)[:-1])), rep) + np.arange(len(out))
        out[:, 1] = self.locs[idx]
        order = np.lexsort((out[:, 1], out[:, 0]))
        return out[order]

    def intervals(self):
        """Canonical interval form: hits sorted by (read, fp) -- columns as in the dump minus first_loc."""
        h = self.hits[:, :10]
        order = np.lexsort((h[:, 1], h[:, 0]))
        return h[order]


class Index:
    def __init__(self, base, _handle=None, _keep=None):
        self.base = base
        self._keep = _keep
        self._h = _handle if _handle is not None else lib().orc_index_open(base.encode())
        if not self._h:
            raise IOError("cannot open index %s" % base)
        self.txt_len = lib().orc_txt_length(self._h)
        self.sar_bits = lib().orc_sar_bits(self._h)
        self.seq_cnt = lib().orc_seq_count(self._h)
        self.wil_cnt = lib().orc_wildcard_count(self._h)
        self.c = [int(lib().orc_c_array(self._h)[i]) for i in range(6)]

    @classmethod
    def from_arrays(cls, occ, c, sa, sar_bits, txt_len, wil_cnt, text=None, mark_intv=14):
        """Index over numpy arrays laid out as in the .bwt / .sar (/.txt) files."""
        occ = np.ascontiguousarray(occ, dtype=np.uint64)
        c = np.ascontiguousarray(c, dtype=np.uint64)
        sa = np.ascontiguousarray(sa).view(np.uint64)
        h = lib().orc_index_from_arrays(occ.ctypes.data, len(occ), mark_intv, c.ctypes.data, sa.ctypes.data,
                                        int(sar_bits), len(sa), int(txt_len), int(wil_cnt),
                                        text.ctypes.data if text is not None else None)
        return cls("<arrays>", _handle=h, _keep=(occ, c, sa, text))

    def close(self):
        if self._h:
            lib().orc_index_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def occ_all(self, ptr):
        out = (C.c_int64 * 5)()
        lib().orc_occ_all(self._h, ptr, out)
        return list(out)

    def occ(self, ptr, sym):
        return lib().orc_occ(self._h, ptr, sym)

    def extend_all(self, end, fp, rp, sz):
        out = (C.c_int64 * 15)()
        lib().orc_extend_all(self._h, end, fp, rp, sz, out)
        return [tuple(out[3 * j:3 * j + 3]) for j in range(5)]

    def prefix_all(self, fp, sz):
        out = (C.c_int64 * 10)()
        lib().orc_prefix_all(self._h, fp, sz, out)
        return [tuple(out[2 * j:2 * j + 2]) for j in range(5)]

    def ann_build(self, kmer, tau, threads=1, trail_reuse=False):
        """ann_build (src/index_ann.c:56-257): the annotation table (ann->info, txt_len / 2 bytes) and the work
        figures (k-mers searched, extension steps, writes of the compression outside the table)."""
        table = np.zeros(self.txt_len // 2, dtype=np.uint8)
        nq, ne, dr = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        rc = lib().orc_ann_build(self._h, kmer, tau, threads, int(trail_reuse), table.ctypes.data_as(C.c_void_p), C.byref(nq), C.byref(ne), C.byref(dr))
        if rc != 0:
            return None, {}
        return table, {"queries": nq.value, "n_ext": ne.value, "dropped": dr.value}

    def sar_get(self, pos):
        return lib().orc_sar_get(self._h, pos)

    def sar_get_range(self, beg, size):
        out = (C.c_int64 * max(size, 1))()
        rc = lib().orc_sar_get_range(self._h, beg, size, out)
        if rc != 0:
            raise ValueError("sar_get_range(%d,%d) out of bounds" % (beg, size))
        return list(out[:size])

    def encode(self, s):
        b = s.encode() if isinstance(s, str) else bytes(s)
        out = (C.c_uint8 * max(len(b), 1))()
        lib().orc_encode(self._h, b, len(b), out)
        return np.frombuffer(out, dtype=np.uint8, count=len(b)).copy()

    def pos_to_str(self, pos):
        buf = C.create_string_buffer(512)
        n = lib().orc_pos_to_str(self._h, pos, buf, 512)
        if n < 0:
            raise ValueError("position out of bounds")
        return buf.value.decode()

    @staticmethod
    def _hits_to_list(ptr, n):
        res = []
        for i in range(n):
            h = ptr[i]
            res.append(dict(fp=h.fp, rp=h.rp, sz=h.sz, dp=h.dp, score=h.score, align=list(h.align)))
        return res

    def blocksearch(self, codes, tau):
        """blocksearch_trail_rec(q,0,m-1,tau+1,0,...): list of hits in reference order, or None on error."""
        q = np.ascontiguousarray(codes, dtype=np.uint8)
        hp = C.POINTER(OrcHit)()
        next_ = C.c_int64(0)
        n = lib().orc_blocksearch(self._h, q.ctypes.data_as(C.POINTER(C.c_uint8)), len(q), tau,
                                  C.byref(hp), C.byref(next_))
        if n < 0:
            return None
        res = self._hits_to_list(hp, n)
        lib()._libc.free(C.cast(hp, C.c_void_p))
        return res

    def seqsearch(self, direction, start, codes, pos, end, tau, score_ref=0, score_diff=0):
        q = np.ascontiguousarray(codes, dtype=np.uint8)
        st = OrcHit(start.get("fp", 0), start.get("rp", 0), start.get("sz", self.txt_len),
                    start.get("dp", 0), start.get("score", 0), (C.c_uint64 * 4)(*start.get("align", [0] * 4)))
        hp = C.POINTER(OrcHit)()
        n = lib().orc_seqsearch(self._h, direction, C.byref(st), q.ctypes.data_as(C.POINTER(C.c_uint8)),
                                pos, end, tau, score_ref, score_diff, C.byref(hp))
        res = self._hits_to_list(hp, n)
        lib()._libc.free(C.cast(hp, C.c_void_p))
        return res

    def scsearch(self, start, codes, pos, end, tau, score_ref=0, score_diff=0, boundary=1):
        q = np.ascontiguousarray(codes, dtype=np.uint8)
        st = OrcHit(start.get("fp", 0), start.get("rp", 0), start.get("sz", self.txt_len),
                    start.get("dp", 0), start.get("score", 0), (C.c_uint64 * 4)(*start.get("align", [0] * 4)))
        hp = C.POINTER(OrcHit)()
        n = lib().orc_scsearch(self._h, C.byref(st), q.ctypes.data_as(C.POINTER(C.c_uint8)),
                               pos, end, tau, score_ref, score_diff, boundary, C.byref(hp))
        res = self._hits_to_list(hp, n)
        lib()._libc.free(C.cast(hp, C.c_void_p))
        return res

    def blocksc(self, codes, tau):
        """blocksc_trail(q, qarray, slen, tau, 0, tree): list of hits in reference order, or None on error."""
        q = np.ascontiguousarray(codes, dtype=np.uint8)
        hp = C.POINTER(OrcHit)()
        next_ = C.c_int64(0)
        n = lib().orc_blocksc(self._h, q.ctypes.data_as(C.POINTER(C.c_uint8)), len(q), tau,
                              C.byref(hp), C.byref(next_))
        if n < 0:
            return None
        res = self._hits_to_list(hp, n)
        lib()._libc.free(C.cast(hp, C.c_void_p))
        return res

    def blocksc_batch(self, codes, offsets, tau, threads=1):
        return self.map_batch(codes, offsets, tau, threads, _fn="orc_blocksc_batch")

    def map_batch(self, codes, offsets, tau, threads=1, _fn="orc_map_batch"):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n_reads = len(offsets) - 1
        cp = codes.ctypes.data_as(C.POINTER(C.c_uint8)) if len(codes) else C.POINTER(C.c_uint8)()
        bp = getattr(lib(), _fn)(self._h, cp, offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                 n_reads, tau, threads)
        b = bp.contents
        status = np.ctypeslib.as_array(b.status, shape=(max(n_reads, 1),))[:n_reads].copy()
        hits = np.ctypeslib.as_array(b.hits, shape=(max(b.n_hits, 1) * 11,))[:b.n_hits * 11].copy()
        locs = np.ctypeslib.as_array(b.locs, shape=(max(b.n_locs, 1),))[:b.n_locs].copy()
        d = HitDump(status, hits, locs, n_ext=int(b.n_ext), seconds=float(b.seconds))
        lib().orc_batch_free(bp)
        return d


def ref_available():
    return os.path.exists(REF_DRIVER)
