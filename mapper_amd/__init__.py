"""mapper_amd -- thin Python binding (ctypes) of libgpumap.so, the MI355X-native query hot path
of ezorita/mapper.  The library is the product; this module only hands numpy buffers to the C-ABI
declared in include/gpumap.h and never computes results itself.  There is no CPU fallback: if the
HIP library is missing, importing the compute classes fails loudly.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPUMAP_LIB") or os.path.join(_HERE, "lib", "libgpumap.so")      # (GPUMAP_LIB: A/B runs of two builds)
QUERY_PREFIX, QUERY_SUFFIX = 0, 1
SRC_TEXT, SRC_DEVICE, PIPE_COMPACT = 1, 2, 4
PIPE_PACKED = 32
MAX_TAU = 7
MAX_READ_LEN = 256
NOALLOC_MAX_QUERIES = 16384


def build_library(force=False):
    """Compile libgpumap.so for gfx950 (hipcc cross-compiles without a GPU)."""
    args = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return LIB_PATH


class IndexView(C.Structure):
    _fields_ = [
        ("occ", C.c_void_p), ("occ_length", C.c_int64), ("occ_mark_intv", C.c_int64),
        ("occ_word_size", C.c_int64), ("c", C.c_void_p),
        ("sa", C.c_void_p), ("sar_bits", C.c_int64), ("sar_len", C.c_int64),
        ("txt_len", C.c_int64), ("seq_cnt", C.c_int64), ("seq_len", C.c_void_p), ("seq_beg", C.c_void_p),
        ("seq_name", C.POINTER(C.c_char_p)), ("wil_cnt", C.c_int64), ("rc_flag", C.c_int64), ("text", C.c_void_p),
        ("sym_count", C.c_int32), ("sym_table", C.c_void_p), ("com_table", C.c_void_p),
    ]


class DevIndex(C.Structure):
    _fields_ = [("occ_blocks", C.c_void_p), ("occ_bytes", C.c_int64), ("sa_words", C.c_void_p),
                ("sa_bytes", C.c_int64), ("tables", C.c_void_p), ("tables_bytes", C.c_int64),
                ("text", C.c_void_p), ("text_bytes", C.c_int64)]


class _Hits(C.Structure):
    _fields_ = [
        ("n_reads", C.c_int64), ("status", C.POINTER(C.c_int64)),
        ("n_locs", C.c_int64), ("read_first_loc", C.POINTER(C.c_int64)), ("loc_read", C.POINTER(C.c_int64)),
        ("loc_pos", C.POINTER(C.c_int64)), ("loc_fwd", C.POINTER(C.c_int64)), ("loc_seq", C.POINTER(C.c_int64)),
        ("loc_seqpos", C.POINTER(C.c_int64)), ("loc_strand", C.POINTER(C.c_uint8)), ("loc_score", C.POINTER(C.c_uint8)),
        ("n_hits", C.c_int64), ("hit_read", C.POINTER(C.c_int64)), ("hit_fp", C.POINTER(C.c_int64)),
        ("hit_rp", C.POINTER(C.c_int64)), ("hit_sz", C.POINTER(C.c_int64)), ("hit_dp", C.POINTER(C.c_int64)),
        ("hit_score", C.POINTER(C.c_int64)), ("hit_align", C.POINTER(C.c_uint64)),
        ("n_ext", C.c_int64), ("ms_search", C.c_float), ("ms_locate", C.c_float), ("ms_sort", C.c_float),
        ("ms_total", C.c_float), ("arena", C.c_void_p),
        ("loc_key", C.POINTER(C.c_uint64)), ("key_pos_bits", C.c_int32), ("expanded", C.c_int32),
        ("line_text", C.c_void_p), ("line_bytes", C.c_int64),
        ("loc_packed", C.POINTER(C.c_uint8)), ("read_first32", C.POINTER(C.c_uint32)), ("status8", C.POINTER(C.c_uint8)),
    ]


# every symbol include/gpumap.h declares: name -> (restype, argtypes)
_P, _I64, _I = C.c_void_p, C.c_int64, C.c_int
API = {
    "gpumap_index_read": (_P, [C.c_char_p]),
    "gpumap_index_build": (_P, [C.c_char_p, C.c_char_p]),
    "gpumap_index_free": (None, [_P]),
    "gpumap_index_view_of": (_I, [_P, C.POINTER(IndexView)]),
    "gpumap_pos_to_str": (_I, [_P, _I64, C.c_char_p, C.c_size_t]),
    "gpumap_create": (_P, [C.POINTER(IndexView), _I]),
    "gpumap_create_from_files": (_P, [C.c_char_p, _I]),
    "gpumap_destroy": (None, [_P]),
    "gpumap_devindex_header_size": (C.c_size_t, []),
    "gpumap_devindex_export": (_I, [_P, C.POINTER(DevIndex), _P]),
    "gpumap_create_from_device": (_P, [C.POINTER(DevIndex), _P, _I]),
    "gpumap_create_multi": (_P, [C.POINTER(IndexView), _I, _P]),
    "gpumap_multi_from_ctx": (_P, [_P, _I, _I, _P]),
    "gpumap_multi_size": (_I, [_P]),
    "gpumap_multi_used_rccl": (_I, [_P]),
    "gpumap_multi_ctx": (_P, [_P, _I]),
    "gpumap_multi_ctxs": (C.POINTER(_P), [_P]),
    "gpumap_multi_map_batch": (_I, [_P, _P, _P, _I64, _I, _I, C.POINTER(_Hits)]),
    "gpumap_multi_destroy": (None, [_P]),
    "gpumap_occ_all": (_I, [_P, _I64, _P, _P]),
    "gpumap_query_all": (_I, [_P, _I, _I64, _P, _P, _P, _P]),
    "gpumap_query_all_noalloc": (_I, [_P, _I, _I64, _P, _P, _P, _P]),
    "gpumap_prefix_all": (_I, [_P, _I64, _P, _P, _P]),
    "gpumap_ann_build": (_I, [_P, _I, _I, _P, _P]),
    "gpumap_ann_query": (_I, [_P, _I64, _I, _I64, _P, _P, _P, _P]),
    "gpumap_ann_file_write": (_I, [C.c_char_p, _P, _I64, _I, _I]),
    "gpumap_sar_get": (_I, [_P, _I64, _P, _P]),
    "gpumap_sar_get_range": (_I, [_P, _I64, _I64, _P]),
    "gpumap_encode": (_I, [_P, _P, _P, _I64, _P]),
    "gpumap_map_batch": (_I, [_P, _P, _P, _I64, _I, _I, C.POINTER(_Hits)]),
    "gpumap_hits_free": (None, [C.POINTER(_Hits)]),
    "gpumap_blocksc_batch": (_I, [_P, _P, _P, _I64, _I, C.POINTER(_Hits)]),
    "gpumap_batch_upload": (_P, [_P, _P, _P, _I64]),
    "gpumap_batch_upload_text": (_P, [_P, _P, _P, _I64]),
    "gpumap_batch_refill_text": (_I, [_P, _P, _P, _P, _I64]),
    "gpumap_batch_new": (_P, [_P]),
    "gpumap_batch_stage": (_I, [_P, _P, _P, _P, _I64, _I]),
    "gpumap_hits_expand": (_I, [_P, C.POINTER(_Hits)]),
    "gpumap_batch_stats_ext": (_I, [_P, C.POINTER(_I64)]),
    "gpumap_pipe_create": (_P, [C.POINTER(_P), _I, _I, _I, _I]),
    "gpumap_pipe_try_submit": (_I, [_P, _P, _P, _I64, _P]),
    "gpumap_set_seq_names": (_I, [_P, _P, _I64]),
    "gpumap_host_trim": (None, []),
    "gpumap_multi_setup_ms": (C.c_double, [_P, _I]),
    "gpumap_pipe_submit": (_I, [_P, _P, _P, _I64, _P]),
    "gpumap_pipe_close": (None, [_P]),
    "gpumap_pipe_in_flight": (_I64, [_P]),
    "gpumap_pipe_next": (_I, [_P, C.POINTER(_Hits), C.POINTER(_P)]),
    "gpumap_pipe_release": (None, [_P, C.POINTER(_Hits)]),
    "gpumap_pipe_destroy": (None, [_P]),
    "gpumap_mapper_multi": (_I, [C.c_char_p, _P, C.POINTER(_P), _I, _I, C.c_char_p]),
    "gpumap_batch_run": (_I, [_P, _P, _I, _I]),
    "gpumap_batch_fetch": (_I, [_P, _P, C.POINTER(_Hits)]),
    "gpumap_batch_counters": (_I, [_P, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64), C.POINTER(C.c_float)]),
    "gpumap_batch_stats": (_I, [_P, C.POINTER(_I64)]),
    "gpumap_batch_free": (None, [_P]),
    "gpumap_mapper": (_I, [C.c_char_p, _P, _P, _I, C.c_char_p]),
    "gpumap_set_option": (_I, [_P, C.c_char_p, _I64]),
    "gpumap_host_alloc": (_P, [C.c_size_t]),
    "gpumap_host_free": (None, [_P]),
    "gpumap_build_device": (_P, [_P, _I, _P, _I64, _I, _I]),
    "gpumap_reference_sizes": (_I, [_P, C.POINTER(_I64)]),
    "gpumap_reference_fetch": (_I, [_P, _P, _P, _P, _P]),
    "gpumap_version": (C.c_char_p, []),
}
# host-side extra (not in the reference): index from code arrays, for synthetic genomes
API_EXTRA = {
    "gpumap_index_build_codes": (_P, [_P, _P, C.POINTER(C.c_char_p), _I64, C.c_char_p]),
    "gpumap_suffix_sort": (_I, [_P, _I64, _P]),
    "gpumap_reads_dump": (_I64, [C.c_char_p, C.c_size_t, C.c_char_p]),
    "gpumap_reads_dump_blocks": (_I64, [C.c_char_p, C.c_size_t, C.c_char_p]),
}

_lib = None


def lib():
    """Loads libgpumap.so; raises if it is not built.  No fallback of any kind."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libgpumap.so is not built (%s): run __graft_entry__.build() or make -C mapper_amd/csrc"
                               % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(API.items()) + list(API_EXTRA.items()):
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


class GpumapError(RuntimeError):
    pass


class HostIndex:
    """index_t of the reference (src/index.h:19-33): the four mmapped files, or a fresh build."""

    def __init__(self, handle):
        if not handle:
            raise GpumapError("index could not be read or built (see stderr)")
        self._h = handle
        self.view = IndexView()
        if lib().gpumap_index_view_of(self._h, C.byref(self.view)) != 0:
            raise GpumapError("gpumap_index_view_of failed")

    @classmethod
    def read(cls, base):
        return cls(lib().gpumap_index_read(os.fsencode(base)))

    @classmethod
    def build(cls, fasta, base):
        return cls(lib().gpumap_index_build(os.fsencode(fasta), os.fsencode(base)))

    @classmethod
    def build_codes(cls, seqs, names, base=None):
        """seqs: list of uint8 code arrays (0..4)."""
        codes = np.ascontiguousarray(np.concatenate(seqs) if len(seqs) else np.zeros(0, np.uint8), dtype=np.uint8)
        lens = np.array([len(s) for s in seqs], dtype=np.int64)
        arr = (C.c_char_p * len(names))(*[n.encode() for n in names])
        return cls(lib().gpumap_index_build_codes(_ptr(codes) or 0, _ptr(lens), arr, len(seqs),
                                                  os.fsencode(base) if base else None))

    @property
    def txt_len(self):
        return self.view.txt_len

    def seq_names(self):
        return [self.view.seq_name[i].decode() for i in range(self.view.seq_cnt)]

    def pos_to_str(self, pos):
        buf = C.create_string_buffer(1024)
        n = lib().gpumap_pos_to_str(self._h, pos, buf, 1024)
        if n < 0:
            raise GpumapError("position out of bounds")
        return buf.value.decode()

    def close(self):
        if self._h:
            lib().gpumap_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Hits:
    """Result of one batch as numpy arrays (copied out of the library's malloc'ed buffers)."""

    def __init__(self, raw):
        def arr(p, n, dt):
            return np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].astype(dt, copy=True) if n > 0 else np.zeros(0, dt)
        R, L, H = raw.n_reads, raw.n_locs, raw.n_hits
        self.n_reads, self.n_locs, self.n_hits, self.n_ext = R, L, H, raw.n_ext
        self.key_pos_bits = raw.key_pos_bits
        if raw.loc_packed and not raw.loc_key:
            # packed result (GPUMAP_PIPE_PACKED): 5 bytes per hit, 4-byte offsets, a status byte per read
            self.status = -arr(raw.status8, R, np.int64)
            self.read_first_loc = arr(raw.read_first32, R + 1, np.int64)
            p = arr(raw.loc_packed, 5 * L, np.uint64).reshape(-1, 5)
            low = p[:, 0] | (p[:, 1] << np.uint64(8)) | (p[:, 2] << np.uint64(16)) | (p[:, 3] << np.uint64(24)) | (p[:, 4] << np.uint64(32))
            reads = np.repeat(np.arange(R, dtype=np.uint64), np.diff(self.read_first_loc))
            self.loc_key = (reads << np.uint64(raw.key_pos_bits + 4)) | low
        else:
            self.status = arr(raw.status, R, np.int64)
            self.read_first_loc = arr(raw.read_first_loc, R + 1, np.int64)
            self.loc_key = arr(raw.loc_key, L, np.uint64) if raw.loc_key else None
        if raw.loc_read:
            for k in ("loc_read", "loc_pos", "loc_fwd", "loc_seq", "loc_seqpos"):
                setattr(self, k, arr(getattr(raw, k), L, np.int64))
            self.loc_strand = arr(raw.loc_strand, L, np.uint8)
            self.loc_score = arr(raw.loc_score, L, np.uint8)
        else:
            # compact result: read, text position and score follow from the key (GPUMAP_KEY_*)
            b = np.uint64(raw.key_pos_bits)
            self.loc_read = (self.loc_key >> (b + np.uint64(4))).astype(np.int64)
            self.loc_pos = ((self.loc_key >> np.uint64(4)) & ((np.uint64(1) << b) - np.uint64(1))).astype(np.int64)
            self.loc_score = (self.loc_key & np.uint64(15)).astype(np.uint8)
        if raw.hit_read:
            for k in ("hit_read", "hit_fp", "hit_rp", "hit_sz", "hit_dp", "hit_score"):
                setattr(self, k, arr(getattr(raw, k), H, np.int64))
            self.hit_align = arr(raw.hit_align, 4 * H, np.uint64).reshape(-1, 4)
        self.ms = dict(search=raw.ms_search, locate=raw.ms_locate, sort=raw.ms_sort, total=raw.ms_total)

    def located(self):
        """int64 [n_locs][3] = (read, text_pos, score): reads in input order, positions ascending."""
        return np.stack([self.loc_read, self.loc_pos, self.loc_score.astype(np.int64)], axis=1) if self.n_locs else \
            np.zeros((0, 3), np.int64)

    def intervals(self):
        """int64 [n_hits][10] = read, fp, rp, sz, dp, score, align0..3, sorted by (read, fp)."""
        if self.n_hits == 0:
            return np.zeros((0, 10), np.int64)
        return np.concatenate([np.stack([self.hit_read, self.hit_fp, self.hit_rp, self.hit_sz, self.hit_dp,
                                         self.hit_score], axis=1), self.hit_align.view(np.int64)], axis=1)


class Batch:
    def __init__(self, ctx, handle, n_reads):
        if not handle:
            raise GpumapError("batch upload failed (see stderr)")
        self.ctx, self._h, self.n_reads = ctx, handle, n_reads

    def run(self, tau, want_intervals=False):
        if lib().gpumap_batch_run(self.ctx._h, self._h, tau, int(want_intervals)) != 0:
            raise GpumapError("gpumap_batch_run failed (see stderr)")
        return self.counters()

    def counters(self):
        nh, nl, ne = C.c_int64(), C.c_int64(), C.c_int64()
        ms = (C.c_float * 4)()
        if lib().gpumap_batch_counters(self._h, C.byref(nh), C.byref(nl), C.byref(ne), ms) != 0:
            raise GpumapError("gpumap_batch_counters failed")
        st = (_I64 * 8)()
        if lib().gpumap_batch_stats(self._h, st) != 0:
            raise GpumapError("gpumap_batch_stats failed")
        return dict(n_hits=nh.value, n_locs=nl.value, n_ext=ne.value, ms_search=ms[0], ms_locate=ms[1],
                    ms_sort=ms[2], ms_total=ms[3], loop_passes=st[0], continuations=st[1], jumped_steps=st[2], list_entries=st[3],
                    verify_steps=st[4], list_length=st[5], us_k_search=st[6], us_k_verify=st[7])

    def fetch(self, expand=False):
        raw = _Hits()
        if lib().gpumap_batch_fetch(self.ctx._h, self._h, C.byref(raw)) != 0:
            raise GpumapError("gpumap_batch_fetch failed (see stderr)")
        if expand and lib().gpumap_hits_expand(self.ctx._h, C.byref(raw)) != 0:
            raise GpumapError("gpumap_hits_expand failed")
        h = Hits(raw)
        lib().gpumap_hits_free(C.byref(raw))
        return h

    def stage(self, src, offsets, flags=0):
        """gpumap_batch_stage: host numpy arrays, or (flags & SRC_DEVICE) integer device pointers."""
        if flags & SRC_DEVICE:
            sp, op, n = C.c_void_p(int(src)), C.c_void_p(int(offsets[0])), int(offsets[1])
        else:
            self._src = np.ascontiguousarray(src, dtype=np.uint8)
            self._off = np.ascontiguousarray(offsets, dtype=np.int64)
            sp, op, n = _ptr(self._src), _ptr(self._off), len(self._off) - 1
        if lib().gpumap_batch_stage(self.ctx._h, self._h, sp, op, n, flags) != 0:
            raise GpumapError("gpumap_batch_stage failed (see stderr)")
        self.n_reads = n

    def stats_ext(self):
        st = (_I64 * 8)()
        if lib().gpumap_batch_stats_ext(self._h, st) != 0:
            raise GpumapError("gpumap_batch_stats_ext failed")
        return dict(attempts=st[0], us_encode=st[1], compact=st[2], hit_list_capacity=st[3], spill_chunks=st[4],
                    verify_windows=st[5], big_candidates=st[6])

    def close(self):
        if self._h:
            lib().gpumap_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Multi:
    """gpumap_multi: replicas of one index on several GPUs (or several members on one GPU)."""

    def __init__(self, handle):
        if not handle:
            raise GpumapError("gpumap_multi could not be created (see stderr)")
        self._h = handle

    @classmethod
    def from_index(cls, index, n_members, devices=None):
        d = np.ascontiguousarray(devices, dtype=np.int32) if devices is not None else None
        return cls(lib().gpumap_create_multi(C.byref(index.view), n_members, _ptr(d) if d is not None else None))

    @classmethod
    def from_ctx(cls, ctx, n_members, devices=None, root_device=0):
        d = np.ascontiguousarray(devices, dtype=np.int32) if devices is not None else None
        m = cls(lib().gpumap_multi_from_ctx(ctx._h, root_device, n_members, _ptr(d) if d is not None else None))
        m._root = ctx
        return m

    def __len__(self):
        return lib().gpumap_multi_size(self._h)

    @property
    def used_rccl(self):
        return bool(lib().gpumap_multi_used_rccl(self._h))

    def setup_ms(self):
        """{"broadcast": ms, "members": [ms per member; member 0 is the root]}"""
        return {"broadcast": lib().gpumap_multi_setup_ms(self._h, -1), "members": [lib().gpumap_multi_setup_ms(self._h, i) for i in range(len(self))]}

    def members(self):
        """The members as Context objects that do not own their handles."""
        out = []
        for i in range(len(self)):
            c = Context.__new__(Context)
            c._h, c._keep, c._borrowed = lib().gpumap_multi_ctx(self._h, i), self, True
            out.append(c)
        return out

    def map_batch(self, codes, offsets, tau, want_intervals=False):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        raw = _Hits()
        if lib().gpumap_multi_map_batch(self._h, _ptr(codes), _ptr(offsets), len(offsets) - 1, tau, int(want_intervals),
                                        C.byref(raw)) != 0:
            raise GpumapError("gpumap_multi_map_batch failed (see stderr)")
        h = Hits(raw)
        lib().gpumap_hits_free(C.byref(raw))
        return h

    def mapper(self, qfile, index, tau, out_path):
        rc = lib().gpumap_mapper_multi(os.fsencode(qfile), index._h, lib().gpumap_multi_ctxs(self._h), len(self), tau,
                                       os.fsencode(out_path) if out_path else None)
        if rc != 0:
            raise GpumapError("gpumap_mapper_multi failed (see stderr)")

    def close(self):
        if self._h:
            lib().gpumap_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def ann_query(table, kmer, pos):
    """gpumap_ann_query: (dist, neigh_cnt, align positions) of the reference's locinfo_t."""
    table = np.ascontiguousarray(table, dtype=np.uint8)
    d, c, a = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    pos_out = np.zeros(max(kmer, 1), dtype=np.int32)
    if lib().gpumap_ann_query(_ptr(table), len(table), kmer, pos, C.byref(d), C.byref(c), C.byref(a), _ptr(pos_out)) != 0:
        raise GpumapError("gpumap_ann_query failed")
    return d.value, c.value, pos_out[:a.value].tolist()


class Pipe:
    """gpumap_pipe: chunks in, results out in the same order, the stages of consecutive chunks overlapping."""

    def __init__(self, ctxs, depth, tau, flags):
        self._ctxs = list(ctxs)
        arr = (_P * len(ctxs))(*[c._h for c in ctxs])
        self._h = lib().gpumap_pipe_create(arr, len(ctxs), depth, tau, flags)
        if not self._h:
            raise GpumapError("gpumap_pipe_create failed (see stderr)")
        self.flags = flags
        self._keep = {}
        self._seq = 0

    def submit(self, src, offsets, n_reads=None):
        """Host numpy arrays (kept alive until the result is back), or with SRC_DEVICE integer device pointers + n_reads."""
        self._seq += 1
        if self.flags & SRC_DEVICE:
            sp, op, n = C.c_void_p(int(src)), C.c_void_p(int(offsets)), int(n_reads)
        else:
            src = np.ascontiguousarray(src, dtype=np.uint8)
            offsets = np.ascontiguousarray(offsets, dtype=np.int64)
            self._keep[self._seq] = (src, offsets)
            sp, op, n = _ptr(src), _ptr(offsets), len(offsets) - 1
        if lib().gpumap_pipe_submit(self._h, sp, op, n, C.c_void_p(self._seq)) != 0:
            self._keep.pop(self._seq, None)
            raise GpumapError("gpumap_pipe_submit failed (see stderr)")
        return self._seq

    def try_submit(self, src, offsets):
        """gpumap_pipe_try_submit with host numpy arrays: the chunk's tag, or None when every slot is taken."""
        src = np.ascontiguousarray(src, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        rc = lib().gpumap_pipe_try_submit(self._h, _ptr(src), _ptr(offsets), len(offsets) - 1, C.c_void_p(self._seq + 1))
        if rc < 0:
            raise GpumapError("gpumap_pipe_try_submit failed (see stderr)")
        if rc == 0:
            return None
        self._seq += 1
        self._keep[self._seq] = (src, offsets)
        return self._seq

    def next_raw(self):
        """(raw gpumap_hits, tag) or None when nothing is in flight; release_raw() when done with it."""
        if lib().gpumap_pipe_in_flight(self._h) == 0:
            return None
        raw = _Hits()
        user = _P()
        rc = lib().gpumap_pipe_next(self._h, C.byref(raw), C.byref(user))
        if rc < 0:
            raise GpumapError("a chunk failed in the pipeline (see stderr)")
        if rc == 0:
            return None
        self._keep.pop(user.value, None)
        return raw, user.value

    def release_raw(self, raw):
        lib().gpumap_pipe_release(self._h, C.byref(raw))

    def next(self):
        got = self.next_raw()
        if got is None:
            return None
        h = Hits(got[0])
        self.release_raw(got[0])
        return h

    def close(self):
        if self._h:
            lib().gpumap_pipe_close(self._h)
            lib().gpumap_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """One GPU: device-resident index (gpumap_ctx)."""

    def __init__(self, handle, keep=None):
        if not handle:
            raise GpumapError("gpumap context could not be created (see stderr)")
        self._h = handle
        self._keep = keep

    @classmethod
    def from_index(cls, index, device=0):
        return cls(lib().gpumap_create(C.byref(index.view), device))

    @classmethod
    def from_files(cls, base, device=0):
        return cls(lib().gpumap_create_from_files(os.fsencode(base), device))

    @classmethod
    def from_device(cls, dev, header, device=0, keep=None):
        hb = np.frombuffer(bytes(header), dtype=np.uint8).copy()
        return cls(lib().gpumap_create_from_device(C.byref(dev), _ptr(hb), device), keep=keep)

    @classmethod
    def build_device(cls, seqs, device=0, keep_reference=False, device_ptr=None):
        """index_build on the GPU.  seqs: list of uint8 code arrays, or (device_ptr != None) their
        lengths only, the codes lying one after another in device memory at device_ptr."""
        if device_ptr is None:
            codes = np.ascontiguousarray(np.concatenate(seqs), dtype=np.uint8)
            lens = np.array([len(x) for x in seqs], dtype=np.int64)
            return cls(lib().gpumap_build_device(_ptr(codes), 0, _ptr(lens), len(lens), device, int(keep_reference)))
        lens = np.ascontiguousarray(seqs, dtype=np.int64)
        return cls(lib().gpumap_build_device(C.c_void_p(int(device_ptr)), 1, _ptr(lens), len(lens), device, int(keep_reference)))

    def reference_arrays(self, want_text=True):
        """Host copies of the arrays a device build kept: dict(txt_len, sar_bits, wil_cnt, occ, c, sa, text)."""
        sz = (_I64 * 6)()
        if lib().gpumap_reference_sizes(self._h, sz) != 0:
            raise GpumapError("gpumap_reference_sizes failed")
        occ = np.empty(sz[1], dtype=np.uint64)
        c = np.zeros(6, dtype=np.uint64)
        sa = np.empty(sz[2], dtype=np.int64)
        text = np.empty(sz[5], dtype=np.uint8) if want_text and sz[5] else None
        if lib().gpumap_reference_fetch(self._h, _ptr(occ), _ptr(c), _ptr(sa), _ptr(text) if text is not None else None) != 0:
            raise GpumapError("gpumap_reference_fetch failed (see stderr)")
        return dict(txt_len=sz[0], sar_bits=sz[3], wil_cnt=sz[4], occ=occ, c=c, sa=sa, text=text)

    def export(self):
        dev = DevIndex()
        hdr = np.zeros(lib().gpumap_devindex_header_size(), dtype=np.uint8)
        if lib().gpumap_devindex_export(self._h, C.byref(dev), _ptr(hdr)) != 0:
            raise GpumapError("gpumap_devindex_export failed")
        return dev, hdr

    def set_option(self, name, value):
        if lib().gpumap_set_option(self._h, name.encode(), int(value)) != 0:
            raise GpumapError("gpumap_set_option(%s) failed" % name)

    # ---- batched operators
    def occ_all(self, ptr):
        ptr = np.ascontiguousarray(ptr, dtype=np.int64)
        out = np.empty((len(ptr), 5), dtype=np.int64)
        if lib().gpumap_occ_all(self._h, len(ptr), _ptr(ptr), _ptr(out)) != 0:
            raise GpumapError("gpumap_occ_all failed")
        return out

    def query_all(self, end, fp, rp, sz):
        fp, rp, sz = (np.ascontiguousarray(x, dtype=np.int64) for x in (fp, rp, sz))
        out = np.empty((len(fp), 5, 3), dtype=np.int64)
        if lib().gpumap_query_all(self._h, end, len(fp), _ptr(fp), _ptr(rp), _ptr(sz), _ptr(out)) != 0:
            raise GpumapError("gpumap_query_all failed")
        return out

    def prefix_all(self, fp, sz):
        """bwt_prefix_all for len(fp) intervals: out[i][j] = (fp, sz) of the prefix extension by symbol j."""
        fp, sz = (np.ascontiguousarray(x, dtype=np.int64) for x in (fp, sz))
        out = np.empty((len(fp), 5, 2), dtype=np.int64)
        if lib().gpumap_prefix_all(self._h, len(fp), _ptr(fp), _ptr(sz), _ptr(out)) != 0:
            raise GpumapError("gpumap_prefix_all failed")
        return out

    def ann_build(self, kmer, tau, threads=1, trail_reuse=True):
        """gpumap_ann_build: the annotation table (txt_len / 2 bytes) and the work figures.  threads = the `threads` of the
        reference's ann_build whose job list is reproduced; trail_reuse = with the reference's reuse of the last k-mer's
        search (False: every k-mer searched afresh)."""
        self.set_option("ann_threads", threads)
        self.set_option("ann_trail_reuse", 1 if trail_reuse else 0)
        txt_len = int(self.export()[1][:16].view(np.int64)[1])        # (the header of an exported index: magic, n, ...)
        table = np.zeros(txt_len // 2, dtype=np.uint8)
        stats = np.zeros(8, dtype=np.int64)
        if lib().gpumap_ann_build(self._h, kmer, tau, _ptr(table), _ptr(stats)) != 0:
            raise GpumapError("gpumap_ann_build failed (see stderr)")
        return table, {"queries": int(stats[0]), "n_ext": int(stats[1]), "hits": int(stats[2]), "updates": int(stats[3]),
                       "us": int(stats[4]), "jump2_k": int(stats[5]), "split": int(stats[6])}

    def query_all_noalloc(self, end, fp, rp, sz):
        """gpumap_query_all_noalloc: at most NOALLOC_MAX_QUERIES intervals, staged in the library's static scratch."""
        fp, rp, sz = (np.ascontiguousarray(x, dtype=np.int64) for x in (fp, rp, sz))
        out = np.empty((len(fp), 5, 3), dtype=np.int64)
        if lib().gpumap_query_all_noalloc(self._h, end, len(fp), _ptr(fp), _ptr(rp), _ptr(sz), _ptr(out)) != 0:
            raise GpumapError("gpumap_query_all_noalloc failed (see stderr)")
        return out

    def sar_get(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.int64)
        out = np.empty(len(pos), dtype=np.int64)
        if lib().gpumap_sar_get(self._h, len(pos), _ptr(pos), _ptr(out)) != 0:
            raise GpumapError("gpumap_sar_get failed")
        return out

    def sar_get_range(self, beg, size):
        out = np.empty(max(size, 1), dtype=np.int64)
        if lib().gpumap_sar_get_range(self._h, beg, size, _ptr(out)) != 0:
            raise GpumapError("gpumap_sar_get_range failed")
        return out[:size]

    def encode(self, text, offsets):
        t = np.frombuffer(text if isinstance(text, bytes) else text.encode(), dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        out = np.zeros(len(t), dtype=np.uint8)
        if lib().gpumap_encode(self._h, _ptr(t), _ptr(offsets), len(offsets) - 1, _ptr(out)) != 0:
            raise GpumapError("gpumap_encode failed")
        return out

    # ---- mapping
    def new_batch(self):
        return Batch(self, lib().gpumap_batch_new(self._h), 0)

    def pipe(self, depth=3, tau=2, flags=0, others=()):
        return Pipe([self] + list(others), depth, tau, flags)

    def upload(self, codes, offsets):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        return Batch(self, lib().gpumap_batch_upload(self._h, _ptr(codes), _ptr(offsets), len(offsets) - 1),
                     len(offsets) - 1)

    def upload_text(self, text, offsets):
        t = np.frombuffer(text if isinstance(text, bytes) else text.encode(), dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        return Batch(self, lib().gpumap_batch_upload_text(self._h, _ptr(t), _ptr(offsets), len(offsets) - 1),
                     len(offsets) - 1)

    def map_batch(self, codes, offsets, tau, want_intervals=False):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        raw = _Hits()
        if lib().gpumap_map_batch(self._h, _ptr(codes), _ptr(offsets), len(offsets) - 1, tau, int(want_intervals),
                                  C.byref(raw)) != 0:
            raise GpumapError("gpumap_map_batch failed (see stderr)")
        h = Hits(raw)
        lib().gpumap_hits_free(C.byref(raw))
        return h

    def blocksc_batch(self, codes, offsets, tau):
        """blocksc_trail (trail 0) for a batch of queries: interval hits + located hits."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        raw = _Hits()
        if lib().gpumap_blocksc_batch(self._h, _ptr(codes), _ptr(offsets), len(offsets) - 1, tau, C.byref(raw)) != 0:
            raise GpumapError("gpumap_blocksc_batch failed (see stderr)")
        h = Hits(raw)
        lib().gpumap_hits_free(C.byref(raw))
        return h

    def mapper(self, qfile, index, tau, out_path):
        """mapper() of src/mapper.c."""
        rc = lib().gpumap_mapper(os.fsencode(qfile), index._h, self._h, tau, os.fsencode(out_path) if out_path else None)
        if rc != 0:
            raise GpumapError("gpumap_mapper failed (see stderr)")

    def close(self):
        if self._h and not getattr(self, "_borrowed", False):
            lib().gpumap_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
