"""Index construction on the device (gpumap_build_device) against the host builder, whose files are
byte-identical to the reference's (tests/test_host.py): every array must match bit for bit."""
import os

import numpy as np
import pytest

import mapper_amd as ma
from mapper_amd import synth
from oracle import pyoracle as po
from tests import common

pytestmark = pytest.mark.gpu


def _view_arrays(hix):
    import ctypes as C
    v = hix.view

    def arr(ptr, ctype, count):
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).copy()
    return (arr(v.occ, C.c_uint64, v.occ_length), arr(v.c, C.c_uint64, 6), arr(v.sa, C.c_int64, v.sar_len),
            arr(v.text, C.c_uint8, v.txt_len))


def _check(seqs, monkeypatch=None):
    names = ["s%d" % i for i in range(len(seqs))]
    hix = ma.HostIndex.build_codes(seqs, names)
    occ, c, sa, text = _view_arrays(hix)
    ctx = ma.Context.build_device(seqs, 0, keep_reference=True)
    got = ctx.reference_arrays()
    assert got["txt_len"] == hix.view.txt_len
    assert got["sar_bits"] == hix.view.sar_bits
    assert got["wil_cnt"] == hix.view.wil_cnt
    assert np.array_equal(got["text"], text)
    assert np.array_equal(got["c"], c)
    assert np.array_equal(got["sa"].view(np.uint64), sa.view(np.uint64))
    assert np.array_equal(got["occ"], occ)
    hix.close()
    return ctx


def test_device_build_small_texts():
    rng = np.random.default_rng(5)
    cases = [
        [np.array([3, 3, 0, 2, 1, 0, 2, 3, 0, 2, 3, 1, 2, 3, 0], np.uint8)],          # the reference's 32-symbol KAT text
        [np.array([0], np.uint8)],
        [np.zeros(300, np.uint8)],                                                     # one long run: deep doubling
        [np.full(70, 4, np.uint8), rng.integers(0, 4, 50).astype(np.uint8)],           # N run + a second sequence
        [np.tile(np.array([0, 1, 2, 3, 3, 2, 1, 0], np.uint8), 125)] * 2,               # periodic, two equal sequences
        [rng.integers(0, 5, 1000).astype(np.uint8) for _ in range(5)],
    ]
    for seqs in cases:
        _check(seqs).close()


@pytest.mark.parametrize("two_pass,classes", [(False, 1), (True, 1), (False, 2)])
def test_device_build_repeat_rich(two_pass, classes, monkeypatch):
    # the paths large texts take (two-pass group sort, classes of two leading symbols), forced at a small size
    if two_pass:
        monkeypatch.setenv("GPUMAP_BUILD_TWO_PASS", "1")
    monkeypatch.setenv("GPUMAP_BUILD_P", str(classes))
    monkeypatch.setenv("GPUMAP_BUILD_VERIFY", "100000")
    seqs, _ = synth.make_genome(400_000, 11, n_seqs=3, gc=0.41, repeat_rich=True)
    _check(seqs).close()


def test_device_built_index_maps_like_the_oracle(workdir):
    base, seqs = common.synth_index(workdir, "devbuild", 300_000, 21, n_seqs=2, repeat_rich=True)
    ctx = ma.Context.build_device(seqs, 0)
    orc = po.Index(base)
    codes, offs = synth.make_reads(seqs, 3000, 100, 2, seed=77)
    h = ctx.map_batch(codes.reshape(-1), offs, 2, want_intervals=True)
    d = orc.map_batch(codes.reshape(-1), offs, 2, threads=4)
    assert np.array_equal(h.status, d.status)
    assert np.array_equal(h.located(), d.located())
    assert h.n_ext == d.n_ext
    ctx.close()
