"""Pins the oracle's restatement of ann_build (src/index_ann.c:56-257: the job list, ann_build_mt / neigh_next, hits_push,
the compression with neigh_push) to the reference itself: tests/golden/*.ann.*.bin are the files the unmodified
reference wrote (tools/make_golden_ann.py -> oracle/_ref/ref_driver ann), threads = 1 and 2.

Two forms of the restatement:
  trail_reuse = True    with the reference's reuse of the last k-mer's search, as it is: every table byte for byte,
                        also where that reuse is stale (next to k-mers with N)
  trail_reuse = False   every k-mer searched afresh (blocksc_trail with trail = 0): gpumap_ann_build with option
                        ann_trail_reuse = 0.  Equal to the reference on every text without N; on the texts with N it
                        differs next to k-mers with N (counted here; annN.fa is made of such places)
and the known answers of test/test_index_ann.c:145-277 through gpumap_ann_query (host code of the library)."""
import glob
import os
import re

import numpy as np
import pytest

import mapper_amd as ma
from oracle import pyoracle as po
from tests import common

MAGIC = 0x0fcb0fcb00050001
FILES = sorted(glob.glob(os.path.join(common.GOLDEN, "*.ann.k*.tau*.t*.bin")))
CASES = []
for f in FILES:
    m = re.match(r"(\w+)\.ann\.k(\d+)\.tau(\d+)\.t(\d+)\.bin", os.path.basename(f))
    CASES.append((m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4))))
WITH_N = {"dense", "small", "annN"}


def load_golden(name, k, tau, threads):
    raw = np.fromfile(os.path.join(common.GOLDEN, "%s.ann.k%d.tau%d.t%d.bin" % (name, k, tau, threads)), dtype=np.uint8)
    head = raw[:32].view(np.int64)
    assert int(head[0]) == MAGIC and (int(head[1]), int(head[2])) == (k, tau) and int(head[3]) == len(raw) - 32
    return raw[32:]


def test_goldens_are_there():
    assert len(CASES) >= 12


@pytest.mark.parametrize("name,k,tau,threads", CASES)
def test_restatement_equals_the_reference_table(workdir, name, k, tau, threads):
    orc = po.Index(common.build_golden_index(workdir, name))
    ref = load_golden(name, k, tau, threads)
    got, st = orc.ann_build(k, tau, threads, trail_reuse=True)
    assert np.array_equal(got, ref)
    fresh, st2 = orc.ann_build(k, tau, threads, trail_reuse=False)
    assert st2["queries"] == st["queries"] and st2["n_ext"] >= st["n_ext"]          # the reuse only saves steps
    differ = int(np.count_nonzero(fresh != ref))
    if name not in WITH_N:
        assert differ == 0
    else:
        assert differ <= len(ref) // 4           # (dense, k 12, tau 2: 230 of 3002; small, k 24 / 16: none; annN, k 12, tau 3: 4415 of 20003)
        if name == "annN" and k < 16:
            assert differ > 100                  # these texts pin the stale reuse: a fresh search per k-mer gives another table


def test_bad_arguments(workdir):
    orc = po.Index(common.build_golden_index(workdir, "repeats"))
    for k, tau, th in ((1, 1, 1), (-1, 1, 1), (10, 0, 1), (20, -1, 1), (3, 4, 1), (25, 8, 1), (25, 1, 0)):     # test/test_index_ann.c:39-48
        assert orc.ann_build(k, tau, th)[0] is None


def test_kat_ann_query(workdir):
    """test/test_index_ann.c:132-277 (ann_build(25, 1) on the eight sequences of tests/golden/repeats.fa)."""
    orc = po.Index(common.build_golden_index(workdir, "repeats"))
    table, _ = orc.ann_build(25, 1)
    tlen = orc.txt_len

    def fwd(seq):
        return 26 * seq

    def rev(seq):
        return tlen - 2 - 26 * seq               # txt_str_to_pos("seqN:1:-")
    assert ma.ann_query(table, 25, 0) == (1, 7, [])                       # :145-150
    assert ma.ann_query(table, 25, 26) == (1, 1, [11])                    # :152-158
    assert ma.ann_query(table, 25, 23) == (0, 0, [])                      # :160-165
    want = {1: (1, 11), 2: (1, 20), 3: (1, 0), 4: (1, 24), 5: (1, 10), 6: (2, 8), 7: (2, 8)}     # :167-277
    for seq, (cnt, pos) in want.items():
        assert ma.ann_query(table, 25, fwd(seq)) == (1, cnt, [pos]), seq
        assert ma.ann_query(table, 25, rev(seq)) == (1, cnt, [24 - pos]), seq
    with pytest.raises(ma.GpumapError):
        ma.ann_query(table, 25, -1)
    with pytest.raises(ma.GpumapError):
        ma.ann_query(table, 25, tlen)
