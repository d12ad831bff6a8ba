"""Shared helpers of the tests (index building through the product's host code, read loading)."""
import os

import numpy as np

import mapper_amd as ma
from mapper_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_TABLE = np.full(256, 4, dtype=np.uint8)
for _i, _pair in enumerate(("Aa", "Cc", "Gg", "Tt", "Nn")):
    for _c in _pair:
        _TABLE[ord(_c)] = _i


def encode_reads(reads):
    """sym_str_index (src/index_sym.c:293-319) restated with the table of sym_new_dna: test-side only."""
    codes = [_TABLE[np.frombuffer(r.encode(), dtype=np.uint8)] for r in reads]
    offs = np.concatenate(([0], np.cumsum([len(c) for c in codes]))).astype(np.int64)
    flat = np.concatenate(codes) if len(codes) and offs[-1] > 0 else np.zeros(0, np.uint8)
    return flat.astype(np.uint8), offs


def load_reads(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return f.read().split("\n")[:-1]


def build_golden_index(workdir, name):
    """Index of tests/golden/<name>.fa built by the product's host builder; returns the base path."""
    base = os.path.join(workdir, name)
    if not os.path.exists(base + ".bwt"):
        ix = ma.HostIndex.build(os.path.join(GOLDEN, name + ".fa"), base)
        ix.close()
    return base


def synth_index(workdir, tag, total_len, seed, n_seqs=1, repeat_rich=False, gc=0.5):
    base = os.path.join(workdir, tag)
    seqs, names = synth.make_genome(total_len, seed, n_seqs=n_seqs, gc=gc, repeat_rich=repeat_rich)
    if not os.path.exists(base + ".bwt"):
        ix = ma.HostIndex.build_codes(seqs, names, base)
        ix.close()
    return base, seqs
