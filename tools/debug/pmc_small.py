"""Debug aid: the scenario of tests/test_gpu_parity.py::test_verify_path_matches_reference_dump that fails under
rocprofv3 --pmc (pass 8 of tools/profile_passes.sh): which option setting, which part of the result."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mapper_amd as ma
from oracle import pyoracle as po
from tests import common

workdir = tempfile.mkdtemp(prefix="pmc_small_")
name = "small"
base = common.build_golden_index(workdir, name)
ctx = ma.Context.from_files(base, 0)
reads = common.load_reads(name + ".reads.txt")
codes, offs = common.encode_reads(reads)
for o in sys.argv[1:]:
    ctx.set_option(o.split("=")[0], int(o.split("=")[1]))
for tau in (0, 1, 2, 3):
    ref = po.HitDump.load(os.path.join(common.GOLDEN, "%s.tau%d.bin" % (name, tau)))
    for verify, seed in ((16, 1), (4, 1), (4, 0), (2, 1), (1, 1), (1, 0), (0, 1)):
        ctx.set_option("verify", verify)
        ctx.set_option("seed_candidates", seed)
        for rep in range(2):
            b = ctx.upload(codes, offs)
            c = b.run(tau)
            h = b.fetch()
            b.close()
            got, want = h.located(), ref.located()
            ok = np.array_equal(got, want)
            print("tau", tau, "verify", verify, "seed", seed, "rep", rep, "ok" if ok else "WRONG", "n", len(got), len(want),
                  {k: c[k] for k in ("n_hits", "list_length", "n_locs", "verify_steps") if k in c}, flush=True)
            if not ok:
                gs = set(map(tuple, got.tolist())); ws = set(map(tuple, want.tolist()))
                print("   distinct got", len(gs), "want", len(ws), "missing", len(ws - gs), "extra", len(gs - ws), "first extra", sorted(gs - ws)[:5], "first rows", got[:6].tolist())
ctx.close()
