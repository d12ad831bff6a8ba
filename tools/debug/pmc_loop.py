"""Debug aid: many small batches on fresh contexts in one process (as a pytest run does), each compared with the
reference dump; prints the counters of the runs that differ.  Run under rocprofv3 --pmc (pass 8 of tools/profile_passes.sh)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mapper_amd as ma
from oracle import pyoracle as po
from tests import common

workdir = tempfile.mkdtemp(prefix="pmc_loop_")
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    for name in ("small", "repeats"):
        base = common.build_golden_index(workdir, name)
        ctx = ma.Context.from_files(base, 0)
        reads = common.load_reads(name + ".reads.txt")
        codes, offs = common.encode_reads(reads)
        for tau in (0, 1, 2):
            ref = po.HitDump.load(os.path.join(common.GOLDEN, "%s.tau%d.bin" % (name, tau)))
            for verify in (16, 4, 2, 1, 0):
                ctx.set_option("verify", verify)
                b = ctx.upload(codes, offs)
                c = b.run(tau)
                h = b.fetch()
                e = b.stats_ext()
                b.close()
                got, want = h.located(), ref.located()
                ok = np.array_equal(got, want) and np.array_equal(h.status, ref.status)
                if not ok:
                    bad += 1
                    gs = set(map(tuple, got.tolist())); ws = set(map(tuple, want.tolist()))
                    print("WRONG it", it, name, "tau", tau, "verify", verify, "n", len(got), len(want), "status ok", np.array_equal(h.status, ref.status),
                          {k: c[k] for k in ("n_hits", "list_length", "n_locs", "n_ext", "verify_steps", "jumped_steps", "loop_passes") if k in c}, e,
                          "distinct got", len(gs), "missing", len(ws - gs), "extra", len(gs - ws), "sorted", bool((np.diff(got[:, 0]) >= 0).all()), flush=True)
                else:
                    print("ok it", it, name, tau, verify, {k: c[k] for k in ("n_hits", "list_length", "n_locs", "n_ext", "verify_steps", "jumped_steps") if k in c}, flush=True)
        ctx.close()
print("bad runs:", bad)
