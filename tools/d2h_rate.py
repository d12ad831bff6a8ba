#!/usr/bin/env python3
"""Rate of the link for page-locked copies on this box (GB/s), D2H and H2D, alone and both at once: the figure the
pipelined bench is to be read against (the results of a chunk cross the link while the next chunk is searched)."""
import json, time, torch

def rate(fn, nbytes, reps=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t) / 1e9

out = {}
for mb in (64, 256, 1024):
    n = mb << 20
    d = torch.empty(n, dtype=torch.uint8, device="cuda"); h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d2 = torch.empty(n, dtype=torch.uint8, device="cuda"); h2 = torch.empty(n, dtype=torch.uint8).pin_memory()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def d2h():
        with torch.cuda.stream(s1): h.copy_(d, non_blocking=True)
    def h2d():
        with torch.cuda.stream(s2): d2.copy_(h2, non_blocking=True)
    def both(): d2h(); h2d()
    out["%dMiB" % mb] = {"d2h": rate(d2h, n), "h2d": rate(h2d, n), "both_each": rate(both, n)}
print(json.dumps(out))
