import sys, time, os
import numpy as np, torch
sys.path.insert(0, '.')
import mapper_amd as ma
from mapper_amd import synth_device
L = int(sys.argv[1]); nseq = int(sys.argv[2])
g, lens, names = synth_device.make_genome(L, 4, n_seqs=nseq)
torch.cuda.synchronize(); torch.cuda.empty_cache()
t = time.time()
ctx = ma.Context.build_device(lens, 0, keep_reference=True, device_ptr=g.data_ptr())
print("build s", time.time() - t, flush=True)
ref = ctx.reference_arrays()
txt = ref["text"]
print("n", ref["txt_len"], "text N", int((txt == 4).sum()), "text $", int((txt == 5).sum()), "c", ref["c"], flush=True)
gh = g.cpu().numpy()
tl0 = len(txt) // 2
# forward half check
off = 0; pos = 0; ok = True
for ln in lens:
    ok &= bool(np.array_equal(txt[pos:pos + ln], gh[off:off + ln])) and txt[pos + ln] == 5
    off += ln; pos += ln + 1
comp = np.array([3, 2, 1, 0, 4, 5], np.uint8)
ok2 = np.array_equal(txt[tl0:2 * tl0 - 1], comp[txt[:tl0 - 1][::-1]])
print("fwd ok", ok, "rc ok", ok2, flush=True)
