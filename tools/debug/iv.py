import json, sys, statistics
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
iv = d["step_intervals_ms"]
print(sys.argv[1], "value %.1f M" % (d["value"] / 1e6), "ms/step %.2f" % d["ms_per_step"], "first %.1f median %.2f max-after-first %.1f" % (iv[0], statistics.median(iv[1:-1]), max(iv[1:])))
