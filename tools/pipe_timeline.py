#!/usr/bin/env python3
"""Timeline of the pipeline from a rocprofv3 --kernel-trace --memory-copy-trace run (csv): for the last `window`
milliseconds of the trace, every kernel and copy longer than `min_us` with start, duration and queue / stream, and
per millisecond of wall time which engines were busy.  tools/pipe_timeline.py DIR [window_ms] [min_us]"""
import csv, glob, os, sys

def main(root, window_ms=120.0, min_us=200.0):
    ev = []
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40], "q%s" % r.get("Queue_Id", "?"), "s%s" % r.get("Stream_Id", "?")))
    for f in glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "?")[:30], "-", "s%s" % r.get("Stream_Id", "?")))
    ev.sort()
    t_end = max(e[1] for e in ev)
    t0 = t_end - int(window_ms * 1e6)
    print("start_ms  dur_ms  what")
    for s, e, name, q, st in ev:
        if s >= t0 and (e - s) >= min_us * 1e3:
            print("%8.3f %7.3f  %-42s %s %s" % ((s - t0) / 1e6, (e - s) / 1e6, name, q, st))

if __name__ == "__main__":
    main(sys.argv[1], *(float(x) for x in sys.argv[2:]))
