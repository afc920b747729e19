#!/usr/bin/env python3
"""Concurrency analysis of a rocprofv3 --kernel-trace CSV: python tools/trace_overlap.py <kernel_trace.csv> [steps]
Looks at the last `steps` prove steps (delimited by k_assemble pairs) and reports how long 0/1/2 MSM kernels were running."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("bbp::", ""), r["Stream_Id"]) for r in rows)
ass = [e for e in ev if e[2] == "k_assemble"]
per_step = 2
t1 = ass[-1][1]
t0 = ass[-1 - per_step * steps][1]
sel = [e for e in ev if e[1] > t0 and e[0] < t1]
print("window %.1f ms for %d steps -> %.1f ms/step" % ((t1 - t0) / 1e6, steps, (t1 - t0) / 1e6 / steps))
pts = []
for s, e, n, q in sel:
    if n == "k_msm":
        pts += [(max(s, t0), 1), (min(e, t1), -1)]
pts.sort()
cur, last, dur = 0, t0, {}
for t, d in pts:
    dur[cur] = dur.get(cur, 0) + t - last
    cur += d
    last = t
dur[cur] = dur.get(cur, 0) + t1 - last
print("MSM concurrency (ms/step):", {k: round(v / 1e6 / steps, 1) for k, v in sorted(dur.items())})
# what runs while no MSM is running
idle = []
cur, last = 0, t0
gaps = []
for t, d in pts:
    if cur == 0 and t > last:
        gaps.append((last, t))
    cur += d
    last = t
if last < t1 and cur == 0: gaps.append((last, t1))
busy = {}
for gs, ge_ in gaps:
    for s, e, n, q in sel:
        if n != "k_msm":
            o = min(e, ge_) - max(s, gs)
            if o > 0: busy[n] = busy.get(n, 0) + o
print("kernels running during zero-MSM time (ms/step):", {k: round(v / 1e6 / steps, 2) for k, v in sorted(busy.items(), key=lambda kv: -kv[1])[:8]})
print("largest zero-MSM gaps (ms):", [round((b - a) / 1e6, 2) for a, b in sorted(gaps, key=lambda g: g[0] - g[1])[:8]])
