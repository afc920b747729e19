#!/usr/bin/env python3
"""Per-kernel dispatch gaps in a rocprofv3 --kernel-trace CSV of bench.py: python tools/timeline_gaps.py <kernel_trace.csv>
For every kernel: the time between the end of the previous kernel on the SAME hardware queue and its own start (what it waited for
besides its stream predecessor: free registers / LDS / wave slots, or the host), and its duration; steady-state steps only."""
import collections, csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Queue_Id']), int(r['Start_Timestamp']), int(r['End_Timestamp']),
                     r['Kernel_Name'].split('(')[0].replace('void ', '').replace('bbp::', ''), int(r['VGPR_Count']) + int(r['Accum_VGPR_Count'])))
opens = sorted(r[1] for r in rows if r[3] in ('k_open_serial',))
a, b = opens[-4], opens[-1]
steps = 3
byq = collections.defaultdict(list)
for r in rows:
    byq[r[0]].append(r)
gap, dur, cnt, vg = collections.Counter(), collections.Counter(), collections.Counter(), {}
for q, lst in byq.items():
    lst.sort(key=lambda r: r[1])
    for prev, cur in zip(lst, lst[1:]):
        if cur[1] < a or cur[1] > b:
            continue
        g = cur[1] - prev[2]
        if g > 5e6:  # idle queue
            continue
        gap[cur[3]] += max(g, 0)
        dur[cur[3]] += cur[2] - cur[1]
        cnt[cur[3]] += 1
        vg[cur[3]] = cur[4]
print("per step %.2f ms; queues: %s" % ((b - a) / steps / 1e6, {q: len(l) for q, l in byq.items()}))
print("%-22s %6s %9s %10s %10s %10s %10s" % ("kernel", "VGPRs", "n/step", "gap us", "dur us", "gap ms/st", "dur ms/st"))
for k, _ in sorted(dur.items(), key=lambda kv: -(kv[1] + gap[kv[0]])):
    n = cnt[k]
    print("%-22s %6d %9.1f %10.1f %10.1f %10.2f %10.2f" % (k[:22], vg[k], n / steps, gap[k] / n / 1e3, dur[k] / n / 1e3, gap[k] / steps / 1e6, dur[k] / steps / 1e6))
