#!/usr/bin/env python3
"""Where a mid-size back-to-back prove loop spends its time: python tools/midsize_trace.py <kernel_trace.csv> [calls]
Takes the window of the last `calls` k_witness_head launches (one per call) and prints, per call: wall time, per-queue busy time,
time during which SOME accumulate kernel runs, and the summed duration and launch count of every kernel."""
import collections, csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Queue_Id']), int(r['Start_Timestamp']), int(r['End_Timestamp']),
                 r['Kernel_Name'].split('(')[0].replace('void ', '').replace('bbp::', '')))
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 9
heads = sorted(r[1] for r in rows if r[3] == 'k_witness_head')
a, b = heads[-calls - 1], heads[-1]
win = [r for r in rows if a <= r[1] < b]
print("window: %d calls, %.2f ms per call" % (calls, (b - a) / calls / 1e6))


def union(iv):
    iv.sort()
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


byq = collections.defaultdict(list)
for r in win:
    byq[r[0]].append((r[1], r[2]))
for q, iv in sorted(byq.items()):
    print("  queue %2d: busy %.0f %% (%d launches)" % (q, 100.0 * union(iv) / (b - a), len(iv)))
print("  some k_msm_acc running: %.0f %%" % (100.0 * union([(r[1], r[2]) for r in win if r[3].startswith('k_msm_acc')]) / (b - a)))
print("  anything running:       %.0f %%" % (100.0 * union([(r[1], r[2]) for r in win]) / (b - a)))
d, n = collections.Counter(), collections.Counter()
for r in win:
    d[r[3]] += r[2] - r[1]
    n[r[3]] += 1
for k, v in d.most_common(24):
    print("  %-26s %5.1f launches/call  %8.3f ms/call (summed)  %7.1f us each" % (k, n[k] / calls, v / calls / 1e6, v / n[k] / 1e3))
