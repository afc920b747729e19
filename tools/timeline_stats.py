#!/usr/bin/env python3
"""Concurrency statistics of a rocprofv3 --kernel-trace CSV of bench.py: python tools/timeline_stats.py <kernel_trace.csv>
(union / summed time of the MSM accumulate kernels, time with none of them running, per-kernel summed time, per steady-state step)"""
import collections
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((r['Kernel_Name'].split('(')[0].replace('void ', '').replace('bbp::', ''), int(r['Start_Timestamp']), int(r['End_Timestamp'])))
rows.sort(key=lambda r: r[1])
opens = [r for r in rows if r[0] in ('k_tr_open', 'k_open_serial')]
a, b = opens[-4][1], opens[-1][1]  # three steady-state steps
steps = 3
span = (b - a) / 1e6


def union(pred):
    iv = sorted((max(r[1], a), min(r[2], b)) for r in rows if pred(r[0]) and r[2] > a and r[1] < b)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None:
                tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce is not None:
        tot += ce - cs
    return tot / 1e6


def total(pred):
    return sum(min(r[2], b) - max(r[1], a) for r in rows if pred(r[0]) and r[2] > a and r[1] < b) / 1e6


fat = lambda n: n.startswith('k_msm_acc')
print("per step: %.1f ms" % (span / steps))
print("accumulate kernels: union %.1f ms, summed %.1f ms; none running %.1f ms" % (union(fat) / steps, total(fat) / steps, (span - union(fat)) / steps))
agg = collections.Counter()
cnt = collections.Counter()
for r in rows:
    if r[2] > a and r[1] < b:
        agg[r[0]] += (min(r[2], b) - max(r[1], a)) / 1e6
        cnt[r[0]] += 1
for k, v in agg.most_common(16):
    print("  %-20s %7.1f ms/step  %5.1f launches/step  avg %7.1f us" % (k, v / steps, cnt[k] / steps, v / cnt[k] * 1e3))
