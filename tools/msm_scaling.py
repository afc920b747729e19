#!/usr/bin/env python3
"""Accumulate-kernel time per term against MSM width and table footprint (1024 MSMs per launch, random full-size scalars).
Run under rocprofv3 --kernel-trace (tools/msm_scaling.sh); `parse TRACE.csv` prints the table.  Configs run in a fixed order, REPS
launches each, so the k-th group of REPS accumulate launches in the trace belongs to the k-th config."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CONFIGS = [(0, m) for m in (256, 512, 733, 1024, 1466, 2048)] + [(1, m) for m in (512, 1024, 1466, 2048)]
REPS, B = 5, 1024


def n_terms(layout, m):
    return 1 + (2 * m if layout == 0 else m)


if len(sys.argv) > 2 and sys.argv[1] == "parse":
    import csv
    rows = [r for r in csv.DictReader(open(sys.argv[2]))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for kern in ("k_msm_acc", "k_msm_sort", "k_msm_fold"):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if kern in r["Kernel_Name"]]
        assert len(d) == len(CONFIGS) * REPS, (kern, len(d))
        for i, (layout, m) in enumerate(CONFIGS):
            g = sorted(d[i * REPS:(i + 1) * REPS])[1:-1]
            us = sum(g) / len(g)
            n = n_terms(layout, m)
            print("%-11s layout %s m %4d terms %4d table %5.1f MB: %8.1f us per launch, %6.3f ns per term and MSM"
                  % (kern, "G+H" if layout == 0 else "G  ", m, n, n * 256 * 128 / 1e6, us, us * 1e3 / n / B))
    raise SystemExit(0)

import torch
import dusk_blindbidproof_amd as bbp
from bench import synth_scalars_device
ctx = bbp.Context(0)
dev = torch.device("cuda:0")
out = torch.zeros((B, 32), dtype=torch.uint8, device=dev)
for i, (layout, m) in enumerate(CONFIGS):
    n = n_terms(layout, m)
    s = synth_scalars_device(torch, B, n, 100 + i, dev)
    torch.cuda.synchronize()
    for _ in range(REPS):
        ctx.msm_batch_dev(B, n, s.data_ptr(), layout, out.data_ptr(), None)
    torch.cuda.synchronize()
ctx.close()
