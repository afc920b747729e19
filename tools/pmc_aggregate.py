#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter, tools/profile_round.sh) per kernel:
   python tools/pmc_aggregate.py gpurun_out/prof/fetch gpurun_out/prof/write > profiles/rNN_rocprofv3_pmc_hbm.csv
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of wide reads (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, json, os, re, sys

# --traffic-json OUT KEY STEPS SOURCE: also write the dominant kernel's measured bytes per step into OUT (profiles/traffic.json):
# 2 x FETCH_SIZE + WRITE_SIZE of every k_msm_acc dispatch, divided by the number of bench steps the counter runs executed
tj = None
if "--traffic-json" in sys.argv:
    i = sys.argv.index("--traffic-json")
    tj = sys.argv[i + 1:i + 5]
    del sys.argv[i:i + 5]
# --only-grid N: count a k_msm_acc dispatch towards the traffic figure only if its Grid_Size is N (threads): the verify workload
# makes its proofs with one prove step first, whose accumulate launches (1024 or 2048 workgroups) must not be charged to the
# verifier's (one workgroup per verification: 8192 x 256 threads)
only_grid = None
if "--only-grid" in sys.argv:
    i = sys.argv.index("--only-grid")
    only_grid = sys.argv[i + 1]
    del sys.argv[i:i + 2]
acc = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
acc_n = {"FETCH_SIZE": 0, "WRITE_SIZE": 0}

print("counter,kernel,dispatches,total_KiB,avg_KiB_per_dispatch,grid,workgroup,vgpr,lds_bytes,scratch_bytes")
for d in sys.argv[1:]:
    for path in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
        agg = {}
        for r in csv.DictReader(open(path)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            key = (r["Counter_Name"], name)
            a = agg.setdefault(key, [0, 0.0, r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("VGPR_Count", ""),
                                     r.get("LDS_Block_Size", ""), r.get("Scratch_Size", "")])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            if name.startswith("bbp::k_msm_acc") and r["Counter_Name"] in acc and (only_grid is None or r.get("Grid_Size", "") == only_grid):
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
                acc_n[r["Counter_Name"]] += 1
        for (counter, name), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if not name.startswith("bbp::"):
                continue
            print("%s,%s,%d,%d,%d,%s,%s,%s,%s,%s" % (counter, name, a[0], a[1], a[1] / a[0], a[2], a[3], a[4], a[5], a[6]))

if tj:
    out, key, steps, source = tj[0], tj[1], int(tj[2]), tj[3]
    try:
        cur = json.load(open(out))
    except Exception:
        cur = {}
    raw = (acc["FETCH_SIZE"] + acc["WRITE_SIZE"]) * 1024 / steps
    cur[key] = {"bytes_per_step": int((2 * acc["FETCH_SIZE"] + acc["WRITE_SIZE"]) * 1024 / steps), "raw_counter_bytes_per_step": int(raw),
                "fetch_KiB_per_step": int(acc["FETCH_SIZE"] / steps), "write_KiB_per_step": int(acc["WRITE_SIZE"] / steps),
                "kernel": "k_msm_acc (both instances)", "source": source, "measured_on": __import__("time").strftime("%Y-%m-%d"), "dispatches_counted": acc_n, "steps": steps,
                "note": "bytes_per_step = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction for wide reads, MI355X_MICROARCH.md HBM section); the row "
                        "gathers are 8 x 16 B per lane, a width the guide calls uncalibrated: the true fabric-side figure lies between "
                        "raw_counter_bytes_per_step and bytes_per_step; Infinity-Cache hits are counted"}
    json.dump(cur, open(out, "w"), indent=1)
