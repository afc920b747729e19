#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter, tools/profile_round.sh) per kernel:
   python tools/pmc_aggregate.py gpurun_out/prof/fetch gpurun_out/prof/write > profiles/rNN_rocprofv3_pmc_hbm.csv
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of wide reads (MI355X_MICROARCH.md, HBM section)."""
import csv, glob, os, re, sys

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
        for (counter, name), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if not name.startswith("bbp::"):
                continue
            print("%s,%s,%d,%d,%d,%s,%s,%s,%s,%s" % (counter, name, a[0], a[1], a[1] / a[0], a[2], a[3], a[4], a[5], a[6]))
