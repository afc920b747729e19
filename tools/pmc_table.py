#!/usr/bin/env python3
"""Pivot the counter passes of tools/pmc_issue.sh into one table: rows = kernels, columns = counters (summed over the kernel's
dispatches, and over the dimension rows rocprofv3 emits per dispatch).

    python tools/pmc_table.py gpurun_out/pmc_r04 > profiles/r04_msm_acc_issue_breakdown.csv   (derived shares -> stderr)

SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave (MI355X_MICROARCH.md, cycle constants), so ratios between
them are what is read; SQ_INSTS_* count wave-instructions."""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
only = sys.argv[2] if len(sys.argv) > 2 else None  # optional kernel-name filter for the derived block
tab = defaultdict(lambda: defaultdict(float))
disp = defaultdict(lambda: defaultdict(set))
meta = {}
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
        if not name.startswith("bbp::"):
            continue
        # the accumulate kernel is launched with different grids by prove (1024 / 2048 MSMs) and by the verifier: keep them apart
        key = name + " grid=" + r.get("Grid_Size", "?")
        c = r["Counter_Name"]
        tab[key][c] += float(r["Counter_Value"])
        disp[key][c].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
        meta[key] = (r.get("Workgroup_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("LDS_Block_Size", ""), r.get("Scratch_Size", ""))

counters = sorted({c for k in tab for c in tab[k]})
w = csv.writer(sys.stdout)
w.writerow(["kernel", "dispatches", "workgroup", "vgpr", "agpr", "lds", "scratch"] + counters)
def total(k):
    return tab[k].get("SQ_WAVE_CYCLES", 0.0) or tab[k].get("GRBM_GUI_ACTIVE", 0.0)
for k in sorted(tab, key=lambda k: -total(k)):
    nd = max((len(s) for s in disp[k].values()), default=0)
    w.writerow([k, nd] + list(meta[k]) + ["%d" % tab[k][c] if c in tab[k] else "" for c in counters])

def share(k, num, den="SQ_WAVE_CYCLES"):
    t = tab[k]
    return t[num] / t[den] if t.get(den) and num in t else None

e = sys.stderr
for k in sorted(tab, key=lambda k: -total(k)):
    if "k_msm_acc" not in k and not (only and only in k):
        continue
    t = tab[k]
    nd = max((len(s) for s in disp[k].values()), default=0)
    print("== %s (%d dispatches per pass)" % (k, nd), file=e)
    def line(label, v, fmt="%.3f"):
        if v is not None:
            print("   %-64s " % label + fmt % v, file=e)
    line("wave-cycles spent with an instruction issuing   ACTIVE_INST_ANY / WAVE_CYCLES", share(k, "SQ_ACTIVE_INST_ANY"))
    line("  of which VALU                                 ACTIVE_INST_VALU / WAVE_CYCLES", share(k, "SQ_ACTIVE_INST_VALU"))
    line("  of which VMEM issue                           ACTIVE_INST_VMEM / WAVE_CYCLES", share(k, "SQ_ACTIVE_INST_VMEM"))
    line("  of which scalar                               ACTIVE_INST_SCA / WAVE_CYCLES", share(k, "SQ_ACTIVE_INST_SCA"))
    line("  of which LDS                                  ACTIVE_INST_LDS / WAVE_CYCLES", share(k, "SQ_ACTIVE_INST_LDS"))
    line("wave parked on s_waitcnt / barrier              WAIT_ANY / WAVE_CYCLES", share(k, "SQ_WAIT_ANY"))
    line("wave ready, instruction not issued (arbitration, pipe, dependency)  WAIT_INST_ANY / WAVE_CYCLES", share(k, "SQ_WAIT_INST_ANY"))
    line("VMEM read issue cycles                          INST_CYCLES_VMEM_RD / WAVE_CYCLES", share(k, "SQ_INST_CYCLES_VMEM_RD"))
    line("VMEM write issue cycles                         INST_CYCLES_VMEM_WR / WAVE_CYCLES", share(k, "SQ_INST_CYCLES_VMEM_WR"))
    line("TA address FIFO full                            VMEM_TA_ADDR_FIFO_FULL / WAVE_CYCLES", share(k, "SQ_VMEM_TA_ADDR_FIFO_FULL"))
    line("TA command FIFO full                            VMEM_TA_CMD_FIFO_FULL / WAVE_CYCLES", share(k, "SQ_VMEM_TA_CMD_FIFO_FULL"))
    line("TA write-data FIFO full                         VMEM_WR_TA_DATA_FIFO_FULL / WAVE_CYCLES", share(k, "SQ_VMEM_WR_TA_DATA_FIFO_FULL"))
    if t.get("SQ_WAVES"):
        line("VALU instructions per wave", t.get("SQ_INSTS_VALU", 0) / t["SQ_WAVES"], "%.0f")
        line("VMEM reads per wave", t.get("SQ_INSTS_VMEM_RD", 0) / t["SQ_WAVES"], "%.0f")
        line("VMEM writes per wave", t.get("SQ_INSTS_VMEM_WR", 0) / t["SQ_WAVES"], "%.0f")
        line("wave-cycles (quad-cycles) per wave", t.get("SQ_WAVE_CYCLES", 0) / t["SQ_WAVES"], "%.0f")
        if t.get("SQ_INSTS_VALU"):
            line("quad-cycles of wave lifetime per VALU instruction", t.get("SQ_WAVE_CYCLES", 0) / t["SQ_INSTS_VALU"], "%.3f")
    if t.get("SQ_THREAD_CYCLES_VALU") and t.get("SQ_ACTIVE_INST_VALU"):
        line("active lanes per VALU instruction (of 64)       THREAD_CYCLES_VALU / ACTIVE_INST_VALU", t["SQ_THREAD_CYCLES_VALU"] / t["SQ_ACTIVE_INST_VALU"], "%.1f")
    if t.get("SQ_BUSY_CYCLES"):
        line("SQ busy share of GRBM_GUI_ACTIVE-equivalent: see csv", None)
    if t.get("TA_TA_BUSY_sum") and t.get("GRBM_GUI_ACTIVE"):
        # TA_TA_BUSY_sum sums the busy cycles of every TA (one per CU); GRBM_GUI_ACTIVE sums the 8 XCDs' active cycles
        line("TA busy, average over 256 TAs                   TA_TA_BUSY_sum / 256 / (GRBM_GUI_ACTIVE / 8)", t["TA_TA_BUSY_sum"] / 256 / (t["GRBM_GUI_ACTIVE"] / 8))
    for c in ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_ADDR_STALLED_BY_TD_CYCLES_sum", "TD_TD_BUSY_sum", "TD_TC_STALL_sum",
              "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum",
              "TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_TCP_TA_ADDR_STALL_CYCLES_sum", "TCP_LFIFO_STALL_CYCLES_sum"):
        if t.get(c) is not None and c in t and t.get("GRBM_GUI_ACTIVE"):
            line("%-47s / 256 / (GRBM_GUI_ACTIVE / 8)" % c, t[c] / 256 / (t["GRBM_GUI_ACTIVE"] / 8))
    if t.get("TCP_TOTAL_CACHE_ACCESSES_sum") and t.get("TCP_TCC_READ_REQ_sum"):
        line("vector L1: requests to L2 per cache access      TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES", t["TCP_TCC_READ_REQ_sum"] / t["TCP_TOTAL_CACHE_ACCESSES_sum"])
    if t.get("TCC_HIT_sum") is not None and (t.get("TCC_HIT_sum", 0) + t.get("TCC_MISS_sum", 0)) > 0:
        line("L2 hit rate                                     TCC_HIT / (TCC_HIT + TCC_MISS)", t["TCC_HIT_sum"] / (t["TCC_HIT_sum"] + t["TCC_MISS_sum"]))
    if t.get("TCP_TCP_LATENCY_sum") and t.get("TCP_TOTAL_ACCESSES_sum"):
        line("average vector-L1 latency, cycles per access    TCP_TCP_LATENCY / TCP_TOTAL_ACCESSES", t["TCP_TCP_LATENCY_sum"] / t["TCP_TOTAL_ACCESSES_sum"], "%.0f")
    if t.get("TCP_TCC_READ_REQ_LATENCY_sum") and t.get("TCP_TCC_READ_REQ_sum"):
        line("average L1->L2 read latency, cycles per request TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ", t["TCP_TCC_READ_REQ_LATENCY_sum"] / t["TCP_TCC_READ_REQ_sum"], "%.0f")
