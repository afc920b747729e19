#!/bin/bash
# GPU box: where do the cycles of the accumulate kernel go?  SQ / TA / TCP / TCC / GRBM counter passes (one group per pass: the slots
# per block are few) over the prove step, one heavy-stage slice (BBP_SLICES=1; counter collection serialises dispatches anyway, so every
# k_msm_acc launch has the whole GPU), aggregated per kernel by tools/pmc_table.py.
#   bash tools/pmc_issue.sh [round-tag, default r04] [extra bench.py args, e.g. "--workload verify --batch 1024"] [BBP_LIB_VARIANT]
# Every profiled command is `python3 bench.py ... --no-build` directly after `--` (no env / bash hop: the profiler's preloaded
# library has initialised the GPU).  Counter names not offered by `rocprofv3 -L` on this box are dropped from a pass, not guessed.
set -e
TAG=${1:-r04}
EXTRA=${2:-}
VAR=${3:-}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG${VAR:+_$VAR}
rm -rf "$OUT"; mkdir -p "$OUT"
python3 $REPO/__graft_entry__.py > "$OUT/build.log" 2>&1
cd /tmp && export TMPDIR=/tmp
export BBP_SLICES=1 BBP_BENCH_NO_CHECK=1
[ -n "$VAR" ] && export BBP_LIB_VARIANT=$VAR
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
B="python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --no-exclusive --steps 1 --warmup 1 $EXTRA"
pass() {  # pass NAME counter...
  local name=$1; shift
  local have=()
  for c in "$@"; do
    if grep -qw "$c" "$OUT/counters_available.txt"; then have+=("$c"); else echo "  ($c not offered here)"; fi
  done
  [ ${#have[@]} -eq 0 ] && return 0
  if rocprofv3 --kernel-trace --pmc "${have[@]}" --output-format csv -d "$OUT/$name" -o "$name" -- $B > /dev/null 2> "$OUT/$name.log"; then
    echo "pass $name ok: ${have[*]}"
  else
    echo "pass $name FAILED (see $name.log): ${have[*]}"; tail -3 "$OUT/$name.log"
  fi
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS
pass sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
pass sq3 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_INT64
pass sq4 SQ_IFETCH SQ_INSTS_VALU_INT32 SQ_BUSY_CU_CYCLES SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_REQ SQ_INSTS_SMEM SQ_CYCLES
pass ta1 TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum
pass ta2 TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum
pass ta3 TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
pass tcp2 TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum
pass tcp3 TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TCC_WRITE_REQ_sum TCP_GATE_EN2_sum
pass tcp4 TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass td1 TD_TD_BUSY_sum TD_TC_STALL_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 $REPO/tools/pmc_table.py "$OUT" > "$OUT/${TAG}_issue_breakdown.csv" 2> "$OUT/${TAG}_issue_breakdown.txt" || true
cat "$OUT/${TAG}_issue_breakdown.txt"
# keep what is judged small: the per-dispatch CSVs stay in gpurun_out
du -sh "$OUT" | tail -1
