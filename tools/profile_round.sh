#!/bin/bash
# GPU box: the rocprofv3 passes behind profiles/ (kernel stats, then FETCH_SIZE and WRITE_SIZE in their own runs), for the three
# workloads whose bench line carries `roofline.traffic`: prove (configs[2]), verify 8192 (configs[3] shard), msm (configs[1]).
#   bash tools/profile_round.sh [round-tag, default r03]
# Every profiled command is `python3 bench.py ... --no-build`: native artefacts are (re)built HERE, before any profiler starts,
# so the profiled process never spawns a compiler (its children would inherit the profiler's preloaded library, which has
# initialised the GPU -- the exec hop this pool forbids).  Writes gpurun_out/prof/<TAG>_*.csv|json and UPDATES profiles/traffic.json
# in the snapshot's gpurun_out/prof/ (copy what is to be judged into profiles/ afterwards).
set -e
TAG=${1:-r04}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof
rm -rf "$OUT"; mkdir -p "$OUT"
python3 $REPO/__graft_entry__.py > "$OUT/build.log" 2>&1
cp $REPO/profiles/traffic.json "$OUT/traffic.json" 2>/dev/null || echo "{}" > "$OUT/traffic.json"
cd /tmp && export TMPDIR=/tmp
# no bbp_reserve before the clock: its thirteen dummy prove batches would be counted into every per-launch average and into the traffic
export BBP_BENCH_NO_RESERVE=1
B="python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --no-exclusive"
PMC_STEPS=2   # every counter pass runs --steps 1 --warmup 1: two steps of the workload execute
stats() { rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$1" -o "$1" -- $B "${@:3}" > "$OUT/$2" 2> "$OUT/$1.log"; echo "$1 stats pass done"; }
pmc() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$OUT/$1" -o "$1" -- $B "${@:3}" --steps 1 --warmup 1 > /dev/null 2> "$OUT/$1.log"; echo "$1 $2 pass done"; }
# 1. the shipped configuration (three heavy-stage slices: launches of different slices overlap, per-launch times are stretched)
stats stats bench_under_rocprof.json --steps 5 --warmup 2
# 2. EXCLUSIVE pass: one slice, so no two MSM launches overlap and a launch covers the whole 1024-proof batch
export BBP_SLICES=1
stats excl bench_exclusive_under_rocprof.json --steps 5 --warmup 2
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
python3 $REPO/tools/pmc_aggregate.py "$OUT/fetch" "$OUT/write" --traffic-json "$OUT/traffic.json" prove_b1024_n8 $PMC_STEPS \
  "profiles/${TAG}_rocprofv3_pmc_hbm_exclusive.csv (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --steps 1 --warmup 1, BBP_SLICES=1, B=1024 N=8)" \
  > "$OUT/${TAG}_rocprofv3_pmc_hbm_exclusive.csv"
unset BBP_SLICES
# 3. configs[3] shard: 8192 verifications per step (its proofs are made by one prove step first: --only-grid keeps that step's launches out)
stats verify bench_verify8192_under_rocprof.json --workload verify --batch 8192 --steps 3 --warmup 1
pmc vfetch FETCH_SIZE --workload verify --batch 8192
pmc vwrite WRITE_SIZE --workload verify --batch 8192
python3 $REPO/tools/pmc_aggregate.py "$OUT/vfetch" "$OUT/vwrite" --only-grid $((8192 * 256)) --traffic-json "$OUT/traffic.json" verify_b8192_n8 $PMC_STEPS \
  "profiles/${TAG}_verify8192_rocprofv3_pmc_hbm.csv (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --workload verify --batch 8192 --steps 1 --warmup 1; only the 8192-workgroup accumulate launches are counted)" \
  > "$OUT/${TAG}_verify8192_rocprofv3_pmc_hbm.csv"
# 3b. the same at 1024 verifications per call (what the default bench line's also.verify runs)
pmc v1fetch FETCH_SIZE --workload verify --batch 1024
pmc v1write WRITE_SIZE --workload verify --batch 1024
python3 $REPO/tools/pmc_aggregate.py "$OUT/v1fetch" "$OUT/v1write" --only-grid $((1024 * 256)) --traffic-json "$OUT/traffic.json" verify_b1024_n8 $PMC_STEPS \
  "profiles/${TAG}_verify1024_rocprofv3_pmc_hbm.csv (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --workload verify --batch 1024 --steps 1 --warmup 1; only the 1024-workgroup accumulate launches are counted)" \
  > "$OUT/${TAG}_verify1024_rocprofv3_pmc_hbm.csv"
# 4. configs[1]: commitment MSMs only
stats msm bench_msm_under_rocprof.json --workload msm --steps 5 --warmup 2
pmc mfetch FETCH_SIZE --workload msm
pmc mwrite WRITE_SIZE --workload msm
python3 $REPO/tools/pmc_aggregate.py "$OUT/mfetch" "$OUT/mwrite" --traffic-json "$OUT/traffic.json" msm_b1024_n8 $PMC_STEPS \
  "profiles/${TAG}_msm_rocprofv3_pmc_hbm.csv (rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --workload msm --steps 1 --warmup 1, B=1024 N=8)" \
  > "$OUT/${TAG}_msm_rocprofv3_pmc_hbm.csv"
for d in stats excl verify msm; do f=$(find "$OUT/$d" -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" "$OUT/${TAG}_${d}_kernel_stats.csv"; done
ls -la "$OUT" | head -40
