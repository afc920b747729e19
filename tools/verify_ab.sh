# GPU box: verifier A/B on library variants (1024 verifications per call, exclusive single-lane timing incl. the accumulate launch);
# usage: bash tools/verify_ab.sh VARIANT...   ("base" = the product library).  Wrong-result variants need BBP_BENCH_NO_CHECK=1.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
B="python3 $REPO/bench.py --workload verify --batch 1024 --no-also --no-cpu-baseline --no-build"
P='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels_us"]; e=d["roofline"].get("exclusive",{}); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), "| acc", round(k["k_msm_acc"]["total_us"]/d["steps"]/1e3,2), "| exclusive acc ms", round(e.get("dominant_ms_per_step",0),2), "step", round(e.get("ms_per_step",0),2), "alu", round(e.get("alu_frac",0),3))'
for R in 1 2; do for V in "$@"; do
  if [ "$V" = "base" ]; then $B --steps 48 --warmup 8 | python3 -c "$P" "round $R base";
  else BBP_LIB_VARIANT=$V $B --steps 48 --warmup 8 | python3 -c "$P" "round $R $V"; fi
done; done
