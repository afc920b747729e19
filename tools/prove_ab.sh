# GPU box: prover A/B on library variants (1024 proofs per call); usage: bash tools/prove_ab.sh VARIANT...   ("" = the product library)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
B="python3 $REPO/bench.py --no-also --no-cpu-baseline --no-build --no-exclusive"
P='import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels_us"]; print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],2), "| acc", round(k["k_msm_acc"]["total_us"]/d["steps"]/1e3,1), "sort", round(k["k_msm_sort"]["total_us"]/d["steps"]/1e3,1), "fold", round(k["k_msm_fold"]["total_us"]/d["steps"]/1e3,1), "enc", round(k["k_encode"]["total_us"]/d["steps"]/1e3,1))'
for R in 1 2; do for V in "$@"; do
  if [ "$V" = "base" ]; then $B --steps 24 --warmup 4 | python3 -c "$P" "round $R base";
  else BBP_LIB_VARIANT=$V $B --steps 24 --warmup 4 | python3 -c "$P" "round $R $V"; fi
done; done
