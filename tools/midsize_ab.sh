# GPU box: back-to-back mid-size prove calls under the rng-chain variants; bash tools/midsize_ab.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" python3 $REPO/tools/midsize.py 64 128 256 384 512 768 1024 2>&1 | grep "B="; }
run BBP_RNG_DPP=1
run BBP_RNG_DPP=0
run BBP_RNG_DPP=1 BBP_RNG_COOP_BELOW=1100
run BBP_RNG_DPP=1 BBP_RNG_BLOCK=64
run BBP_RNG_DPP=1 BBP_RNG_BLOCK=256
