# GPU box: two or three host threads, large batches, sliced (default) vs the rotating unsliced path
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" python3 $REPO/tools/host_pairs.py 1024,1024 1536,1536 1024,1024,1024 1536,1536,1536 870,2202 2>&1 | grep "pair"; }
run BBP_X=0
run BBP_ROTATE_BELOW=4096 BBP_DUAL_OPEN_BELOW=4097
