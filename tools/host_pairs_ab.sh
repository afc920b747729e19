# GPU box: unequal batch pairs from two host threads; where the small-batch path (rotation, dual openings) should end
REPO=${GRAFT_REPO_ROOT:-/root/repo}
run() { echo "== $*"; env "$@" python3 $REPO/tools/host_pairs.py 300,2700 600,2400 870,2202 1022,2050 1093,1979 1536,1536 2>&1 | grep "pair"; }
run BBP_X=0
run BBP_ROTATE_MIXED_FROM=0
run BBP_ROTATE_MIXED_FROM=256
