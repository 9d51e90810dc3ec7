# average active lanes per VALU instruction: SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (separate PMC pass)
# usage: bash scripts/gpu_lane_util.sh <tag> [bench args]
set -x
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/lanes_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/pmc -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --prelude-s 0 --no-cpu-baseline --no-split "$@" > $OUT/bench.json 2> $OUT/err.txt
tail -2 $OUT/err.txt
