# rocprofv3 evidence for the bench workload: kernel trace + stats, then HBM counters in separate passes.
# usage (on the GPU box): bash scripts/gpu_profile.sh <tag> [bench args...]
set -x
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --prelude-s 0 --no-cpu-baseline --no-split $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_fetch.json 2> $OUT/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_write.json 2> $OUT/write.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2 -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_l2.json 2> $OUT/l2.err
find $OUT -name "*.csv" | head -50
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo == $f; cat $f; done
for f in $(find $OUT -name "*counter_collection.csv"); do echo == $f; head -3 $f; wc -l $f; done
tail -n 5 $OUT/*.err
