# SQ issue/wait breakdown per kernel (separate PMC pass, kernel-trace only).  usage: bash scripts/gpu_sq_counters.sh <tag> [bench args]
set -x
TAG=${1:-r01}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/sq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --prelude-s 0 --no-cpu-baseline --no-split $@"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/pmc_sq -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq.json 2> $OUT/sq.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq2 -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq2.json 2> $OUT/sq2.err
for f in $(find $OUT -name "*counter_collection.csv"); do echo == $f; head -3 $f; wc -l $f; done
tail -n 5 $OUT/*.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq3 -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_sq3.json 2> $OUT/sq3.err
tail -n 3 $OUT/sq3.err
