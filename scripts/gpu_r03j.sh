mkdir -p gpurun_out
C5="--tris 10000000 --extent 0.02 --no-extras --bounces 4 --shadows 0"
bash scripts/gpu_sq_counters.sh r03c5b2 $C5 > gpurun_out/sq_r03c5b2.log 2>&1 || exit 1
bash scripts/gpu_lane_util.sh r03c5b2 $C5 > gpurun_out/lanes_r03c5b2.log 2>&1 || exit 1
echo done
