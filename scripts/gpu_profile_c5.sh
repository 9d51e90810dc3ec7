# rocprofv3 evidence for the C5 regime (10 M triangles, extent 0.02, 4096^2): kernel stats + FETCH/WRITE + SQ passes of both C5 lines.
# usage (GPU box): bash scripts/gpu_profile_c5.sh <tag> [prefix]   -> gpurun_out/prof_<tag>_{bounces,shadows}, sq_<tag>_{bounces,shadows},
#                   summaries in profiles/r04_c5/<prefix>{bounces,shadows} (prefix "" = the round-start profile, "final_" = the round's end)
TAG=${1:-r03c5}; PRE=${2:-}
C5="--tris 10000000 --extent 0.02 --no-extras"
bash scripts/gpu_profile.sh ${TAG}_bounces $C5 --bounces 4 --shadows 0 > gpurun_out/prof_${TAG}_bounces.log 2>&1 || exit 1
bash scripts/gpu_sq_counters.sh ${TAG}_bounces $C5 --bounces 4 --shadows 0 > gpurun_out/sq_${TAG}_bounces.log 2>&1 || exit 1
bash scripts/gpu_lane_util.sh ${TAG}_bounces $C5 --bounces 4 --shadows 0 > gpurun_out/lanes_${TAG}_bounces.log 2>&1 || exit 1
bash scripts/gpu_profile.sh ${TAG}_shadows $C5 --shadows 100 > gpurun_out/prof_${TAG}_shadows.log 2>&1 || exit 1
bash scripts/gpu_sq_counters.sh ${TAG}_shadows $C5 --shadows 100 > gpurun_out/sq_${TAG}_shadows.log 2>&1 || exit 1
bash scripts/gpu_lane_util.sh ${TAG}_shadows $C5 --shadows 100 > gpurun_out/lanes_${TAG}_shadows.log 2>&1 || exit 1
python scripts/collect_profile.py ${TAG}_bounces profiles/r04_c5/${PRE}bounces bvh_10000000_4096_0_b4 > gpurun_out/collect_${TAG}_bounces.log 2>&1 || exit 1
python scripts/collect_profile.py ${TAG}_shadows profiles/r04_c5/${PRE}shadows bvh_10000000_4096_100 > gpurun_out/collect_${TAG}_shadows.log 2>&1 || exit 1
cp -r profiles/r04_c5 gpurun_out/profiles_r04_c5; cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
