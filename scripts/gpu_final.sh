# End-of-round evidence in one GPU call: smoke, rocprofv3 passes of the headline workload, default bench (with the fresh SQ summary), strip replay,
# the other BASELINE configs through the Renderer mirror, both multi-GPU rehearsals, C5 benches.
# usage (GPU box): bash scripts/gpu_final.sh <tag> <profiles dir>      e.g. bash scripts/gpu_final.sh r03 profiles/r03_final
TAG=${1:-r04}; DIR=${2:-profiles/r04_final}
mkdir -p gpurun_out $DIR
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || exit 1
bash scripts/gpu_profile.sh $TAG > gpurun_out/prof_$TAG.log 2>&1 || exit 1
bash scripts/gpu_sq_counters.sh $TAG > gpurun_out/sq_$TAG.log 2>&1 || exit 1
bash scripts/gpu_lane_util.sh $TAG > gpurun_out/lanes_$TAG.log 2>&1 || exit 1
python scripts/collect_profile.py $TAG $DIR > /dev/null || exit 1
timeout -k 10 400 python bench.py > $DIR/bench_default.json 2> gpurun_out/bench_default.err || exit 1
timeout -k 10 250 python scripts/gpu_strip_balance.py > $DIR/strip_balance.json 2>/dev/null || exit 1
timeout -k 10 300 python scripts/gpu_configs.py > $DIR/configs_c1_c2_c3.jsonl 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 2>/dev/null | grep '^{' > $DIR/bench_2ranks_gloo_one_gpu.json || exit 1
timeout -k 10 200 python bench.py --gpus 8 --in-library --same-device --steps 5 --warmup 2 --prelude-s 0 --verify > $DIR/bench_in_library_8_same_device.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 > $DIR/bench_c5_scene_shadows.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 --shadows 0 --bounces 4 > $DIR/bench_c5_four_bounces.json 2>/dev/null || exit 1
cp profiles/hbm_traffic.json gpurun_out/hbm_traffic.json
rm -rf gpurun_out/profiles_final && cp -r $DIR gpurun_out/profiles_final
python - $DIR <<'PY'
import json, sys
for f in ("bench_default", "bench_c5_scene_shadows", "bench_c5_four_bounces", "bench_2ranks_gloo_one_gpu", "bench_in_library_8_same_device"):
    d = json.load(open("%s/%s.json" % (sys.argv[1], f)))
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), "build_s", d.get("build_s"), (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"))
PY
