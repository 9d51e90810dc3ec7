# End-of-round evidence in one GPU call: smoke, rocprofv3 passes, default bench (with the fresh SQ summary), strip replay, C5 benches.
# usage (GPU box): bash scripts/gpu_final.sh <tag> <profiles dir>      e.g. bash scripts/gpu_final.sh r02 profiles/r02_final
TAG=${1:-r02}; DIR=${2:-profiles/r02_final}
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1 || exit 1
bash scripts/gpu_profile.sh $TAG > gpurun_out/prof_$TAG.log 2>&1 || exit 1
bash scripts/gpu_sq_counters.sh $TAG > gpurun_out/sq_$TAG.log 2>&1 || exit 1
bash scripts/gpu_lane_util.sh $TAG > gpurun_out/lanes_$TAG.log 2>&1 || exit 1
python scripts/collect_profile.py $TAG $DIR > /dev/null || exit 1
timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || exit 1
timeout -k 10 250 python scripts/gpu_strip_balance.py > gpurun_out/strip_balance_final.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 > gpurun_out/bench_c5_shadows.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 --shadows 0 --bounces 4 > gpurun_out/bench_c5_bounces.json 2>/dev/null || exit 1
python - <<'PY'
import json
for f in ("bench_default", "bench_c5_shadows", "bench_c5_bounces"):
    d = json.load(open("gpurun_out/%s.json" % f))
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), "build_s", d.get("build_s"))
PY
