# round-3 first GPU call: baseline of the round (tests, default bench, fan-out rehearsal, host-copy microbench) + the C5 profiles
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r03a_tests.log 2>&1 || { tail -30 gpurun_out/r03a_tests.log; exit 1; }
tail -3 gpurun_out/r03a_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err || { tail -20 gpurun_out/r03a_bench.err; exit 1; }
timeout -k 10 200 python bench.py --gpus 2 --backend gloo --steps 5 --warmup 2 > gpurun_out/r03a_bench_gloo2.json 2> gpurun_out/r03a_bench_gloo2.err || { tail -20 gpurun_out/r03a_bench_gloo2.err; exit 1; }
timeout -k 10 120 python scripts/gpu_hostcopy.py 64 > gpurun_out/r03a_hostcopy.json 2> gpurun_out/r03a_hostcopy.err || { tail -20 gpurun_out/r03a_hostcopy.err; exit 1; }
cat gpurun_out/r03a_hostcopy.json
timeout -k 10 900 bash scripts/gpu_profile_c5.sh r03c5 || { echo c5 profile failed; tail -5 gpurun_out/*r03c5*.log; exit 1; }
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 > gpurun_out/r03a_c5_shadows.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 --shadows 0 --bounces 4 > gpurun_out/r03a_c5_bounces.json 2>/dev/null || exit 1
python - <<'PY'
import json
for f in ("r03a_bench", "r03a_bench_gloo2", "r03a_c5_shadows", "r03a_c5_bounces"):
    d = json.load(open("gpurun_out/%s.json" % f))
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), (d.get("roofline") or {}).get("kernel"), (d.get("roofline") or {}).get("frac"))
PY
