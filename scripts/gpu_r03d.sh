mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03d_tests.log 2>&1 || { tail -40 gpurun_out/r03d_tests.log; exit 1; }
tail -3 gpurun_out/r03d_tests.log
timeout -k 10 250 python scripts/gpu_strip_balance.py > gpurun_out/r03d_strip_balance.json 2> gpurun_out/r03d_strip_balance.err || { tail gpurun_out/r03d_strip_balance.err; exit 1; }
cat gpurun_out/r03d_strip_balance.json
timeout -k 10 300 python scripts/gpu_configs.py > gpurun_out/r03d_configs.jsonl 2> gpurun_out/r03d_configs.err || { tail gpurun_out/r03d_configs.err; exit 1; }
cat gpurun_out/r03d_configs.jsonl
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 > gpurun_out/r03d_c5_shadows.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 --shadows 0 --bounces 4 > gpurun_out/r03d_c5_bounces.json 2>/dev/null || exit 1
python - <<'PY'
import json
for f in ("r03d_c5_shadows", "r03d_c5_bounces"):
    d = json.load(open("gpurun_out/%s.json" % f))
    r = d.get("roofline") or {}
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), "build_s", round(d["build_s"], 2), {k: round(v, 2) for k, v in (r.get("all_kernels_ms_per_launch") or {}).items()})
PY
