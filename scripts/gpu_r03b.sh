mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "packet or goldens or full_size or multi_device or bands" > gpurun_out/r03b_tests.log 2>&1 || { tail -40 gpurun_out/r03b_tests.log; exit 1; }
tail -3 gpurun_out/r03b_tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r03b_bench_wide.json 2> gpurun_out/r03b_bench_wide.err || { tail -20 gpurun_out/r03b_bench_wide.err; exit 1; }
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --dbg 13=1 > gpurun_out/r03b_bench_bvh2.json 2> gpurun_out/r03b_bench_bvh2.err || { tail -20 gpurun_out/r03b_bench_bvh2.err; exit 1; }
python - <<'PY'
import json
for f in ("r03b_bench_wide", "r03b_bench_bvh2"):
    d = json.load(open("gpurun_out/%s.json" % f))
    r = d["roofline"]
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d["frame_crc"], {k: round(v, 2) for k, v in r["all_kernels_ms_per_launch"].items()}, [int(x) for x in d["device_counters"][:12]])
PY
