mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r03c_tests.log 2>&1 || { tail -40 gpurun_out/r03c_tests.log; exit 1; }
tail -3 gpurun_out/r03c_tests.log
timeout -k 10 300 python bench.py --steps 10 > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err || { tail -20 gpurun_out/r03c_bench.err; exit 1; }
timeout -k 10 200 python bench.py --gpus 8 --in-library --same-device --steps 5 --warmup 2 --prelude-s 0 > gpurun_out/r03c_inlib8.json 2> gpurun_out/r03c_inlib8.err || { tail -20 gpurun_out/r03c_inlib8.err; exit 1; }
timeout -k 10 250 python scripts/gpu_strip_balance.py > gpurun_out/r03c_strip_balance.json 2>/dev/null || exit 1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03c_bench.json"))
print("bench", round(d["value"], 1), round(d["ms_per_step"], 2), d["frame_crc"], "drop_in", d["drop_in_call"], "cpu", d["cpu_baseline"]["value"])
d = json.load(open("gpurun_out/r03c_inlib8.json"))
print("inlib8", round(d["value"], 1), round(d["ms_per_step"], 2), d.get("drop_in_call"), d.get("verify"))
d = json.load(open("gpurun_out/r03c_strip_balance.json"))
print("strips", d)
PY
