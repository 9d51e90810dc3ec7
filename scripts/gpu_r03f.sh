mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "reflection or c5_four or packet_primary" > gpurun_out/r03f_tests.log 2>&1 || { tail -40 gpurun_out/r03f_tests.log; exit 1; }
tail -3 gpurun_out/r03f_tests.log
C5="--no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 --shadows 0 --bounces 4 --prelude-s 0"
timeout -k 10 200 python bench.py $C5 > gpurun_out/r03f_c5b_dev.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py $C5 --host-build --dbg 5=7 > gpurun_out/r03f_c5b_host7.json 2>/dev/null || exit 1
python - <<'PY'
import json
for f in ("r03f_c5b_dev", "r03f_c5b_host7"):
    d = json.load(open("gpurun_out/%s.json" % f))
    r = d.get("roofline") or {}
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d.get("frame_crc"), {k: round(v, 2) for k, v in (r.get("all_kernels_ms_per_launch") or {}).items()}, [int(x) for x in d["device_counters"][20:24]])
PY
