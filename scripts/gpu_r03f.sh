mkdir -p gpurun_out
C5="--no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 --shadows 0 --bounces 4 --prelude-s 0"
timeout -k 10 200 python bench.py $C5 > gpurun_out/r03f_c5b_default.json 2>/dev/null || exit 1
python - <<'PY'
import json
for f in ("r03f_c5b_default",):
    d = json.load(open("gpurun_out/%s.json" % f))
    r = d.get("roofline") or {}
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d.get("frame_crc"), {k: round(v, 2) for k, v in (r.get("all_kernels_ms_per_launch") or {}).items()}, [int(x) for x in d["device_counters"][20:24]])
PY
