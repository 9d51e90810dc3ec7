mkdir -p gpurun_out
C5="--no-cpu-baseline --steps 5 --tris 10000000 --extent 0.02 --shadows 0 --bounces 4 --prelude-s 0 --no-extras"
timeout -k 10 200 python bench.py $C5 --host-build > gpurun_out/r03f_c5b_host.json 2>/dev/null || exit 1
timeout -k 10 200 python bench.py $C5 > gpurun_out/r03f_c5b_dev.json 2>/dev/null || exit 1
python - <<'PY'
import json
for f in ("r03f_c5b_host", "r03f_c5b_dev"):
    d = json.load(open("gpurun_out/%s.json" % f))
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), "build", round(d["build_s"], 2))
PY
