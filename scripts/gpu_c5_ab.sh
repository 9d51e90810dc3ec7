# A/B of sr_debug_set hooks on the C5 four-bounce workload: bash scripts/gpu_c5_ab.sh "<name>:<bench args>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 4 --warmup 1 --prelude-s 0 --tris 10000000 --extent 0.02 --shadows 0 --bounces 4 --no-extras $args > gpurun_out/c5ab_$name.json 2> gpurun_out/c5ab_$name.err || { tail -20 gpurun_out/c5ab_$name.err; exit 1; }
  python - $name <<'PY'
import json, sys
d = json.load(open("gpurun_out/c5ab_%s.json" % sys.argv[1]))
print(sys.argv[1], round(d["value"], 1), "ms", round(d["ms_per_step"], 2))
PY
done
