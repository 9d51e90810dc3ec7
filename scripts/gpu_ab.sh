# A/B of sr_debug_set hooks on the default bench workload: bash scripts/gpu_ab.sh "<name>:<bench args>" ...   -> gpurun_out/ab_<name>.json
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 --prelude-s 0 $args > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err || { tail -20 gpurun_out/ab_$name.err; exit 1; }
done
python - "$@" <<'PY'
import json, sys
for spec in sys.argv[1:]:
    name = spec.split(":")[0]
    d = json.load(open("gpurun_out/ab_%s.json" % name))
    r = d.get("roofline") or {}
    print(name, round(d["value"], 1), "ms", round(d["ms_per_step"], 2), "one-pipe", round(d.get("ms_per_step_one_pipeline", 0), 2), d.get("frame_crc"),
          {k: round(v, 2) for k, v in (r.get("all_kernels_ms_per_launch") or {}).items()}, "primary_only", round((d.get("primary_only") or {}).get("ms_per_step", 0), 2))
PY
