cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do
  SR_ROUND_CAP0=$m python bench.py --no-cpu-baseline > gpurun_out/bench_cap_$m.json 2> gpurun_out/bench_cap_$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_cap_$m.json"))
print("cap0 $m", round(d["ms_per_step"],2), {k: round(v,2) for k,v in d["kernels_ms"].items()}, d["pipeline_counters_last_band"])
PY
done
