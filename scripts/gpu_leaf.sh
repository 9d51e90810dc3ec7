cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do
  SR_BVH_LEAF=$m python bench.py --no-cpu-baseline > gpurun_out/bench_leaf_$m.json 2> gpurun_out/bench_leaf_$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_leaf_$m.json"))
print("leaf $m", round(d["ms_per_step"],2), {k: round(v,2) for k,v in d["kernels_ms"].items()}, "primary_only", round(d["primary_only"]["value"]), "build_s", round(d["build_s"],2), d["pipeline_counters_last_band"])
PY
done
