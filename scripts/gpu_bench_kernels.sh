# default bench with the per-kernel table printed; arguments are SR_DEBUG values (0 = production; 7 counts umbra hits)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do
  SR_DEBUG=$m python bench.py --no-cpu-baseline > gpurun_out/bench_dbg_$m.json 2> gpurun_out/bench_dbg_$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_dbg_$m.json"))
print("debug $m", round(d["ms_per_step"],2), {k: round(v,2) for k,v in d["kernels_ms"].items()})
PY
done
