cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do
  SR_BAND_SAMPLES=$m python bench.py --no-cpu-baseline > gpurun_out/bench_band_$m.json 2> gpurun_out/bench_band_$m.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_band_$m.json"))
print("band $m", round(d["ms_per_step"],2), {k: round(v,2) for k,v in d["kernels_ms"].items()}, d["kernel_launches"]["k_primary"])
PY
done
