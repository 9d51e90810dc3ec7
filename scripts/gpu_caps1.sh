# sweep the second-round list length (SR_ROUND_CAP1) on the default and the 10 M-triangle scene
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in "$@"; do
  SR_ROUND_CAP1=$m python bench.py --tris 10000000 --extent 0.02 --no-cpu-baseline --steps 2 > gpurun_out/bench_10m_cap1_$m.json 2> gpurun_out/bench_10m_cap1_$m.err
  SR_ROUND_CAP1=$m python bench.py --no-cpu-baseline --steps 2 > gpurun_out/bench_1m_cap1_$m.json 2> gpurun_out/bench_1m_cap1_$m.err
  python - <<PY
import json
for n in ("10m", "1m"):
    d=json.load(open("gpurun_out/bench_%s_cap1_$m.json" % n))
    print(n, "cap1 $m", round(d["ms_per_step"],2), {k: round(v,2) for k,v in d["kernels_ms"].items()}, d["pipeline_counters_last_band"])
PY
done
