cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/t4.log 2>&1; tail -5 gpurun_out/t4.log
for r in 2048 4096; do
python bench.py --res $r --steps 2 --no-cpu-baseline > gpurun_out/e2_$r.json 2>gpurun_out/e2.err
python3 -c "
import json
d=json.load(open('gpurun_out/e2_$r.json')); print('res=$r', d['value'], d['ms_per_step'], d['kernels_ms'], d['rays_rank0']['tri_tests'], d['rays_rank0']['node_visits'])"
done
tail -2 gpurun_out/e2.err
