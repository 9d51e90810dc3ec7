cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for d in 0 1 2; do
SR_DEBUG=$d python bench.py --res 2048 --steps 2 --no-cpu-baseline > gpurun_out/e1_$d.json 2>gpurun_out/e1.err
python3 -c "
import json
d=json.load(open('gpurun_out/e1_$d.json')); print('debug=$d', d['ms_per_step'], d['kernels_ms'], d['rays_rank0']['tri_tests'], d['rays_rank0']['node_visits'])"
done
