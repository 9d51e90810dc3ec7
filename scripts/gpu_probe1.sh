set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --res 1024 --shadows 0 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_1024_s0.json 2> gpurun_out/b_1024_s0.err
python bench.py --res 4096 --shadows 0 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_4096_s0.json 2> gpurun_out/b_4096_s0.err
timeout -k 10 300 python bench.py --res 1024 --shadows 100 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/b_1024_s100.json 2> gpurun_out/b_1024_s100.err
timeout -k 10 300 python bench.py --res 1024 --shadows 1 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/b_1024_s1.json 2> gpurun_out/b_1024_s1.err
timeout -k 10 200 python bench.py --res 1024 --shadows 0 --mode ref --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/b_1024_ref.json 2> gpurun_out/b_1024_ref.err
tail -c 600 gpurun_out/b_*.json
tail -n 3 gpurun_out/b_*.err
