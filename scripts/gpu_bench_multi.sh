# single-GPU rehearsal of the N>1 bench path: 2 ranks share device 0, gloo backend (RCCL refuses two ranks on one device)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --res 2048 --backend gloo --verify > gpurun_out/multi2.json 2> gpurun_out/multi2.err
tail -c 900 gpurun_out/multi2.json; tail -5 gpurun_out/multi2.err
python bench.py --verify > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 2500 gpurun_out/bench_default.json; tail -3 gpurun_out/bench_default.err
