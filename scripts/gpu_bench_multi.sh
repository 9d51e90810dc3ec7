# rehearsals of the two multi-GPU paths on a one-GPU box: (a) one process, sr_create_multi with device 0 listed N times;
# (b) N ranks sharing the GPU, gloo instead of RCCL (RCCL refuses two ranks on one device)
set -x
N=${1:-2}
python bench.py --gpus $N --in-library --same-device --steps 5 --warmup 2 --prelude-s 0 --verify > gpurun_out/bench_inlib_$N.json 2> gpurun_out/bench_inlib_$N.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N --steps 5 --warmup 2 --backend gloo --verify > gpurun_out/bench_ranks_$N.json 2> gpurun_out/bench_ranks_$N.err
tail -n 2 gpurun_out/bench_inlib_$N.err gpurun_out/bench_ranks_$N.err
