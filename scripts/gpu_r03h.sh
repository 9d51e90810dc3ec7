mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "device_built" > gpurun_out/r03h_tests.log 2>&1 || { tail -40 gpurun_out/r03h_tests.log; exit 1; }
tail -3 gpurun_out/r03h_tests.log
timeout -k 10 300 python scripts/gpu_lbvh.py 1000000 0.05 > gpurun_out/r03h_lbvh_1m.json 2> gpurun_out/r03h_lbvh_1m.err || { tail gpurun_out/r03h_lbvh_1m.err; exit 1; }
cat gpurun_out/r03h_lbvh_1m.json
timeout -k 10 300 python scripts/gpu_lbvh.py 10000000 0.02 > gpurun_out/r03h_lbvh_10m.json 2> gpurun_out/r03h_lbvh_10m.err || { tail gpurun_out/r03h_lbvh_10m.err; exit 1; }
cat gpurun_out/r03h_lbvh_10m.json
