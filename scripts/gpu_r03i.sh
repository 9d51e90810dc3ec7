mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03i_tests.log 2>&1 || { tail -40 gpurun_out/r03i_tests.log; exit 1; }
tail -3 gpurun_out/r03i_tests.log
bash scripts/gpu_ab.sh "dev:" "host:--host-build"
