mkdir -p gpurun_out/r2e
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -x -q > gpurun_out/r2e/dense.log 2>&1; echo "rc=$?" >> gpurun_out/r2e/dense.log; tail -40 gpurun_out/r2e/dense.log
