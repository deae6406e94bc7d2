mkdir -p gpurun_out/r2m
timeout -k 10 600 python -m pytest tests/test_gpu_ramsey.py -m gpu -x -q > gpurun_out/r2m/ramsey.log 2>&1; tail -15 gpurun_out/r2m/ramsey.log
