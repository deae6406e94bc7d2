mkdir -p gpurun_out/r2e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2e/all.log 2>&1; echo "rc=$?" >> gpurun_out/r2e/all.log; tail -8 gpurun_out/r2e/all.log
