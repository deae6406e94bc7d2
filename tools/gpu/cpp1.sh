mkdir -p gpurun_out/r2v
timeout -k 10 600 python -m pytest tests/test_gpu_examples.py -m gpu -x -q > gpurun_out/r2v/pytest.log 2>&1; tail -25 gpurun_out/r2v/pytest.log
