mkdir -p gpurun_out/r2o
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py -m gpu -x -q > gpurun_out/r2o/pytest.log 2>&1; tail -3 gpurun_out/r2o/pytest.log
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 300 python tools/dense_cycle.py 8192 200 > gpurun_out/r2o/dense.txt 2>&1
grep -v amdgpu gpurun_out/r2o/dense.txt
timeout -k 10 400 python bench.py --config E --steps 400 --warmup 50 --no-cpu-baseline > gpurun_out/r2o/bench_E.log 2>&1; tail -1 gpurun_out/r2o/bench_E.log | cut -c1-130
