mkdir -p gpurun_out/r2e
timeout -k 10 600 python -m pytest tests/test_gpu_dense.py tests/test_gpu_parity.py -x -q > gpurun_out/r2e/dense2.log 2>&1; echo "rc=$?" >> gpurun_out/r2e/dense2.log; tail -15 gpurun_out/r2e/dense2.log
timeout -k 10 500 python bench.py --config E --steps 200 --warmup 20 > gpurun_out/r2e/bench_E_200.json 2> gpurun_out/r2e/bench_E_200.err; echo "E rc=$?"; tail -c 2500 gpurun_out/r2e/bench_E_200.json; tail -5 gpurun_out/r2e/bench_E_200.err
AZD_NO_CALL_GRAPH=1 timeout -k 10 500 python bench.py --config E --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r2e/bench_E_200_nograph.json 2>/dev/null; tail -c 600 gpurun_out/r2e/bench_E_200_nograph.json
