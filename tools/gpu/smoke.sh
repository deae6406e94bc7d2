mkdir -p gpurun_out/r2g
timeout -k 10 300 python -m pytest tests/test_gpu_pool.py -x -q > gpurun_out/r2g/pool5.log 2>&1; echo "rc=$?" >> gpurun_out/r2g/pool5.log; tail -6 gpurun_out/r2g/pool5.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
