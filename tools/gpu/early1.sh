export TMPDIR=/tmp
O=gpurun_out/r2j
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pool.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for cfg in B C D; do timeout -k 10 200 python bench.py --no-cpu-baseline --config $cfg > $O/b_$cfg.log 2>&1; tail -1 $O/b_$cfg.log | cut -c1-120; done
timeout -k 10 200 python bench.py --no-cpu-baseline --agents 8192 > $O/b_8192.log 2>&1; tail -1 $O/b_8192.log | cut -c1-120
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b_20.log 2>&1; tail -1 $O/b_20.log | cut -c1-120
