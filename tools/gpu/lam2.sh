export TMPDIR=/tmp
O=gpurun_out/r2s
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pool.py -m gpu -x -q > $O/pytest.log 2>&1; tail -2 $O/pytest.log
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/b.log 2>&1; echo "B $(tail -1 $O/b.log | cut -c40-70)"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --config C > $O/b.log 2>&1; echo "C $(tail -1 $O/b.log | cut -c40-70)"
timeout -k 10 300 python bench.py --no-cpu-baseline --agents 8192 > $O/b.log 2>&1; echo "8192 $(tail -1 $O/b.log | cut -c40-70)"
timeout -k 10 300 python bench.py --no-cpu-baseline --agents 2048 > $O/b.log 2>&1; echo "2048 $(tail -1 $O/b.log | cut -c40-70)"
