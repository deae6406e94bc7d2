export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
for i in 1 2; do
timeout -k 10 200 python bench.py --no-cpu-baseline > $O/e4_B.log 2>&1; echo "B $(tail -1 $O/e4_B.log | cut -c40-70)"
AZD_POOL_EARLY_POST=1 timeout -k 10 200 python bench.py --no-cpu-baseline > $O/e4_B1.log 2>&1; echo "B mode1 $(tail -1 $O/e4_B1.log | cut -c40-70)"
done
timeout -k 10 200 python bench.py --no-cpu-baseline --agents 8192 > $O/e4_8.log 2>&1; echo "8192 $(tail -1 $O/e4_8.log | cut -c40-70)"
timeout -k 10 200 python bench.py --no-cpu-baseline --config C > $O/e4_C.log 2>&1; echo "C $(tail -1 $O/e4_C.log | cut -c40-70)"
timeout -k 10 200 python bench.py --no-cpu-baseline --config D > $O/e4_D.log 2>&1; echo "D $(tail -1 $O/e4_D.log | cut -c40-70)"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/e4_20.log 2>&1; echo "B20 $(tail -1 $O/e4_20.log | cut -c40-70)"
