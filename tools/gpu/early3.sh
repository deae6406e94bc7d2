export TMPDIR=/tmp
O=gpurun_out/r2l
mkdir -p $O
for m in 0 1 2 0 1 2; do
AZD_POOL_EARLY_POST=$m timeout -k 10 200 python bench.py --no-cpu-baseline > $O/e3_B.log 2>&1; echo "B mode $m $(tail -1 $O/e3_B.log | cut -c40-70)"
done
for m in 0 1 2; do
AZD_POOL_EARLY_POST=$m timeout -k 10 200 python bench.py --no-cpu-baseline --agents 8192 > $O/e3_8.log 2>&1; echo "8192 mode $m $(tail -1 $O/e3_8.log | cut -c40-70)"
AZD_POOL_EARLY_POST=$m timeout -k 10 200 python bench.py --no-cpu-baseline --config C > $O/e3_C.log 2>&1; echo "C mode $m $(tail -1 $O/e3_C.log | cut -c40-70)"
AZD_POOL_EARLY_POST=$m timeout -k 10 200 python bench.py --no-cpu-baseline --config D > $O/e3_D.log 2>&1; echo "D mode $m $(tail -1 $O/e3_D.log | cut -c40-70)"
AZD_POOL_EARLY_POST=$m timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/e3_20.log 2>&1; echo "B20 mode $m $(tail -1 $O/e3_20.log | cut -c40-70)"
done
