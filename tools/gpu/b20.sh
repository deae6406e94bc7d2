export TMPDIR=/tmp
O=gpurun_out/r2q
mkdir -p $O
for rep in 1 2; do
for m in 2 1; do
AZD_POOL_EARLY_POST=$m timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b20.log 2>&1; echo "B20 mode $m $(tail -1 $O/b20.log | cut -c40-70)"
done
done
for ev in 64 76 100 112; do
AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b20.log 2>&1; echo "B20 ev $ev $(tail -1 $O/b20.log | cut -c40-70)"
done
AZD_STEP_FORM=async timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b20.log 2>&1; echo "B20 async $(tail -1 $O/b20.log | cut -c40-70)"
