export TMPDIR=/tmp
O=gpurun_out/r2m
mkdir -p $O
for ev in 80 88 96 104 112; do
AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --config B > $O/B_$ev.log 2>&1; echo "B $ev $(tail -1 $O/B_$ev.log | cut -c40-70)"
done
for ev in 64 72 80 88; do
AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --agents 8192 > $O/B8_$ev.log 2>&1; echo "B8192 $ev $(tail -1 $O/B8_$ev.log | cut -c40-70)"
done
for ev in 40 47 54 61; do
AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --config C > $O/C_$ev.log 2>&1; echo "C $ev $(tail -1 $O/C_$ev.log | cut -c40-70)"
done
for ev in 100 113 126; do
AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --config D > $O/D_$ev.log 2>&1; echo "D $ev $(tail -1 $O/D_$ev.log | cut -c40-70)"
done
