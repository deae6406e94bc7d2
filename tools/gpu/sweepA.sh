export TMPDIR=/tmp
O=gpurun_out/r2t
mkdir -p $O
for ev in 24 33 48 64 96 128; do
AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --config A > $O/A_$ev.log 2>&1; echo "A $ev $(tail -1 $O/A_$ev.log | cut -c40-70)"
done
for ev in 33 48 65 96; do
AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --agents 1024 > $O/k_$ev.log 2>&1; echo "1024 $ev $(tail -1 $O/k_$ev.log | cut -c40-70)"
done
for ev in 64 88 112; do
AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --agents 2048 > $O/k2_$ev.log 2>&1; echo "2048 $ev $(tail -1 $O/k2_$ev.log | cut -c40-70)"
done
