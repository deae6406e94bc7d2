export TMPDIR=/tmp
O=gpurun_out/r2t
mkdir -p $O
for ev in 128 160 192; do
AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --config A > $O/A_$ev.log 2>&1; echo "A $ev $(tail -1 $O/A_$ev.log | cut -c40-70)"
done
for se in 32 48 64 96; do
AZD_POOL_EVAL_WGS=160 AZD_POOL_SEARCH_WGS=$se timeout -k 10 200 python bench.py --no-cpu-baseline --config A > $O/As_$se.log 2>&1; echo "A 160 search $se $(tail -1 $O/As_$se.log | cut -c40-70)"
done
