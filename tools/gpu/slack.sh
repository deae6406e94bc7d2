export TMPDIR=/tmp
O=gpurun_out/r2r
mkdir -p $O
for v in "" _s1 _s0 _s4; do
export AZD_LIB=azdopt_amd/libazdopt_amd$v.so
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/b.log 2>&1; echo "B$v $(tail -1 $O/b.log | cut -c40-70)"
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b.log 2>&1; echo "B20$v $(tail -1 $O/b.log | cut -c40-70)"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --agents 8192 > $O/b.log 2>&1; echo "8192$v $(tail -1 $O/b.log | cut -c40-70)"
done
