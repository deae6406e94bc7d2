export TMPDIR=/tmp
O=gpurun_out/r2w
mkdir -p $O
for v in "" _fw0 _fw1 _fw4; do
export AZD_LIB=azdopt_amd/libazdopt_amd$v.so
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/b.log 2>&1; echo "B$v $(tail -1 $O/b.log | cut -c40-70)"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --config C > $O/b.log 2>&1; echo "C$v $(tail -1 $O/b.log | cut -c40-70)"
timeout -k 10 300 python bench.py --no-cpu-baseline --agents 2048 > $O/b.log 2>&1; echo "2048$v $(tail -1 $O/b.log | cut -c40-70)"
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b.log 2>&1; echo "B20$v $(tail -1 $O/b.log | cut -c40-70)"
done
