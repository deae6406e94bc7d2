export TMPDIR=/tmp
O=gpurun_out/quick
mkdir -p $O
for cfg in A B C D; do timeout -k 10 200 python bench.py --no-cpu-baseline --config $cfg > $O/b_$cfg.log 2>&1; echo "$cfg $(tail -1 $O/b_$cfg.log | cut -c40-70)"; done
for ag in 1024 2048 8192; do timeout -k 10 200 python bench.py --no-cpu-baseline --agents $ag > $O/b_$ag.log 2>&1; echo "$ag $(tail -1 $O/b_$ag.log | cut -c40-70)"; done
timeout -k 10 200 python bench.py --no-cpu-baseline --step async > $O/b_async.log 2>&1; echo "B async $(tail -1 $O/b_async.log | cut -c40-70)"
timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/b_20.log 2>&1; echo "B 20 $(tail -1 $O/b_20.log | cut -c40-70)"
