export TMPDIR=/tmp
O=gpurun_out/r2q
mkdir -p $O
for ev in 88 100 112 124 136; do
AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --no-cpu-baseline --config D > $O/D_$ev.log 2>&1; echo "D $ev $(tail -1 $O/D_$ev.log | cut -c40-70)"
done
timeout -k 10 200 python bench.py --no-cpu-baseline --config D > $O/D.log 2>&1; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2q/D.log').read().strip().splitlines()[-1]); print('D default', d['value']/1e6, d['pool_split'])
PY
