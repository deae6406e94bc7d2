mkdir -p gpurun_out/r2b
timeout -k 10 400 python -m pytest tests/test_gpu_pool.py -x -q -s > gpurun_out/r2b/pool2.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pool2.log; tail -15 gpurun_out/r2b/pool2.log
for cfg in B C A; do timeout -k 10 200 python bench.py --config $cfg --step pool --no-cpu-baseline > gpurun_out/r2b/bench_pool_$cfg.json 2> gpurun_out/r2b/bench_pool_$cfg.err; echo "$cfg rc=$?"; done
timeout -k 10 200 python bench.py --agents 8192 --step pool --no-cpu-baseline > gpurun_out/r2b/bench_pool_8192f32.json 2>&1
timeout -k 10 200 python bench.py --agents 8192 --no-cpu-baseline > gpurun_out/r2b/bench_async_8192f32.json 2>&1
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2b/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(j["value"]/1e6,2), "M/s", j["step_form"], j["pool_split"], round(j["ms_per_step"]*1e3,1),"us")
    except Exception as e: print(f,"ERR",e)
PY
