mkdir -p gpurun_out/r2c
timeout -k 10 500 python -m pytest tests/test_gpu_pool.py -x -q > gpurun_out/r2c/pool.log 2>&1; echo "rc=$?" >> gpurun_out/r2c/pool.log; tail -6 gpurun_out/r2c/pool.log
for cfg in A B C D; do for step in pool async; do timeout -k 10 300 python bench.py --config $cfg --step $step --no-cpu-baseline > gpurun_out/r2c/bench_${step}_$cfg.json 2> gpurun_out/r2c/bench_${step}_$cfg.err; done; done
for ag in 64 512; do for step in pool async; do timeout -k 10 200 python bench.py --agents $ag --step $step --no-cpu-baseline > gpurun_out/r2c/bench_${step}_c21_$ag.json 2>/dev/null; done; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2c/bench_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["value"]/1e6,2), "M/s", j["step_form"], j["pool_split"], round(j["ms_per_step"]*1e3,1),"us", "best", round(j["best_cost_found"],4))
    except Exception as e: print(f,"ERR",e)
PY
