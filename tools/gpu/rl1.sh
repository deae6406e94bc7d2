mkdir -p gpurun_out/r2f
timeout -k 10 300 python -m pytest tests/test_gpu_pool.py -x -q > gpurun_out/r2f/pool_rl.log 2>&1; echo "rc=$?" >> gpurun_out/r2f/pool_rl.log; tail -4 gpurun_out/r2f/pool_rl.log
AZD_POOL_READY_LANES=1 timeout -k 10 300 python -m pytest tests/test_gpu_pool.py -x -q > gpurun_out/r2f/pool_rl1.log 2>&1; echo "rc=$?" >> gpurun_out/r2f/pool_rl1.log; tail -4 gpurun_out/r2f/pool_rl1.log
for v in 0 1; do
  AZD_POOL_READY_LANES=$v timeout -k 10 200 python bench.py --config B --no-cpu-baseline > gpurun_out/r2f/rl_B_$v.json 2>/dev/null
  AZD_POOL_READY_LANES=$v timeout -k 10 200 python bench.py --agents 8192 --no-cpu-baseline > gpurun_out/r2f/rl_B8192_$v.json 2>/dev/null
  AZD_POOL_READY_LANES=$v timeout -k 10 200 python bench.py --config C --no-cpu-baseline > gpurun_out/r2f/rl_C_$v.json 2>/dev/null
  AZD_POOL_READY_LANES=$v timeout -k 10 200 python bench.py --config D --no-cpu-baseline > gpurun_out/r2f/rl_D_$v.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2f/rl_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["value"]/1e6,2), "M/s", j["step_form"], j["pool_split"], round(j["ms_per_step"]*1e3,1),"us")
    except Exception as e: print(f,"ERR",e)
PY
