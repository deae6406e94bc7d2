mkdir -p gpurun_out/r2h
timeout -k 10 300 python -m pytest tests/test_gpu_pool.py -x -q > gpurun_out/r2h/pool.log 2>&1; echo "rc=$?" >> gpurun_out/r2h/pool.log; tail -4 gpurun_out/r2h/pool.log
for v in "" _one; do
  for ev in 64 88; do
  AZD_POOL_EVAL_WGS=$ev AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so timeout -k 10 200 python bench.py --config B --no-cpu-baseline > gpurun_out/r2h/t_B${v}_$ev.json 2>/dev/null
  AZD_POOL_EVAL_WGS=$ev AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so timeout -k 10 200 python bench.py --agents 8192 --no-cpu-baseline > gpurun_out/r2h/t_B8192${v}_$ev.json 2>/dev/null
  done
  for ev in 40 57; do
  AZD_POOL_EVAL_WGS=$ev AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so timeout -k 10 200 python bench.py --config C --no-cpu-baseline > gpurun_out/r2h/t_C${v}_$ev.json 2>/dev/null
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2h/t_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["value"]/1e6,2), "M/s", j["step_form"], j["pool_split"], round(j["ms_per_step"]*1e3,1),"us")
    except Exception as e: print(f,"ERR",e)
PY
