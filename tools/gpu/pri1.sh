mkdir -p gpurun_out/r2f
timeout -k 10 300 python -m pytest tests/test_gpu_pool.py -x -q > gpurun_out/r2f/pool_rp.log 2>&1; echo "rc=$?" >> gpurun_out/r2f/pool_rp.log; tail -4 gpurun_out/r2f/pool_rp.log
for v in "" _norp _rp4; do
  AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so timeout -k 10 200 python bench.py --config B --no-cpu-baseline > gpurun_out/r2f/rp_B$v.json 2>/dev/null
  AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so timeout -k 10 200 python bench.py --config B --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r2f/rp_B20$v.json 2>/dev/null
  AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so timeout -k 10 200 python bench.py --agents 8192 --no-cpu-baseline > gpurun_out/r2f/rp_B8192$v.json 2>/dev/null
  AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so timeout -k 10 200 python bench.py --config C --no-cpu-baseline > gpurun_out/r2f/rp_C$v.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2f/rp_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["value"]/1e6,2), "M/s", j["step_form"], j["pool_split"], round(j["ms_per_step"]*1e3,1),"us")
    except Exception as e: print(f,"ERR",e)
PY
