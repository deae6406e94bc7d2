mkdir -p gpurun_out/r2c
for v in "" _fw1 _fw3; do for ev in 72 88; do
  AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --config B --step pool --no-cpu-baseline > gpurun_out/r2c/fw_B${v}_$ev.json 2>/dev/null
done; done
for v in "" _fw1 _fw3; do
  AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so AZD_POOL_EVAL_WGS=72 timeout -k 10 200 python bench.py --agents 8192 --step pool --no-cpu-baseline > gpurun_out/r2c/fw_B8192${v}_72.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2c/fw_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["value"]/1e6,2), "M/s", j["step_form"], j["pool_split"], round(j["ms_per_step"]*1e3,1),"us")
    except Exception as e: print(f,"ERR",e)
PY
