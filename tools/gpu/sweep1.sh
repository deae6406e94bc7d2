mkdir -p gpurun_out/r2c
for ev in 64 88 100 112 128; do AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --config B --step pool --no-cpu-baseline > gpurun_out/r2c/sw_B_$ev.json 2>/dev/null; done
for ev in 32 56 64 80; do AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --config C --step pool --no-cpu-baseline > gpurun_out/r2c/sw_C_$ev.json 2>/dev/null; done
for ev in 64 80 96; do AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --agents 8192 --step pool --no-cpu-baseline > gpurun_out/r2c/sw_B8192_$ev.json 2>/dev/null; done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2c/sw_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["value"]/1e6,2), "M/s", j["step_form"], j["pool_split"], round(j["ms_per_step"]*1e3,1),"us")
    except Exception as e: print(f,"ERR",e)
PY
