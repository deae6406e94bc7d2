mkdir -p gpurun_out/r2h
rm -f gpurun_out/r2h/t_*.json
for ev in 72 88; do
  AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --config B --no-cpu-baseline > gpurun_out/r2h/t_B_$ev.json 2>/dev/null
done
for ev in 56 72 88; do
  AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --agents 8192 --no-cpu-baseline > gpurun_out/r2h/t_B8192_$ev.json 2>/dev/null
done
for ev in 32 40 48; do
  AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --config C --no-cpu-baseline > gpurun_out/r2h/t_C_$ev.json 2>/dev/null
done
for ev in 72 96 113; do
  AZD_POOL_EVAL_WGS=$ev timeout -k 10 200 python bench.py --config D --no-cpu-baseline > gpurun_out/r2h/t_D_$ev.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2h/t_*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["value"]/1e6,2), "M/s", j["step_form"], j["pool_split"], round(j["ms_per_step"]*1e3,1),"us")
    except Exception as e: print(f,"ERR",e)
PY
