mkdir -p gpurun_out/r2g
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2g/all.log 2>&1; echo "rc=$?" >> gpurun_out/r2g/all.log; tail -6 gpurun_out/r2g/all.log
for i in 1 2; do timeout -k 10 200 python bench.py --config B --no-cpu-baseline > gpurun_out/r2g/B_$i.json 2>/dev/null; done
timeout -k 10 200 python bench.py --config B --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r2g/B20.json 2>/dev/null
timeout -k 10 200 python bench.py --config C --no-cpu-baseline > gpurun_out/r2g/C.json 2>/dev/null
timeout -k 10 200 python bench.py --config B --step async --no-cpu-baseline > gpurun_out/r2g/B_async.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r2g/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], round(j["value"]/1e6,2), "M/s", j["step_form"], j["pool_split"], round(j["ms_per_step"]*1e3,1),"us")
    except Exception as e: print(f,"ERR",e)
PY
