cd /root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_dense.py -x -q -m gpu 2>&1 | tail -5 || exit 1
for cfg in "--config E --steps 200 --warmup 50 --agents 1024 --max-slots 612 --prediction-capacity 524288" "--config E --steps 100 --warmup 20 --agents 8192 --max-slots 612 --prediction-capacity 524288" "--config E --steps 100 --warmup 20 --agents 2048 --max-slots 1024 --prediction-capacity 786432"; do
timeout -k 10 500 python bench.py $cfg --no-cpu-baseline > gpurun_out/e612.json 2> gpurun_out/e612.err || { tail -3 gpurun_out/e612.err; exit 1; }
python - <<'PY'
import json
j=json.loads(open('/root/repo/gpurun_out/e612.json').read().strip().splitlines()[-1])
print(round(j["value"]/1e6,3), j["step_form"], j["step_form_reason"], j["pool_split"], j["roofline"]["kernel"], j["config"]["agents_total"])
PY
done
