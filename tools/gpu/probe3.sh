mkdir -p gpurun_out/r2b
{
for B in 4096 8192 16384; do timeout -k 10 120 python tools/pool_probe_hash.py $B 400; AZD_STEP_FORM=async timeout -k 10 120 python tools/pool_probe_hash.py $B 400; done
AZD_POOL_SEARCH_WGS=128 timeout -k 10 120 python tools/pool_probe_hash.py 8192 400
AZD_POOL_SEARCH_WGS=192 timeout -k 10 120 python tools/pool_probe_hash.py 8192 400
for ev in 80 96; do AZD_POOL_EVAL_WGS=$ev timeout -k 10 120 python tools/pool_probe.py 4096 400; done
AZD_POOL_EVAL_WGS=80 timeout -k 10 120 python tools/pool_probe.py 8192 400
} > gpurun_out/r2b/probe3.log 2>&1
cat gpurun_out/r2b/probe3.log
