mkdir -p gpurun_out/r2b
{
for ev in 96 112 128; do AZD_POOL_EVAL_WGS=$ev timeout -k 10 120 python tools/pool_probe.py 4096 400; done
AZD_POOL_EVAL_WGS=64 timeout -k 10 120 python tools/pool_probe.py 8192 400
AZD_POOL_EVAL_WGS=96 timeout -k 10 120 python tools/pool_probe.py 8192 400
AZD_POOL_EVAL_WGS=24 timeout -k 10 120 python tools/pool_probe.py 8192 400 bf16
AZD_POOL_EVAL_WGS=40 timeout -k 10 120 python tools/pool_probe.py 8192 400 bf16
AZD_POOL_EVAL_WGS=40 timeout -k 10 120 python tools/pool_probe.py 4096 400 bf16
} > gpurun_out/r2b/probe4.log 2>&1
cat gpurun_out/r2b/probe4.log
