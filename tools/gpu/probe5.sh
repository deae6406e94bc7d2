mkdir -p gpurun_out/r2b
timeout -k 10 300 python -m pytest tests/test_gpu_pool.py -x -q -k "equals_async_and_barrier or call_by_call" > gpurun_out/r2b/pool4.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pool4.log; tail -5 gpurun_out/r2b/pool4.log
{
for ev in 64 80 96; do AZD_POOL_EVAL_WGS=$ev timeout -k 10 120 python tools/pool_probe.py 4096 400; done
AZD_POOL_EVAL_WGS=64 timeout -k 10 120 python tools/pool_probe.py 8192 400
AZD_POOL_EVAL_WGS=80 timeout -k 10 120 python tools/pool_probe.py 8192 400
AZD_POOL_EVAL_WGS=24 timeout -k 10 120 python tools/pool_probe.py 8192 400 bf16
AZD_POOL_EVAL_WGS=32 timeout -k 10 120 python tools/pool_probe.py 4096 400 bf16
} > gpurun_out/r2b/probe5.log 2>&1
cat gpurun_out/r2b/probe5.log
