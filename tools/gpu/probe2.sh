mkdir -p gpurun_out/r2b
timeout -k 10 300 python -m pytest tests/test_gpu_pool.py -x -q -k "hash_stream or equals_async_and_barrier or call_by_call or capacity" > gpurun_out/r2b/pool3.log 2>&1; echo "rc=$?" >> gpurun_out/r2b/pool3.log; tail -5 gpurun_out/r2b/pool3.log
{
for ev in 16 32 48 69; do AZD_POOL_EVAL_WGS=$ev timeout -k 10 120 python tools/pool_probe.py 4096 200; done
AZD_POOL_EVAL_WGS=48 timeout -k 10 120 python tools/pool_probe.py 8192 200
AZD_POOL_EVAL_WGS=16 timeout -k 10 120 python tools/pool_probe.py 8192 200 bf16
} > gpurun_out/r2b/probe2.log 2>&1
cat gpurun_out/r2b/probe2.log
