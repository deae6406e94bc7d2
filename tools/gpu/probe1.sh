mkdir -p gpurun_out/r2b
{
for ev in 16 40 69 128; do AZD_POOL_EVAL_WGS=$ev timeout -k 10 120 python tools/pool_probe.py 4096 200; done
AZD_STEP_FORM=async timeout -k 10 120 python tools/pool_probe.py 4096 200
} > gpurun_out/r2b/probe1.log 2>&1
cat gpurun_out/r2b/probe1.log
