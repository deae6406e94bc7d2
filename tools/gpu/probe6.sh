mkdir -p gpurun_out/r2b
{
for v in "" _p0 _p1b0 _p1s0; do
  echo "== variant libazdopt_amd$v.so"
  AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so AZD_POOL_EVAL_WGS=80 timeout -k 10 120 python tools/pool_probe.py 8192 300
  AZD_LIB=$PWD/azdopt_amd/libazdopt_amd$v.so AZD_POOL_EVAL_WGS=40 timeout -k 10 120 python tools/pool_probe.py 8192 300 bf16
done
} > gpurun_out/r2b/probe6.log 2>&1
cat gpurun_out/r2b/probe6.log
