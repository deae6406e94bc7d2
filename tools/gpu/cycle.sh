mkdir -p gpurun_out/r2f
{
AZD_LIB=$PWD/azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_cycle.py 4096 800
AZD_LIB=$PWD/azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_cycle.py 8192 800
AZD_LIB=$PWD/azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_cycle.py 4096 20
} > gpurun_out/r2f/cycle.txt 2>&1
grep -v amdgpu gpurun_out/r2f/cycle.txt
