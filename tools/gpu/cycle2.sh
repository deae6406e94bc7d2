mkdir -p gpurun_out/r2l
{
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_cycle.py 4096 800
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_tail.py 4096 800
} > gpurun_out/r2l/cycle.txt 2>&1
grep -v amdgpu gpurun_out/r2l/cycle.txt
