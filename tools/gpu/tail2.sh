mkdir -p gpurun_out/r2j
make -s -C azdopt_amd/csrc PROFILE=1 -j8 2>&1 | grep -v warning | head -5
{
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_tail.py 4096 800
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_cycle.py 4096 800
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_cycle.py 8192 800
} > gpurun_out/r2j/tail.txt 2>&1
grep -v amdgpu gpurun_out/r2j/tail.txt
