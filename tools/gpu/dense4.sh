mkdir -p gpurun_out/r2o
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 300 python tools/dense_cycle.py 8192 200 > gpurun_out/r2o/dense.txt 2>&1
grep -v amdgpu gpurun_out/r2o/dense.txt
