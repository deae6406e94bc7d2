mkdir -p gpurun_out/final_r02
{
echo "# tools/pool_tail.py: which agents a launch waits for"
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_tail.py 4096 800
} > gpurun_out/final_r02/pool_tail.txt 2>&1
tail -5 gpurun_out/final_r02/pool_tail.txt
