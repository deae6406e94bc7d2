mkdir -p gpurun_out/final_r02
{
echo "# tools/pool_cycle.py (diagnostic build, make PROFILE=1): where an agent's cycle goes in the pool step"
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_cycle.py 4096 800
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_cycle.py 8192 800
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_cycle.py 2048 800
echo "# tools/pool_tail.py: which agents a launch waits for"
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 200 python tools/pool_tail.py 4096 800
echo "# tools/dense_cycle.py: config E"
AZD_LIB=azdopt_amd/libazdopt_amd_prof.so timeout -k 10 300 python tools/dense_cycle.py 8192 200
echo "# tools/probes/chase_latency: round trip of a 64-lane x 16-B gather by footprint and waves per CU"
timeout -k 10 200 tools/probes/chase_latency
echo "# tools/max_frontier.py"
timeout -k 10 300 python tools/max_frontier.py
} > gpurun_out/final_r02/pool_cycle.txt 2>&1
grep -v amdgpu gpurun_out/final_r02/pool_cycle.txt | head -80
