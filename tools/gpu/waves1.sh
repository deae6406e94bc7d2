mkdir -p gpurun_out/r2j
{
for v in prof w8 w4; do
echo "== $v"
AZD_LIB=azdopt_amd/libazdopt_amd_$v.so timeout -k 10 200 python tools/pool_cycle.py 2048 400
done
} > gpurun_out/r2j/waves.txt 2>&1
grep -v amdgpu gpurun_out/r2j/waves.txt
