mkdir -p gpurun_out/r2p
{
for dt in bf16; do
timeout -k 10 100 python tools/time_gemm.py $dt 8192
timeout -k 10 100 python tools/time_gemm.py $dt 65536 304,256,256,256,152
timeout -k 10 100 python tools/time_gemm.py $dt 8192 4096,4096,4096
done
timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -m gpu -x -q 2>&1 | tail -2
} > gpurun_out/r2p/gemm.txt 2>&1
grep -v amdgpu gpurun_out/r2p/gemm.txt
