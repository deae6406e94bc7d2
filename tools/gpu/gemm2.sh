mkdir -p gpurun_out/r2g
timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -x -q > gpurun_out/r2g/bf16.log 2>&1; echo "rc=$?" >> gpurun_out/r2g/bf16.log; tail -3 gpurun_out/r2g/bf16.log
{
for sb in 0 1024 100000; do
echo "== AZD_GEMM_SMALL_BELOW=$sb (0: always 128x128; 100000: always 64x128)"
AZD_GEMM_SMALL_BELOW=$sb timeout -k 10 100 python tools/time_gemm.py bf16 8192
AZD_GEMM_SMALL_BELOW=$sb timeout -k 10 100 python tools/time_gemm.py bf16 8192 304,256,256,256,152
AZD_GEMM_SMALL_BELOW=$sb timeout -k 10 100 python tools/time_gemm.py bf16 65536 304,256,256,256,152
AZD_GEMM_SMALL_BELOW=$sb timeout -k 10 100 python tools/time_gemm.py bf16 8192 4096,4096,4096
done
} > gpurun_out/r2g/gemm2.txt 2>&1
grep -v "amdgpu\|Exception\|Traceback\|File\|TypeError" gpurun_out/r2g/gemm2.txt
