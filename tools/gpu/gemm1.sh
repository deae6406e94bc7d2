mkdir -p gpurun_out/r2d
timeout -k 10 300 python -m pytest tests/test_gpu_bf16.py -x -q > gpurun_out/r2d/bf16.log 2>&1; echo "rc=$?" >> gpurun_out/r2d/bf16.log; tail -5 gpurun_out/r2d/bf16.log
{
for dt in bf16 f32; do
timeout -k 10 100 python tools/time_gemm.py $dt 8192
timeout -k 10 100 python tools/time_gemm.py $dt 8192 304,256,256,256,152
timeout -k 10 100 python tools/time_gemm.py $dt 65536 304,256,256,256,152
timeout -k 10 100 python tools/time_gemm.py $dt 8192 4096,4096,4096
done
} > gpurun_out/r2d/gemm.log 2>&1
cat gpurun_out/r2d/gemm.log
