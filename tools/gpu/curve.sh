mkdir -p gpurun_out/r2f
{
timeout -k 10 120 python tools/launch_curve.py 4096 20
timeout -k 10 120 python tools/launch_curve.py 4096 100
timeout -k 10 120 python tools/launch_curve.py 4096 800
AZD_STEP_FORM=async timeout -k 10 120 python tools/launch_curve.py 4096 20
AZD_STEP_FORM=async timeout -k 10 120 python tools/launch_curve.py 4096 800
} > gpurun_out/r2f/curve.txt 2>&1
grep -v amdgpu gpurun_out/r2f/curve.txt
