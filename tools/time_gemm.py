#!/usr/bin/env python3
"""Time the evaluator's batched forward (write_predictions_dev) per dtype and print achieved TFLOP/s.
usage: time_gemm.py [f32|bf16] [batch] [dims, comma separated]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import azdopt_amd as az  # noqa: E402

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
dims = tuple(int(x) for x in sys.argv[3].split(",")) if len(sys.argv) > 3 else (3676, 512, 512, 512, 2450)
m = az.ActionModel(B, dims[0], dims[-1], hidden=dims[1:-1], seed=0, dtype=dtype)
x = (torch.rand(B, dims[0], device="cuda") < 0.3).float()
y = torch.zeros(B, dims[-1], device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    m.write_predictions_dev(B, x.data_ptr(), y.data_ptr(), stream=st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    m.write_predictions_dev(B, x.data_ptr(), y.data_ptr(), stream=st)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flop = 2.0 * B * sum(a * b for a, b in zip(dims[:-1], dims[1:]))
print("%s forward B=%d dims=%s: %.3f ms  %.1f TFLOP/s" % (dtype, B, "-".join(map(str, dims)), ms, flop / ms / 1e9))
