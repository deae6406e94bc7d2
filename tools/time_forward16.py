#!/usr/bin/env python3
"""The bf16 evaluator's whole forward (ActionModel.write_predictions_dev) timed with the GPU otherwise idle.
usage: time_forward16.py [rows ...]   (default 1792 3584 8192; config E's model 3676-512-512-512-2450;
AZD_MLP_FUSE_HIDDEN=0: one launch per layer)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import azdopt_amd as az  # noqa: E402

dims = (3676, 512, 512, 512, 2450)
for rows in [int(a) for a in sys.argv[1:]] or [1792, 3584, 8192]:
    m = az.ActionModel(rows, dims[0], dims[-1], hidden=dims[1:-1], seed=1, dtype="bf16")
    x = (torch.rand(rows, dims[0], device="cuda") < 0.3).float()
    y = torch.zeros(rows, dims[-1], device="cuda")
    for _ in range(3):
        m.write_predictions_dev(rows, x.data_ptr(), y.data_ptr())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        m.write_predictions_dev(rows, x.data_ptr(), y.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    print("forward %s at %5d rows: %.1f us (fuse_hidden=%s; includes the f32 -> bf16 conversion of the input rows)"
          % ("-".join(map(str, dims)), rows, e0.elapsed_time(e1) / reps * 1e3, os.environ.get("AZD_MLP_FUSE_HIDDEN", "1")), flush=True)
