"""Time NablaModel::update_model on the device for the batch sizes an N-GPU run hands to every rank
(the training triple is all-gathered: B_total = N x 4096 rows)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import azdopt_amd as az

S, A = 304, 152
for B in (4096, 8192, 16384, 32768, 65536):
    model = az.ActionModel(B, S, A, hidden=(256, 256, 256), seed=1)
    g = torch.Generator(device="cuda").manual_seed(0)
    s = (torch.rand(B, S, device="cuda", generator=g) < 0.1).float()
    o = torch.rand(B, A, device="cuda", generator=g)
    w = (torch.rand(B, A, device="cuda", generator=g) < 0.2).float()
    torch.cuda.synchronize()
    losses = []
    for it in range(3):
        losses.append(model.update_model_dev(B, s.data_ptr(), o.data_ptr(), w.data_ptr()))
    torch.cuda.synchronize()
    t = time.perf_counter()
    n = 10
    for it in range(n):
        model.update_model_dev(B, s.data_ptr(), o.data_ptr(), w.data_ptr())
    torch.cuda.synchronize()
    print("B=%6d  update_model %.3f ms  (first losses %s)" % (B, (time.perf_counter() - t) / n * 1e3, " ".join("%.6f" % x for x in losses)), flush=True)
