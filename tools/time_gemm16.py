#!/usr/bin/env python3
"""The evaluator's bf16 forward GEMM (k_gemm16: LDS-DMA staged, v_mfma_f32_32x32x16_bf16) in isolation: checked against torch
(A . W^T in f32 from the same bf16 values; A = I with an asymmetric W as well) and timed.
usage: time_gemm16.py [M N K ...]   (triples; default: config E's layers and 8192 x 4096 x 4096)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from azdopt_amd import _lib  # noqa: E402

L = _lib.lib()
args = [int(x) for x in sys.argv[1:]]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(8192, 512, 3676), (8192, 512, 512), (8192, 2450, 512), (8192, 4096, 4096), (2048, 512, 3676)]
dev = torch.device("cuda")


def run(M, N, K, a, w, bias, out_bf16, act, reps):
    Kp = (K + 63) // 64 * 64
    a16 = torch.zeros(M, Kp, dtype=torch.bfloat16, device=dev)
    w16 = torch.zeros(N, Kp, dtype=torch.bfloat16, device=dev)
    a16[:, :K] = a
    w16[:, :K] = w
    y = torch.zeros(M, N, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=dev)
    ms = C.c_float()
    _lib.check(L.azd_debug_gemm_bf16(0, M, N, Kp, C.c_void_p(a16.data_ptr()), C.c_void_p(w16.data_ptr()), C.c_void_p(bias.data_ptr()),
                                     C.c_void_p(y.data_ptr()), N, int(out_bf16), act, reps, C.byref(ms)), "gemm")
    return y, ms.value, a16, w16


# layout check: A = I, asymmetric W
M = N = K = 256
w = (torch.arange(N, device=dev)[:, None] * 3 + torch.arange(K, device=dev)[None, :] * 7) % 61
y, _, _, _ = run(M, N, K, torch.eye(M, K, device=dev), w.float(), torch.zeros(N, device=dev), False, 0, 1)
assert torch.equal(y, w.float().t().contiguous()), "A = I check failed"
for (M, N, K) in shapes:
    g = torch.Generator(device=dev).manual_seed(M + N + K)
    a = (torch.rand(M, K, device=dev, generator=g) * 2 - 1)
    w = (torch.rand(N, K, device=dev, generator=g) * 2 - 1) / K ** 0.5
    if os.environ.get("AZD_GEMM_ZEROS"):  # zero-filled operands read higher (less switching power: the guide's methodology note); for comparison only
        a, w = a * 0, w * 0
    bias = torch.rand(N, device=dev, generator=g) - 0.5
    for out_bf16, act in ((False, 2), (True, 1)):
        y, ms, a16, w16 = run(M, N, K, a, w, bias, out_bf16, act, 20)
        ref = a16.float() @ w16.float().t() + bias
        ref = torch.sigmoid(ref) if act == 2 else torch.relu(ref)
        err = (y.float() - ref).abs().max().item()
        tol = 2e-2 if out_bf16 else 2e-5 * K ** 0.5
        assert err < tol, (M, N, K, out_bf16, err)
        print("gemm16 %5d x %5d x %5d  out %s: %.3f ms  %7.1f TFLOP/s   max |err| %.2e" % (M, N, K, "bf16" if out_bf16 else "f32 ", ms, 2.0 * M * N * K / ms / 1e9, err), flush=True)
