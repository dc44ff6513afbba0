#!/usr/bin/env python3
"""Accuracy of the bf16x3 split GEMM vs the fp32-MFMA GEMM vs float64, and their rates (regt_linear)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R
lib = R.load_library()

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

torch.manual_seed(0)
for m, k, n in [(4096, 288, 512), (4096, 2048, 256), (1000, 100, 36)]:
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") / 16; b = torch.randn(n, device="cuda")
    ref = (a.double() @ w.double().T + b.double())
    for mode in (0, 1):
        lib.regt_set_gemm_mode(mode)
        y = R.ops.linear(a, w, b, 0)
        err = (y.double() - ref).abs().max().item()
        print(f"M={m} K={k} N={n} mode={mode}: max|err| vs f64 = {err:.3e}  (ref max {ref.abs().max().item():.2f})")
for m, k, n in [(1_200_000, 256, 512), (1_200_000, 256, 256), (150_000, 2048, 512)]:
    a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") / 16; b = torch.randn(n, device="cuda")
    for mode in (0, 1):
        lib.regt_set_gemm_mode(mode)
        ms = timeit(lambda: R.ops.linear(a, w, b, 1))
        print(f"linear M={m} K={k} N={n} mode={mode}: {ms:7.3f} ms  {2.0*m*k*n/ms/1e9:7.1f} TFLOP/s (fp32-equivalent)")
    del a, w, b
lib.regt_set_gemm_mode(0)
