#!/usr/bin/env python3
"""Micro-benchmark of the fp32-MFMA GEMM entry points (regt_linear / regt_wgrad) at pipeline shapes."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


def main():
    R.load_library()
    shapes = [(1_200_000, 256, 512), (1_200_000, 256, 256), (150_000, 2048, 512), (75_000, 4096, 512), (1_200_000, 64, 256), (100_000, 256, 128)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split('x')) for a in sys.argv[1:]]
    for m, k, n in shapes:
        a = torch.randn(m, k, device="cuda")
        w = torch.randn(n, k, device="cuda") / 16
        b = torch.randn(n, device="cuda")
        ms = timeit(lambda: R.ops.linear(a, w, b, 1))
        print(f"linear  M={m:8d} K={k:4d} N={n:4d}: {ms:8.3f} ms  {2.0*m*k*n/ms/1e9:7.1f} TFLOP/s")
        if n <= 512 and k <= 256:
            d = torch.randn(m, n, device="cuda")
            ms = timeit(lambda: R.ops.wgrad(d, a))
            print(f"wgrad   M={m:8d} N={n:4d} K={k:4d}: {ms:8.3f} ms  {2.0*m*k*n/ms/1e9:7.1f} TFLOP/s")
        del a, w, b


if __name__ == "__main__":
    main()
