#!/usr/bin/env python3
"""One GEMM shape in one arithmetic mode, a few launches (target of rocprofv3 --pmc): split_pmc.py MODE M K N"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R
lib = R.load_library()
mode, m, k, n = (int(v) for v in sys.argv[1:5])
lib.regt_set_gemm_mode(mode)
a = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") / 16; b = torch.randn(n, device="cuda")
for _ in range(6):
    R.ops.linear(a, w, b, 1)
torch.cuda.synchronize()
