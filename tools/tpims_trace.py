#!/usr/bin/env python3
"""TPIMS-scale training steps only (target of rocprofv3 --kernel-trace --stats): which kernels, how long, how many."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R
d = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "tpims_fixture.npz"))
fx = {k: torch.from_numpy(d[k]) for k in d.files if d[k].ndim > 0}
REG = ("IA", "KS", "KY", "OH", "WI")
T, O = 12, 1
dev = torch.device("cuda")
n = fx["node_data"].shape[0]
torch.manual_seed(0)
model = R.RegionalTemporalGCN(8, n, T, O).to(dev)
graph = model.prepare_graph(fx["edge_index"].to(dev), [fx[f"edge_{r}_index"].to(dev) for r in REG], [fx[f"edge_{r}_attr"].to(dev) for r in REG])
xs, ys = R.data.snapshot_windows(fx["node_data"], T, O)
xs, ys = [x.to(dev) for x in xs], [y.to(dev) for y in ys]
K = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for i in range(K):
    pred, _ = model.forward_prepared(xs[i % len(xs)], graph)
    torch.mean((pred - ys[i % len(xs)]) ** 2).backward()
torch.cuda.synchronize()
