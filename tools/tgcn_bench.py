#!/usr/bin/env python3
"""Step time of the TemporalGCN (A3T-GCN) baseline at the cfg-3 shape: the same kernels as RegT-GCN with one Chebyshev
operator on the full weighted graph instead of the regional ones (models/TemporalGCN.py:82-91)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R

nodes, edges, F, T = 100_000, 1_000_000, 32, 12
dev = torch.device("cuda")
lib = R.load_library()
if os.environ.get("REGT_TGCN_COLLAPSE") == "0":
    print("uncollapsed gates (REGT_TGCN_COLLAPSE=0)")
g = R.data.synthetic_regional_graph(nodes, edges, 8, seed=42)
torch.manual_seed(42)
model = R.TemporalGCN(F, T, 1).to(dev)
graph = model.prepare_graph(g.edge_index.to(dev), g.edge_attr.to(dev), nodes)
snaps = [(x.to(dev), y.to(dev)) for x, y in R.data.synthetic_snapshots(nodes, F, T, 1, 2, seed=42)]


def step(i):
    x, y = snaps[i % 2]
    pred, _ = model.forward_prepared(x, graph)
    loss = torch.mean((pred - y) ** 2)
    loss.backward()
    return loss


for i in range(3):
    loss = step(i)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for i in range(K):
    loss = step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"TemporalGCN N={nodes} E={edges} F={F} T={T}: {1e3 * dt / K:.2f} ms/step ({K / dt:.1f} snapshots/s), loss {float(loss.detach()):.4f}")
