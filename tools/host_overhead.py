#!/usr/bin/env python3
"""Host-side enqueue time per training step (no device sync inside the loop) vs device time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R

nodes, edges, regions, F, T, O = 100_000, 1_000_000, 8, 32, 12, 1
dev = torch.device("cuda")
g = R.data.synthetic_regional_graph(nodes, edges, regions, seed=42)
torch.manual_seed(42)
model = R.RegionalTemporalGCN(F, nodes, T, O, num_regions=regions).to(dev)
graph = R.prepare_graph(g.edge_index.to(dev), None, [t.to(dev) for t in g.region_index], [t.to(dev) for t in g.region_attr], nodes)
snaps = R.data.synthetic_snapshots(nodes, F, T, O, 2)
xs = [x.to(dev) for x, _ in snaps]; ys = [y.to(dev) for _, y in snaps]
def step(i):
    pred, _ = model.forward_prepared(xs[i % 2], graph)
    loss = ((pred - ys[i % 2]) ** 2).sum() * 1e-5
    loss.backward()
    return loss
for i in range(3): loss = step(i)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for i in range(20):
    a = time.perf_counter(); loss = step(i); host.append(time.perf_counter() - a)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
host.sort()
print(f"wall/step {1e3*wall/20:.2f} ms; host enqueue per step: min {1e3*host[0]:.2f} median {1e3*host[10]:.2f} max {1e3*host[-1]:.2f} ms")
print("mem allocated GB", torch.cuda.memory_allocated()/2**30, "reserved GB", torch.cuda.memory_reserved()/2**30, "num_alloc_retries", torch.cuda.memory_stats().get("num_alloc_retries"), "segments", torch.cuda.memory_stats().get("segment.all.allocated"))
