#!/usr/bin/env python3
"""Wall-clock split of one training step (forward / loss / backward) with device synchronisation."""
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
(x, y), = R.data.synthetic_snapshots(nodes, F, T, O, 1)
x, y = x.to(dev), y.to(dev)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(6):
    t0 = sync()
    pred, _ = model.forward_prepared(x, graph)
    t1h = time.perf_counter(); t1 = sync()
    loss = ((pred - y) ** 2).mean()
    t2 = sync()
    loss.backward()
    t3h = time.perf_counter(); t3 = sync()
    print(f"iter {it}: fwd {1e3*(t1-t0):7.2f} ms (host enqueue {1e3*(t1h-t0):6.2f})  loss {1e3*(t2-t1):5.2f}  bwd {1e3*(t3-t2):7.2f} ms (host enqueue {1e3*(t3h-t2):6.2f})")
