#!/usr/bin/env python3
"""Per-rank step time of the region-sharded path WITHOUT communication: rank `r` of a `world`-GPU weak-scaling run (global
graph = world x cfg-3) on this one GPU, halo rows filled with random data.  What the step costs next to the single-GPU
step is the part of the weak-scaling efficiency that is not RCCL: global region count in the compositions, halo rows in
the aggregation.   python tools/shard_step_bench.py [world=8] [rank=0]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R
from regtgcn_amd import _lib

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
nodes, edges, regions, F, T, O = 100_000, 1_000_000, 8, 32, 12, 1
dev = torch.device("cuda")
lib = R.load_library()
g = R.data.synthetic_regional_graph(nodes * world, edges * world, regions * world, seed=42)
bounds = np.asarray(g.region_bounds[::regions], dtype=np.int64)
owner = [r // regions for r in range(regions * world)]
sh = R.dist.build_shard(g.edge_index, g.region_index, g.region_attr, nodes * world, bounds, owner, rank, world, dev)
print(f"world {world} rank {rank}: {sh.topo.n_local} local rows + {sh.topo.halo_rows} halo rows, regions {sh.graph.region_lo}..{sh.graph.region_hi} of {sh.graph.num_regions}")
torch.manual_seed(42)
model = R.RegionalTemporalGCN(F, nodes, T, O, num_regions=regions * world).to(dev)
xp = torch.rand(sh.topo.x_rows, T, F, device=dev)
y = torch.rand(nodes, O, device=dev)
inv = 1.0 / float(nodes * world * O)

def step():
    pred, _ = model.forward_packed(xp, sh.graph)
    loss = ((pred - y) ** 2).sum() * inv
    loss.backward()
    return loss

loss = None
for _ in range(3):
    loss = step()
torch.cuda.synchronize()
K = 10
lib.regt_profile_enable(1)
t0 = time.perf_counter()
for _ in range(K):
    loss = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.regt_profile_enable(0)
buf = (ctypes.c_char * 16384)()
_lib.check(lib.regt_profile_collect(buf, 16384), "regt_profile_collect")
print(f"{1e3 * dt / K:.3f} ms/step")
for line in sorted(buf.value.decode().splitlines(), key=lambda l: -float(l.split()[2]))[:24]:
    name, cnt, ms = line.split()
    print(f"  {name:18s} {float(ms) / K:8.3f} ms/step")
