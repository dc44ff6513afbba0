#!/usr/bin/env python3
"""Per-rank step time of the region-sharded path WITHOUT communication: rank `r` of a `world`-GPU run executed alone on this
one GPU, halo rows filled with random data.

    python tools/shard_step_bench.py [world=8] [rank=0] [weak|strong] [gemm mode]

weak   (BASELINE configs[4] shape of growth): global graph = world x cfg-3, the rank owns 100k nodes / 8 of 8*world regions;
strong (BASELINE configs[3]):                 the ONE cfg-3 graph split by regions, the rank owns 100k / world nodes.
What the step costs next to (weak) the single-GPU step or (strong) 1/world of it is the part of the scaling efficiency that
is not RCCL: global region count in the compositions, halo rows in the aggregation, launch latency of short kernels."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R
from regtgcn_amd import _lib

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rank = int(sys.argv[2]) if len(sys.argv) > 2 else 0
scaling = sys.argv[3] if len(sys.argv) > 3 else "weak"
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
nodes, edges, regions, F, T, O = 100_000, 1_000_000, 8, 32, 12, 1
dev = torch.device("cuda")
lib = R.load_library()
lib.regt_set_gemm_mode(mode)
if scaling == "weak":
    gn, ge, gr = nodes * world, edges * world, regions * world
else:
    gn, ge, gr = nodes, edges, regions
g = R.data.synthetic_regional_graph(gn, ge, gr, seed=42)
rpg = gr // world
bounds = np.asarray(g.region_bounds[::rpg], dtype=np.int64)
owner = [r // rpg for r in range(gr)]
sh = R.dist.build_shard(g.edge_index, g.region_index, g.region_attr, gn, bounds, owner, rank, world, dev)
n_local = sh.topo.n_local
print(f"{scaling} world {world} rank {rank}: {n_local} local rows + {sh.topo.halo_rows} halo rows, regions {sh.graph.region_lo}..{sh.graph.region_hi} of {sh.graph.num_regions}")
torch.manual_seed(42)
model = R.RegionalTemporalGCN(F, n_local, T, O, num_regions=gr).to(dev)
xp = torch.rand(sh.topo.x_rows, T, F, device=dev)
y = torch.rand(n_local, O, device=dev)
inv = 1.0 / float(gn * O)

# the step of bench.py / train.py: library MSE (value + gradient in one kernel), gradient accumulation inside the model's backward
# (REGT_ACC_IN_BACKWARD=0: autograd's per-parameter adds and the torch loss expression, for A/B)
ACC = os.environ.get("REGT_ACC_IN_BACKWARD", "1") != "0"
R.functional.set_grad_accumulation_in_backward(ACC)


def step():
    pred, _ = model.forward_packed(xp, sh.graph)
    loss = R.functional.mse_loss(pred, y, gn * O) if ACC else ((pred - y) ** 2).sum() * inv
    loss.backward()
    return loss

loss = None
for _ in range(5):
    loss = step()
torch.cuda.synchronize()
K = 20
lib.regt_profile_enable(1)
t0 = time.perf_counter()
for _ in range(K):
    loss = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.regt_profile_enable(0)
buf = (ctypes.c_char * 16384)()
_lib.check(lib.regt_profile_collect(buf, 16384), "regt_profile_collect")
tot = sum(float(l.split()[2]) for l in buf.value.decode().splitlines()) / K
print(f"{1e3 * dt / K:.3f} ms/step wall (with profiling events), {tot:.3f} ms/step summed device stages")
for line in sorted(buf.value.decode().splitlines(), key=lambda l: -float(l.split()[2]))[:24]:
    name, cnt, ms = line.split()
    print(f"  {name:18s} {float(ms) / K:8.3f} ms/step")
# the same loop without per-stage events (what bench.py's wall clock sees) and the host's enqueue time per step
for _ in range(3):
    loss = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    loss = step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"{1e3 * dt / K:.3f} ms/step wall without events; host enqueue {1e3 * t_enq / K:.3f} ms/step")
