#!/usr/bin/env python3
"""Step time and per-stage device time of one workload under each GEMM arithmetic (0 fp32 MFMA, 1 bf16x3 split, 2 bf16):
    python tools/mode_bench.py [cfg3|cfg5shard] [modes, e.g. 0,2] [steps] [option=value ...]
cfg5shard = rank 0's share of BASELINE configs[4] (1M nodes / 10M edges / 64 regions / F=64 over 8 GPUs: 125k nodes, 8 of
the 64 regions, + halo rows filled with random data; no communication)."""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R
from regtgcn_amd import _lib

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
modes = [int(m) for m in (sys.argv[2] if len(sys.argv) > 2 else "0,1,2").split(",")]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda")
lib = R.load_library()
for kv in sys.argv[4:]:                                         # runtime options, e.g. fused_rows=2
    name, val = kv.split("=")
    assert lib.regt_set_option(name.encode(), int(val)) >= 0, kv
T, O = 12, 1
if wl == "cfg5shard":
    world, F = 8, 64
    gn, ge, gr = 1_000_000, 10_000_000, 64
    g = R.data.synthetic_regional_graph(gn, ge, gr, seed=42)
    rpg = gr // world
    bounds = np.asarray(g.region_bounds[::rpg], dtype=np.int64)
    sh = R.dist.build_shard(g.edge_index, g.region_index, g.region_attr, gn, bounds, [r // rpg for r in range(gr)], 0, world, dev)
    graph, nodes, regions = sh.graph, sh.topo.n_local, gr
    x = torch.rand(sh.topo.x_rows, T, F, device=dev)
    print(f"cfg5shard: {nodes} local rows + {sh.topo.halo_rows} halo rows, merged nnz {graph.m_col.numel()}, regions {graph.region_lo}..{graph.region_hi} of {gr}")
    run = lambda model: model.forward_packed(x, graph)
    inv = 1.0 / float(gn * O)
else:
    nodes, edges, regions, F = 100_000, 1_000_000, 8, 32
    g = R.data.synthetic_regional_graph(nodes, edges, regions, seed=42)
    graph = R.prepare_graph(g.edge_index.to(dev), None, [t.to(dev) for t in g.region_index], [t.to(dev) for t in g.region_attr], nodes)
    x = torch.rand(nodes, F, T, device=dev)
    run = lambda model: model.forward_prepared(x, graph)
    inv = 1.0 / float(nodes * O)
torch.manual_seed(42)
model = R.RegionalTemporalGCN(F, nodes, T, O, num_regions=regions).to(dev)
y = torch.rand(nodes, O, device=dev)


def step():
    pred, _ = run(model)
    loss = ((pred - y) ** 2).sum() * inv
    loss.backward()
    return loss


for mode in modes:
    lib.regt_set_gemm_mode(mode)
    loss = None
    for _ in range(3):
        loss = step()
    torch.cuda.synchronize()
    lib.regt_profile_enable(1)
    t0 = time.perf_counter()
    for _ in range(K):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    lib.regt_profile_enable(0)
    buf = (ctypes.c_char * 16384)()
    _lib.check(lib.regt_profile_collect(buf, 16384), "regt_profile_collect")
    print(f"mode {mode}: {1e3 * dt / K:.3f} ms/step  loss {float(loss):.6f}")
    for line in sorted(buf.value.decode().splitlines(), key=lambda l: -float(l.split()[2]))[:22]:
        name, cnt, ms = line.split()
        print(f"  {name:18s} {float(ms) / K:8.3f} ms/step")
lib.regt_set_gemm_mode(0)
