#!/usr/bin/env python3
"""Phase durations inside the fused forward kernel (csrc/fused.hip): REGT_FUSED_TRACE=1 python tools/fused_trace.py
Prints, over all 64-row tiles of one launch at the cfg-5 shard shape, the mean / median shader-clock cycles between the stamps."""
import ctypes, os, sys
os.environ["REGT_FUSED_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import regtgcn_amd as R
lib = R.load_library()
lib.regt_set_gemm_mode(2)
n, e, regions, f, t, o = 40000, 400000, 8, 64, 12, 1
g = R.data.synthetic_regional_graph(n, e, regions, seed=1)
dev = torch.device("cuda")
model = R.RegionalTemporalGCN(f, n, t, o, num_regions=regions).to(dev)
graph = R.prepare_graph(g.edge_index.to(dev), None, [x.to(dev) for x in g.region_index], [x.to(dev) for x in g.region_attr], n)
(x, y), = R.data.synthetic_snapshots(n, f, t, o, 1, seed=1)
x = x.to(dev)
for _ in range(3):
    with torch.no_grad():
        model.forward_prepared(x, graph)
torch.cuda.synchronize()
tiles = (n * t + 63) // 64
buf = (ctypes.c_int64 * (8 * tiles))()
got = lib.regt_debug_trace(buf, 8 * tiles)
a = np.frombuffer(buf, dtype=np.int64)[:got].reshape(-1, 8)
d = np.diff(a, axis=1)
names = ["tables", "h (2 tiles)", "R, q (2 tiles)", "Z_0", "cand_0", "Z_1", "cand_1"]
print(f"{a.shape[0]} tiles; cycles per phase (mean / median / p90):")
for i, nm in enumerate(names):
    print(f"  {nm:16s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {np.percentile(d[:, i], 90):9.0f}")
tot = a[:, 7] - a[:, 0]
print(f"  {'tile total':16s} {tot.mean():9.0f} {np.median(tot):9.0f} {np.percentile(tot, 90):9.0f}")
