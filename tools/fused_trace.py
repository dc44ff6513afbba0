#!/usr/bin/env python3
"""Phase durations inside the fused kernels (csrc/fused.hip): python tools/fused_trace.py [fwd|bwd|rows] [option=value ...]
Prints, over all 64-row tiles of one launch, the mean / median shader-clock cycles between the stamps."""
import ctypes, os, sys
BWD = len(sys.argv) > 1 and sys.argv[1] == "bwd"
ROWS = len(sys.argv) > 1 and sys.argv[1] == "rows"      # the row-owning forward kernel (fused_rows.hip): 128-row tiles, 17 stamps
os.environ["REGT_FUSED_TRACE"] = "2" if BWD else "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import regtgcn_amd as R
lib = R.load_library()
lib.regt_set_gemm_mode(2)
for kv in sys.argv[2:]:                                         # runtime options, e.g. fused_rows=2
    name, val = kv.split("=")
    assert lib.regt_set_option(name.encode(), int(val)) >= 0, kv
n, e, regions, f, t, o = 40000, 400000, 8, 64, 12, 1
g = R.data.synthetic_regional_graph(n, e, regions, seed=1)
dev = torch.device("cuda")
model = R.RegionalTemporalGCN(f, n, t, o, num_regions=regions).to(dev)
graph = R.prepare_graph(g.edge_index.to(dev), None, [x.to(dev) for x in g.region_index], [x.to(dev) for x in g.region_attr], n)
(x, y), = R.data.synthetic_snapshots(n, f, t, o, 1, seed=1)
x = x.to(dev)
for _ in range(3):
    if BWD:
        pred, _h = model.forward_prepared(x, graph)
        pred.sum().backward()
    else:
        with torch.no_grad():
            model.forward_prepared(x, graph)
torch.cuda.synchronize()
tiles = (n * t + 63) // 64
if ROWS:
    tiles = (n * t + 127) // 128
    buf = (ctypes.c_int64 * (32 * tiles))()
    got = lib.regt_debug_trace(buf, 32 * tiles)
    f = np.frombuffer(buf, dtype=np.int64)[:got].reshape(-1, 32)
    units = ["embed j=0", "embed j=1", "R j=0", "R j=1", "Z j=0", "cand j=0", "Z j=1", "cand j=1"]
    print(f"{f.shape[0]} tiles of 128 rows; cycles (mean / median / p90):")
    for u, nm in enumerate(units):
        k = f[:, 1 + 2 * u] - f[:, 2 * u]
        e = f[:, 2 + 2 * u] - f[:, 1 + 2 * u]
        print(f"  {nm:10s} K loop {k.mean():8.0f} {np.median(k):8.0f} {np.percentile(k, 90):8.0f}   epilogue {e.mean():8.0f} {np.median(e):8.0f} {np.percentile(e, 90):8.0f}")
    tot = f[:, 16] - f[:, 0]
    print(f"  tile total {tot.mean():9.0f} {np.median(tot):9.0f} {np.percentile(tot, 90):9.0f}")
    sys.exit(0)
SLOTS = 32
buf = (ctypes.c_int64 * (SLOTS * tiles))()
got = lib.regt_debug_trace(buf, SLOTS * tiles)
full = np.frombuffer(buf, dtype=np.int64)[:got].reshape(-1, SLOTS)
a = full[:, :8]
d = np.diff(a, axis=1)
names = ["tables", "h (2 tiles)", "R, q (2 tiles)", "Z_0", "cand_0", "Z_1", "cand_1"]
if BWD:
    names = ["A: dhp, dzp", "B: dq_0 (+barrier)", "B: dq_1", "D: barrier, drp Ur_0", "D: ds_0 rest", "D: ds_1"]
    a = a[:, :8]
print(f"{a.shape[0]} tiles; cycles per phase (mean / median / p90):")
for i, nm in enumerate(names):
    print(f"  {nm:16s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {np.percentile(d[:, i], 90):9.0f}")
tot = a[:, 6 if BWD else 7] - a[:, 0]
if BWD:
    d = d[:, :6]
    names = names[:6]
print(f"  {'tile total':16s} {tot.mean():9.0f} {np.median(tot):9.0f} {np.percentile(tot, 90):9.0f}")

if not BWD and full[:, 8:].any():      # a -DREGT_FUSED_FINE build: stamps inside the phases (fused.hip FT_FINE)
    f = full
    def seg(name, i, j):
        dd = f[:, j] - f[:, i]
        print(f"  {name:34s} {dd.mean():9.0f} {np.median(dd):9.0f} {np.percentile(dd, 90):9.0f}")
    print("fine stamps (cycles: mean / median / p90):")
    seg("h: wait + MFMAs, tile 0", 1, 8); seg("h: epilogue 0 (4 rounds)", 8, 9); seg("h: MFMAs tile 1 (+ issue)", 9, 10); seg("h: epilogue 1", 10, 11)
    seg("barrier h", 2, 12)
    seg("R: K loop 0", 12, 13); seg("R: round 0", 13, 14); seg("R: round 1", 14, 15); seg("R: round 2", 15, 16); seg("R: round 3", 16, 17)
    seg("R: K loop 1", 17, 18); seg("R: epilogue 1", 18, 22)
    seg("barrier q", 3, 23)
    seg("Z0: K loop", 23, 24); seg("Z0: epilogue", 24, 4); seg("cand0: K loop", 4, 25)
    seg("cand0: round 0", 25, 26); seg("cand0: round 1", 26, 27); seg("cand0: round 2", 27, 28); seg("cand0: round 3", 28, 5)
