#!/usr/bin/env python3
"""TPIMS-scale (N=104) step rate: launch-latency regime (BASELINE configs[1])."""
import os, sys, time
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
def step(i):
    pred, _ = model.forward_prepared(xs[i % len(xs)], graph)
    loss = torch.mean((pred - ys[i % len(xs)]) ** 2)
    loss.backward()
for i in range(20): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 300
for i in range(K): step(i)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"TPIMS N={n} T={T}: {1e3*dt/K:.3f} ms/step  {K/dt:.1f} snapshots/s")

import ctypes
from regtgcn_amd import _lib
st = (ctypes.c_int64 * 6)()
_lib.load().regt_graph_stats(st)
print("graph stats fwd eager/captured/replayed:", st[0], st[1], st[2], " bwd:", st[3], st[4], st[5])

# per-stage device time (HIP events inside the library; graphs are bypassed while profiling)
lib = _lib.load()
lib.regt_profile_enable(1)
for i in range(50): step(i)
torch.cuda.synchronize()
lib.regt_profile_enable(0)
buf = (ctypes.c_char * 16384)()
lib.regt_profile_collect(buf, 16384)
rows = [l.split() for l in buf.value.decode().splitlines()]
tot = 0.0
for name, cnt, ms in sorted(rows, key=lambda r: -float(r[2])):
    per = float(ms) / 50 * 1e3
    tot += per
    print(f"  {name:18s} {int(cnt)//50:3d} launches/step  {per:8.1f} us/step")
print(f"  sum {tot:.1f} us/step")
