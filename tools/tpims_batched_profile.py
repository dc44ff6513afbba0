#!/usr/bin/env python3
"""Per-stage device time of a snapshot-batched TPIMS training step (train.train_epoch_batched): python tools/tpims_batched_profile.py [B=64]"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R
from regtgcn_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = R.load_library()
dev = torch.device("cuda")
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tpims_fixture.npz"))
fx = {k: torch.from_numpy(z[k]) for k in z.files if z[k].ndim > 0}
regs = ("IA", "KS", "KY", "OH", "WI")
n, T, O = fx["node_data"].shape[0], 12, 1
torch.manual_seed(42)
model = R.RegionalTemporalGCN(8, n, T, O).to(dev)
ei, ri, rw = fx["edge_index"].to(dev), [fx[f"edge_{r}_index"].to(dev) for r in regs], [fx[f"edge_{r}_attr"].to(dev) for r in regs]
graphs = R.train.BatchedGraphs(lambda b: model.prepare_graph(ei, ri, rw, copies=b))
xs, ys = R.data.snapshot_windows(fx["node_data"], T, O)
xs, ys = [x.to(dev) for x in xs], [y.to(dev) for y in ys]
reps = (4 * B + len(xs) - 1) // len(xs)
store = R.train.WindowStore((xs * reps)[:4 * B], (ys * reps)[:4 * B])
opt = torch.optim.RMSprop(model.parameters(), lr=1e-3, weight_decay=1e-4)
for _ in range(2):
    R.train.train_epoch_batched(model, store, graphs, opt, B)
torch.cuda.synchronize()
K = 5
lib.regt_profile_enable(1)
t0 = time.perf_counter()
for _ in range(K):
    R.train.train_epoch_batched(model, store, graphs, opt, B)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.regt_profile_enable(0)
buf = (ctypes.c_char * 16384)()
_lib.check(lib.regt_profile_collect(buf, 16384), "regt_profile_collect")
steps = K * 4
print(f"B = {B}: {1e3 * dt / steps:.3f} ms per batched step ({B * steps / dt:.0f} snapshots/s), M = {B * n * T} rows")
tot = 0.0
for line in sorted(buf.value.decode().splitlines(), key=lambda l: -float(l.split()[2]))[:24]:
    name, cnt, ms = line.split()
    tot += float(ms) / steps
    print(f"  {name:18s} {float(ms) / steps:8.3f} ms/step")
print(f"  sum of the listed stages {tot:.3f} ms")
