#!/usr/bin/env python3
"""Where the HOST time of a TPIMS-scale step goes (module / autograd path): cProfile over 300 steps, top functions."""
import cProfile, os, pstats, sys, time
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


for i in range(30):
    step(i)
torch.cuda.synchronize()
K = 300
def timed(label):
    t0 = time.perf_counter()
    for i in range(K):
        step(i)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{label}: host enqueue {1e3 * t_host / K:.3f} ms/step, wall {1e3 * t_all / K:.3f} ms/step")
timed("autograd engine threads (default)")
with torch.autograd.set_multithreading_enabled(False):      # backward on the calling thread: no hand-over to the device's engine thread
    timed("backward on the calling thread")
timed("autograd engine threads (default)")
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    step(i)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
