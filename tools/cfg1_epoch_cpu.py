#!/usr/bin/env python3
"""the reference CPU path (oracle = op-for-op restatement, run.py:163-226 loop semantics) timed over one EPOCH
of the TPIMS fixture -- every window of the fixture, train split (forward + loss + backward, one RMSprop step at the end) and
test split (forward only) at --tr 0.2 as in scripts/RegionalTemporalGCN.sh -- at T = 6 and T = 12, with the default thread count
and with one thread.  The authors' 14-day dataset has 2010 windows per epoch (402 train / 1608 test); the fixture holds 60
timesteps, so the per-snapshot rates below are what an epoch of any length costs.  Run on the GPU box's host cores:

    python tools/cfg1_epoch_cpu.py > profiles/r03_cfg1_cpu_epoch.txt
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import loop as L, model as M

z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tpims_fixture.npz"))
fx = {k: torch.from_numpy(z[k]) for k in z.files if z[k].ndim > 0}
regs = ("IA", "KS", "KY", "OH", "WI")
ri, rw = [fx[f"edge_{r}_index"] for r in regs], [fx[f"edge_{r}_attr"] for r in regs]
n = fx["node_data"].shape[0]
model = "unknown"
for line in open("/proc/cpuinfo"):
    if line.startswith("model name"):
        model = line.split(":", 1)[1].strip(); break
print(f"host: {model}; os.cpu_count() = {os.cpu_count()}; affinity = {len(os.sched_getaffinity(0))}; torch default threads = {torch.get_num_threads()}")
default_threads = torch.get_num_threads()
for t_in in (6, 12):
    xs, ys = L.make_windows(fx["node_data"], t_in, 1)
    (tx, ty), (vx, vy) = L.split(xs, ys, 0.2)
    for threads in (default_threads, 1):
        torch.set_num_threads(threads)
        p = {k: v.clone().requires_grad_(True) for k, v in M.init_params("RegionalTemporalGCN", 8, t_in, 1, num_nodes=n, seed=0).items()}
        opt = torch.optim.RMSprop(list(p.values()), lr=1e-3, weight_decay=1e-4)
        fwd = lambda q, x: M.regional_temporal_gcn(q, x, fx["edge_index"], ri, rw)
        L.train_epoch(p, fwd, tx[:2], ty[:2], opt)                      # warm-up
        t0 = time.perf_counter(); L.train_epoch(p, fwd, tx, ty, opt); t_train = time.perf_counter() - t0
        t0 = time.perf_counter(); L.evaluate(p, fwd, vx, vy); t_test = time.perf_counter() - t0
        tr, te = len(tx) / t_train, len(vx) / t_test
        full = 402 / tr + 1608 / te
        print(f"T={t_in:2d} O=1 threads={threads:3d}: train {len(tx)} snapshots in {t_train:6.2f} s = {tr:6.1f} snapshots/s (fwd+bwd); "
              f"test {len(vx)} snapshots in {t_test:6.2f} s = {te:6.1f} snapshots/s (fwd); a 402 + 1608 epoch = {full:6.1f} s")
