#!/usr/bin/env python3
"""A/B of two builds of libregtgcn_hip.so on the same seeded problem: python tools/ab_libs.py DIR_A[:ENV=v,..] DIR_B[:ENV=v,..] [mode] [nodes]
Runs forward + backward of RegionalTemporalGCN in a subprocess per build (REGT_LIB_DIR) and prints, per output / gradient,
the largest absolute difference and the largest magnitude."""
import os, subprocess, sys, tempfile
import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import regtgcn_amd as R
    mode, nodes, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    dev = torch.device("cuda")
    R.load_library().regt_set_gemm_mode(mode)
    F, T, O, regions = int(os.environ.get("AB_F", "8")), 12, 1, 4
    g = R.data.synthetic_regional_graph(nodes, nodes * 8, regions, seed=3)
    graph = R.prepare_graph(g.edge_index.to(dev), None, [t.to(dev) for t in g.region_index], [t.to(dev) for t in g.region_attr], nodes)
    torch.manual_seed(5)
    x = torch.rand(nodes, F, T, device=dev)
    model = R.RegionalTemporalGCN(F, nodes, T, O, num_regions=regions).to(dev)
    y = torch.rand(nodes, O, device=dev)
    pred, hid = model.forward_prepared(x, graph)
    (((pred - y) ** 2).sum() / nodes).backward()
    res = {"pred": pred.detach().cpu().numpy(), "hidden": hid.detach().cpu().numpy()}
    for n, p in model.named_parameters():
        if p.grad is not None:
            res["g:" + n] = p.grad.detach().cpu().numpy()
    np.savez(out, **res)
    sys.exit(0)

a, b = sys.argv[1], sys.argv[2]
mode = sys.argv[3] if len(sys.argv) > 3 else "0"
nodes = sys.argv[4] if len(sys.argv) > 4 else "300"
outs = []
for spec in (a, b):
    d, _, extra = spec.partition(":")
    f = tempfile.mktemp(suffix=".npz")
    env = dict(os.environ, REGT_LIB_DIR=os.path.abspath(d))
    for kv in filter(None, extra.split(",")):
        env[kv.split("=")[0]] = kv.split("=")[1]
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", mode, nodes, f], env=env)
    outs.append(np.load(f))
for k in outs[0].files:
    x, y = outs[0][k], outs[1][k]
    print(f"{k:60s} max|a-b| {np.abs(x - y).max():.3e}   max|a| {np.abs(x).max():.3e}")
