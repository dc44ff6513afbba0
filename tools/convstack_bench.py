#!/usr/bin/env python3
"""Step time of ConvStackedTemporalGCN on a synthetic graph, with the library's per-stage HIP-event
timers for the cell part and torch events around the conv stack.

    python tools/convstack_bench.py [nodes edges F T]        (default: 100000 1000000 32 12 = cfg-3 shape)
"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R
from regtgcn_amd import _lib

nodes, edges, F, T = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (100_000, 1_000_000, 32, 12)
COLLAPSE = os.environ.get("REGT_CONVSTACK_COLLAPSE", "1") != "0"      # 0: layer by layer, hidden state aggregated at width T*512
lib = R.load_library()
dev = torch.device("cuda")
g = R.data.synthetic_regional_graph(nodes, edges, 8, seed=42)
torch.manual_seed(42)
model = R.ConvStackedTemporalGCN(F, T, 1).to(dev)
with torch.no_grad():                       # five un-normalised 512-wide layers: keep activations O(1)
    for layer in range(2, 6):
        getattr(model.tgnn, f"conv{layer}").lin.weight.mul_(0.5)
model.collapse = COLLAPSE
op = model.prepare_graph(g.edge_index.to(dev), g.edge_attr.to(dev), nodes)
snaps = [(x.to(dev), y.to(dev)) for x, y in R.data.synthetic_snapshots(nodes, F, T, 1, 2, seed=42)]
opt = torch.optim.RMSprop(model.parameters(), lr=1e-3, weight_decay=1e-4)


def step(i):
    x, y = snaps[i % 2]
    pred, _ = model.forward_prepared(x, op)
    loss = torch.mean((pred - y) ** 2)
    loss.backward()
    return loss


for i in range(2):
    loss = step(i)
opt.step(); opt.zero_grad(set_to_none=False)
torch.cuda.synchronize()
K = 5
lib.regt_profile_enable(1)
t0 = time.perf_counter()
for i in range(K):
    loss = step(i)
opt.step(); opt.zero_grad(set_to_none=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
lib.regt_profile_enable(0)
buf = (ctypes.c_char * 16384)()
_lib.check(lib.regt_profile_collect(buf, 16384), "regt_profile_collect")
M, C = nodes * T, 512
nnz = int(op.col.numel())
cell = 2.0 * M * (2 * C) * (C + F) * 3 + 2.0 * M * C * (C + F) * 3
flops = (2.0 * M * C * (F + 8) * 2 + cell) if COLLAPSE else (2.0 * M * C * F * 3 + 4 * 2.0 * M * C * C * 3 + cell)
print(f"ConvStackedTemporalGCN ({'collapsed conv stack' if COLLAPSE else 'layer by layer'})  N={nodes} E={edges} F={F} T={T}: {1e3 * dt / K:.1f} ms/step  ({K / dt:.2f} snapshots/s), "
      f"loss {float(loss):.4f}, ~{flops / 1e12:.1f} TFLOP/step dense => {flops / (dt / K) / 1e12:.0f} TFLOP/s overall; "
      f"{'aggregations: 5 of the input at width T*F' if COLLAPSE else f'hidden-state aggregation: 8 SpMMs of {nnz * T * C * 4 / 1e9:.1f} GB gathered each'}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
for line in buf.value.decode().splitlines():
    name, cnt, ms = line.split()
    print(f"  cell stage {name:18s} {float(ms) / K:8.3f} ms/step")
