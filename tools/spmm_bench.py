#!/usr/bin/env python3
"""SpMM micro-benchmark at the cfg-3 shape (dual-operator and stacked variants)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R
nodes, edges, regions, W = 100_000, 1_000_000, 8, 384
g = R.data.synthetic_regional_graph(nodes, edges, regions, seed=42)
dev = torch.device("cuda")
pg = R.prepare_graph(g.edge_index.to(dev), None, [t.to(dev) for t in g.region_index], [t.to(dev) for t in g.region_attr], nodes)
x = torch.rand(nodes, W, device=dev)
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def t_cold(fn, n=15):
    """each launch timed on its own after 1 GB of unrelated writes (what the kernel sees inside a training step)."""
    junk = torch.empty(256 << 20, dtype=torch.float32, device=dev)
    tot = 0.0
    for i in range(n + 3):
        junk.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 3: tot += e0.elapsed_time(e1)
    return tot / n
ya = torch.empty(nodes, W, device=dev); yl = torch.empty_like(ya)
from regtgcn_amd import _lib
lib = _lib.load(); st = torch.cuda.current_stream().cuda_stream
dual = lambda: _lib.check(lib.regt_spmm_dual(_lib.ptr(pg.m_rowptr), _lib.ptr(pg.m_col), _lib.ptr(pg.m_val_a), _lib.ptr(pg.m_val_l), _lib.ptr(x), _lib.ptr(ya), _lib.ptr(yl), nodes, W, st), "dual")
nnz = pg.m_col.numel()
algo = nodes*W*4 + nnz*12 + (nodes+1)*4 + 2*nodes*W*4
for rows in (1, 0):          # row-block kernel (CSR entries in LDS) vs column-panel kernel
    lib.regt_set_option(b"spmm_rows", rows)
    ms = t(dual)
    ms_cold = t_cold(dual)
    print(f"dual  rows={rows} PL={os.environ.get('REGT_SPMM_PL','auto')}: {ms*1e3:7.1f} us  {algo/ms/1e6:7.1f} GB/s algorithmic ({algo/1e6:.1f} MB)  nnz={nnz}   "
          f"cold: {ms_cold*1e3:7.1f} us = {algo/ms_cold/1e6/8000:.3f} of 8 TB/s")
lib.regt_set_option(b"spmm_rows", 0)
# bf16 rows (REGT_GEMM_MODE=bf16, cfg-5 layout): same row bytes at twice the feature count
xb = torch.rand(nodes, 2 * W, device=dev).to(torch.bfloat16)
yab = torch.empty_like(xb); ylb = torch.empty_like(xb)
dualb = lambda: _lib.check(lib.regt_spmm_dual_bf16(_lib.ptr(pg.m_rowptr), _lib.ptr(pg.m_col), _lib.ptr(pg.m_val_a), _lib.ptr(pg.m_val_l), _lib.ptr(xb), _lib.ptr(yab), _lib.ptr(ylb), nodes, nodes, 2 * W, st), "dual bf16")
ms = t(dualb); ms_cold = t_cold(dualb)
print(f"dual  bf16 rows W={2*W}: {ms*1e3:7.1f} us  {algo/ms/1e6:7.1f} GB/s algorithmic   cold: {ms_cold*1e3:7.1f} us = {algo/ms_cold/1e6/8000:.3f} of 8 TB/s")


# ---- reference-faithful WIDE aggregation: a learned hidden state of width T*512 (ConvStackedTemporalGCN layers 2-5, models/
# ConvStackedTemporalGCN.py:115-126), forward over A_hat and backward over A_hat^T.  Algorithmic bytes: read H once + CSR + write.
if len(sys.argv) > 1 and sys.argv[1] == "wide":
    Wd = 12 * 512
    op = R.graph.prepare_gcn_operator(g.edge_index.to(dev), g.edge_attr.to(dev), nodes)
    h = torch.rand(nodes, Wd, device=dev)
    out = torch.empty_like(h)
    for rows, name, (rp, cl, vl) in ((1, "A_hat  ", (op.rowptr, op.col, op.val)), (1, "A_hat^T", (op.t_rowptr, op.t_col, op.t_val)),
                                     (0, "A_hat  ", (op.rowptr, op.col, op.val))):
        lib.regt_set_option(b"spmm_rows", rows)
        name = f"{name} rows={rows}"
        fn = lambda: _lib.check(lib.regt_spmm_csr(_lib.ptr(rp), _lib.ptr(cl), _lib.ptr(vl), _lib.ptr(h), _lib.ptr(out), nodes, nodes, Wd, st), "spmm")
        ms = t(fn, 10)
        nz = cl.numel()
        algo = 2 * nodes * Wd * 4 + nz * 8 + (nodes + 1) * 4
        print(f"wide {name} W={Wd} PL={os.environ.get('REGT_SPMM_PL','auto')}: {ms:7.3f} ms  {algo/ms/1e6:7.1f} GB/s algorithmic ({algo/1e9:.2f} GB) = {algo/ms/1e6/8000:.3f} of 8 TB/s; "
              f"gathered rows {nz*Wd*4/1e9:.1f} GB -> {nz*Wd*4/ms/1e6:.0f} GB/s through L2")
    ref = torch.sparse_csr_tensor(op.rowptr.long(), op.col.long(), op.val, (nodes, nodes)) if False else None
