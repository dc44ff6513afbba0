#!/usr/bin/env python3
"""Which phases of which workgroups overlap on a CU?  Developer tool for the flat GEMM kernels.

    REGT_LIB_DIR=$PWD/regt-gcn_amd/lib_trace REGT_HIPCC_FLAGS="-DREGT_WG_TRACE -DREGT_WG_TRACE_N=512" python regt-gcn_amd/build.py
    REGT_LIB_DIR=$PWD/regt-gcn_amd/lib_trace python tools/wg_trace.py [gemm mode 0|1|2] [N of the traced GEMM, default 512]

The trace build makes every flat GEMM workgroup whose N matches record the 100 MHz wall clock at the start of its K loop,
at the end of it and after its epilogue, plus HW_ID/XCC_ID.  This script runs a few cfg-3 steps, reads the table of the LAST
traced launch and prints, over all 256 CUs, the fraction of CU-time with 0 / 1 / >= 2 workgroups inside their K loop (the
matrix pipe is only fed from there), the mean loop / epilogue / turn-around durations and one CU's timeline.
The trace build lives in its own directory (git-ignored); the product library is untouched.
"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import regtgcn_amd as R

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda")
lib = R.load_library()
raw = ctypes.CDLL(R._lib.LIB_PATH)
if not hasattr(raw, "regt_wg_trace_read"):
    raise SystemExit("libregtgcn_hip.so was built without -DREGT_WG_TRACE (see the docstring)")
nodes, edges, regions, F, T, O = 100_000, 1_000_000, 8, 32, 12, 1
g = R.data.synthetic_regional_graph(nodes, edges, regions, seed=42)
graph = R.prepare_graph(g.edge_index.to(dev), None, [t.to(dev) for t in g.region_index], [t.to(dev) for t in g.region_attr], nodes)
x = torch.rand(nodes, F, T, device=dev)
model = R.RegionalTemporalGCN(F, nodes, T, O, num_regions=regions).to(dev)
y = torch.rand(nodes, O, device=dev)
lib.regt_set_gemm_mode(mode)
FWD_ONLY = os.environ.get("WG_TRACE_FWD_ONLY", "0") != "0"      # forward only: with N = 256 the last traced launch is then gemm_regional
for _ in range(4):
    if FWD_ONLY:
        with torch.no_grad():
            model.forward_prepared(x, graph)
        continue
    pred, _ = model.forward_prepared(x, graph)
    (((pred - y) ** 2).sum() / nodes).backward()
torch.cuda.synchronize()
ntile = min(40000, -(-nodes * T // 128) * (int(sys.argv[2]) if len(sys.argv) > 2 else 512) // 128)
buf = np.zeros(4 * ntile, dtype=np.int64)
rc = raw.regt_wg_trace_read(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(ntile))
tr = buf.reshape(-1, 4)
tr = tr[tr[:, 2] > 0]
ta, tb, tc, hw = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
t0 = ta.min()
ta, tb, tc = ta - t0, tb - t0, tc - t0
print(f"rc {rc}; {len(tr)} workgroups; kernel span {tc.max() / 100:.1f} us; K loop mean {(tb - ta).mean() / 100:.2f} us, epilogue mean {(tc - tb).mean() / 100:.2f} us")
if hasattr(raw, "regt_wg_marks_read"):
    mb = np.zeros(8 * ntile, dtype=np.int64)
    raw.regt_wg_marks_read(mb.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(ntile))
    mk = mb.reshape(-1, 8)[buf.reshape(-1, 4)[:, 2] > 0] - t0
    rel = (mk[:, :6] - tb[:, None]) / 100.0
    print("epilogue marks after the K loop's end (us, mean over workgroups; thread 0): per 64-row half [aux requested, "
          "accumulators staged, rows applied]:", np.round(rel.mean(axis=0), 2), " end:", round(float((tc - tb).mean()) / 100, 2))
    if mk[:, 6].max() > 0:
        print(f"kernel entry -> K loop start (arguments, iteration table, barriers): mean {(ta - mk[:, 6]).mean() / 100:.2f} us"
              f" (of which up to the iteration-table build: {(mk[:, 7] - mk[:, 6]).mean() / 100:.2f} us)")
hwid, xcc = hw & 0xffffffff, (hw >> 32) & 0xf
key = (((xcc * 8 + ((hwid >> 13) & 7)) * 2 + ((hwid >> 12) & 1)) * 16 + ((hwid >> 8) & 0xf))
cus = np.unique(key)
span = int(tc.max()) + 1
hist = np.zeros(8)
epi_any = 0
gaps = []
for k in cus:
    idx = np.where(key == k)[0]
    nl = np.zeros(span + 1, dtype=np.int32)
    ne = np.zeros(span + 1, dtype=np.int32)
    np.add.at(nl, ta[idx], 1); np.add.at(nl, tb[idx], -1)
    np.add.at(ne, tb[idx], 1); np.add.at(ne, tc[idx], -1)
    nl = np.cumsum(nl)[:span]
    hist += np.bincount(np.minimum(nl, 7), minlength=8)
    epi_any += np.count_nonzero(np.cumsum(ne)[:span] > 0)
    # turn-around: from a workgroup's end to the next K-loop start on the same CU
    ends, starts = np.sort(tc[idx]), np.sort(ta[idx])
    j = np.searchsorted(starts, ends, side="left")
    ok = j < len(starts)
    gaps.append((starts[j[ok]] - ends[ok]))
hist /= hist.sum()
print(f"{len(cus)} CUs; CU-time by workgroups inside the K loop: " + ", ".join(f"{i}: {hist[i]:.3f}" for i in range(5)) +
      f"; >= 1 in epilogue: {epi_any / (len(cus) * span):.3f}; end -> next loop start: median {np.median(np.concatenate(gaps)) / 100:.2f} us")
k = cus[len(cus) // 2]
idx = np.where(key == k)[0]
for i in idx[np.argsort(ta[idx])[:12]]:
    print(f"  loop {ta[i] / 100:8.2f} .. {tb[i] / 100:8.2f} us  epilogue .. {tc[i] / 100:8.2f}  (wave slot {hwid[i] & 0xf}, simd {(hwid[i] >> 4) & 3})")
