"""Golden vectors for the sliding-window construction, produced by the REFERENCE's own ``TruckParkingDataset2.get()``
(load_dataset.py:442-471).

TEST INFRASTRUCTURE; runs only in the build container (needs ``/root/reference``).  ``load_dataset.py`` is imported unmodified;
its two absent third-party imports resolve to stand-ins registered here: ``torch_geometric.data.{Data, Dataset}`` (plain base
classes -- get() never touches them) and ``torch_geometric_temporal...StaticGraphTemporalSignal`` (a container that keeps the
``features`` / ``targets`` lists get() hands it, which is all the real class does with them before iteration).  The method is
called on a bare instance (``object.__new__``) carrying the attributes it reads: ``processed_root``, ``data_size``, ``sc``,
``max_list``, ``min_list``; the processed 13-tuple pickle it loads (load_dataset.py:436-437) is written from the TPIMS fixture
(``tests/golden/tpims_fixture.npz``) in the reference's own layout: a list of per-timestep (N, 8) float64 tensors.

Output: tests/golden/golden_windows.npz -- for (T_in, T_out) in {(6, 1), (12, 3), (24, 12) = get()'s defaults}: the window count,
three whole windows (first, middle, last) of features and targets, and two float64 checksums per window over ALL windows.
"""
from __future__ import annotations

import os
import sys
import tempfile
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
REGIONS = ("IA", "KS", "KY", "OH", "WI")


class _Signal:
    def __init__(self, edge_index, edge_weight, features, targets, **kw):
        self.edge_index, self.edge_weight, self.features, self.targets = edge_index, edge_weight, features, targets


def install_standins():
    tg = types.ModuleType("torch_geometric")
    tgd = types.ModuleType("torch_geometric.data")
    tgd.Data = type("Data", (), {})
    tgd.Dataset = type("Dataset", (), {})
    tg.data = tgd
    tgt = types.ModuleType("torch_geometric_temporal")
    sig = types.ModuleType("torch_geometric_temporal.signal")
    sgs = types.ModuleType("torch_geometric_temporal.signal.static_graph_temporal_signal")
    sgs.StaticGraphTemporalSignal = _Signal
    sig.static_graph_temporal_signal = sgs
    tgt.signal = sig
    sys.modules.update({"torch_geometric": tg, "torch_geometric.data": tgd, "torch_geometric_temporal": tgt,
                        "torch_geometric_temporal.signal": sig,
                        "torch_geometric_temporal.signal.static_graph_temporal_signal": sgs})


def checksums(arrs):
    """per window: plain sum and a position-weighted sum (float64) -- catches shifted / transposed / truncated windows"""
    out = np.zeros((len(arrs), 2))
    for i, a in enumerate(arrs):
        a = np.asarray(a, dtype=np.float64)
        w = np.arange(1, a.size + 1, dtype=np.float64).reshape(a.shape)
        out[i] = (a.sum(), (a * w).sum() / a.size)
    return out


def main():
    install_standins()
    sys.path.insert(0, REF)
    import load_dataset  # noqa  (the reference's file, unmodified)
    z = np.load(os.path.join(OUT, "tpims_fixture.npz"))
    fx = {k: torch.from_numpy(z[k]) for k in z.files if z[k].ndim > 0}
    node = fx["node_data"].double()                                    # (N, 8, steps)
    per_step = [node[:, :, t].contiguous() for t in range(node.shape[2])]
    tup = [fx["edge_index"], fx["edge_attr"]]
    for r in REGIONS:
        tup += [fx[f"edge_{r}_index"], fx[f"edge_{r}_attr"]]
    tup.append(per_step)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        torch.save(tuple(tup), os.path.join(tmp, "tpims_data_small.pkl"))
        ds = object.__new__(load_dataset.TruckParkingDataset2)
        ds.processed_root, ds.data_size, ds.sc, ds.max_list, ds.min_list = tmp, "small", None, None, None
        real_load = torch.load
        torch.load = lambda f, *a, **k: real_load(f, *a, **{**k, "weights_only": False})   # the 13-tuple holds lists (torch >= 2.6 default refuses)
        try:
            for t_in, t_out in ((6, 1), (12, 3), (24, 12)):
                res = load_dataset.TruckParkingDataset2.get(ds, num_timesteps_in=t_in, num_timesteps_out=t_out)
                data = res[0]
                feats, targs = data.features, data.targets
                n = len(feats)
                assert n == len(targs) == node.shape[2] - (t_in + t_out) + 1
                pick = [0, n // 2, n - 1]
                tag = f"in{t_in}_out{t_out}"
                out[f"{tag}_count"] = np.int64(n)
                out[f"{tag}_pick"] = np.asarray(pick)
                out[f"{tag}_features"] = np.stack([np.asarray(feats[i]) for i in pick]).astype(np.float32)
                out[f"{tag}_targets"] = np.stack([np.asarray(targs[i]) for i in pick]).astype(np.float32)
                out[f"{tag}_feature_sums"] = checksums(feats)
                out[f"{tag}_target_sums"] = checksums(targs)
                print(tag, "windows", n, "feature window", np.asarray(feats[0]).shape, "target window", np.asarray(targs[0]).shape)
        finally:
            torch.load = real_load
    np.savez_compressed(os.path.join(OUT, "golden_windows.npz"), **out)
    print("written", os.path.join(OUT, "golden_windows.npz"))


if __name__ == "__main__":
    main()
