"""Host-side data formats next to the path: the reference's processed 13-tuple pickle and snapshot windows."""
import torch

import regtgcn_amd as R
from oracle import loop as oloop


def test_processed_pickle_roundtrip(tmp_path, tpims):
    regs = ("IA", "KS", "KY", "OH", "WI")
    nd = tpims["node_data"][:, :, :9].double()
    node_list = [nd[:, :, t].clone() for t in range(nd.shape[2])]          # per-timestep (N, 8) float64
    tup = (tpims["edge_index"], tpims["edge_attr"]) + tuple(
        v for r in regs for v in (tpims[f"edge_{r}_index"], tpims[f"edge_{r}_attr"])) + (node_list,)
    path = tmp_path / "tpims_data_small.pkl"
    torch.save(tup, path)
    d = R.evaluate.load_processed_pickle(str(path))
    assert torch.equal(d["node_data"], tpims["node_data"][:, :, :9])
    assert torch.equal(d["edge_KY_index"], tpims["edge_KY_index"]) and torch.equal(d["edge_attr"], tpims["edge_attr"])


def test_snapshot_windows_match_reference_slicing(tpims):
    xs, ys = R.data.snapshot_windows(tpims["node_data"][:, :, :20], 6, 3)
    xo, yo = oloop.make_windows(tpims["node_data"][:, :, :20], 6, 3)
    assert len(xs) == len(xo) == 20 - 9 + 1
    for a, b, c, d in zip(xs, xo, ys, yo):
        assert torch.equal(a, b) and torch.equal(c, d)
    (tx, _), (vx, _) = R.train.split(xs, ys, 0.2)
    assert len(tx) == int(0.2 * len(xs)) and len(tx) + len(vx) == len(xs)
