"""predict.py counterpart: MAE / RMSE / MAPE of a trained checkpoint on the test split.

Metric definitions follow predict.py:141-194: per test snapshot ``|y - out|``, ``(y - out)**2`` and
``|y - out| / percentile_95(y)`` (the snapshot is left out of the MAPE if that ratio is infinite anywhere); the
reported numbers are the means over all snapshots, nodes and horizon steps, RMSE = sqrt(mean squared error),
MAPE in percent.  Also reads the reference's processed 13-tuple pickle (load_dataset.py:436-437, 445-471) when the
authors' file is available.
"""
from __future__ import annotations

import argparse
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from . import nn as rnn
from .data import snapshot_windows
from .train import REGIONS, split


@torch.no_grad()
def predict_metrics(model, xs: Sequence[torch.Tensor], ys: Sequence[torch.Tensor], graph) -> Tuple[float, float, float]:
    """(MAE, RMSE, MAPE %) over the given snapshots; everything stays on the device until the final three scalars."""
    model.eval()
    ae, se, ape = [], [], []
    for x, y in zip(xs, ys):
        out, _ = model.forward_prepared(x, graph)
        d = (y - out).abs()
        ae.append(d)
        se.append((y - out) ** 2)
        r = d / torch.quantile(y.flatten().double(), 0.95).to(d.dtype)     # np.percentile(q=95): linear interpolation
        if not torch.isinf(r).any():
            ape.append(r)
    mae = float(torch.cat(ae, dim=0).mean())
    rmse = float(torch.cat(se, dim=0).mean().sqrt())
    mape = float(torch.cat(ape, dim=0).mean()) * 100 if ape else float("nan")
    return mae, rmse, mape


@torch.no_grad()
def predict_metrics_batched(model, store, graphs, snap_batch: int) -> Tuple[float, float, float]:
    """predict_metrics() with ``snap_batch`` snapshots per forward (train.WindowStore / train.BatchedGraphs): the same three means --
    the 95th percentile and the infinity rule stay PER SNAPSHOT, as in predict.py:141-194."""
    model.eval()
    ae = torch.zeros((), dtype=torch.float64, device=store.X.device)
    se = torch.zeros_like(ae)
    ape = torch.zeros_like(ae)
    ape_n = torch.zeros_like(ae)
    per = store.Y.shape[1] * store.Y.shape[2]
    for i in range(0, len(store), snap_batch):
        b = min(snap_batch, len(store) - i)
        x, y = store.batch(i, b)
        out, _ = model.forward_prepared(x, graphs.get(b))
        d = (y - out).abs().view(b, per)
        ae += d.sum(dtype=torch.float64)
        se += (d * d).sum(dtype=torch.float64)
        q = torch.quantile(y.view(b, per).double(), 0.95, dim=1, keepdim=True).to(d.dtype)
        r = d / q
        ok = ~torch.isinf(r).any(dim=1)
        ape += r[ok].sum(dtype=torch.float64)
        ape_n += ok.sum() * per
    n = float(store.Y.numel())
    return float(ae) / n, (float(se) / n) ** 0.5, (float(ape) / float(ape_n) * 100 if float(ape_n) else float("nan"))


def load_processed_pickle(path: str) -> Dict[str, torch.Tensor]:
    """The reference's ``tpims_data_small.pkl``: a ``torch.save``d 13-tuple (edge_index, edge_attr, 5 x (edge_r_index,
    edge_r_attr), node_data_list) with node_data_list = per-timestep (N, 8) float64 (load_dataset.py:436-437).
    Returns the dict layout of tests/golden/tpims_fixture.npz (node_data as (N, 8, steps) float32)."""
    t = torch.load(path, map_location="cpu", weights_only=False)
    if not (isinstance(t, (tuple, list)) and len(t) == 13):
        raise ValueError("expected the reference's 13-tuple (load_dataset.py:436)")
    out = {"edge_index": t[0].long(), "edge_attr": t[1].float()}
    for i, r in enumerate(REGIONS):
        out[f"edge_{r}_index"] = t[2 + 2 * i].long()
        out[f"edge_{r}_attr"] = t[3 + 2 * i].float()
    out["node_data"] = torch.stack(list(t[12]), dim=1).permute(0, 2, 1).float().contiguous()   # :447
    return out


def main(argv=None):
    ap = argparse.ArgumentParser(description="RegT-GCN evaluation (reference predict.py metrics)")
    ap.add_argument("--fixture", help=".npz in the layout of tests/golden/tpims_fixture.npz")
    ap.add_argument("--pickle", help="the reference's processed tpims_data_small.pkl")
    ap.add_argument("--checkpoint", required=True)
    ap.add_argument("--model", default="RegionalTemporalGCN", choices=["RegionalTemporalGCN", "TemporalGCN"])
    ap.add_argument("--num_timesteps_in", default=6, type=int)
    ap.add_argument("--num_timesteps_out", default=1, type=int)
    ap.add_argument("--tr", "--train_ratio", default=0.2, type=float, dest="tr")
    ap.add_argument("--snap_batch", type=int, default=1, help="test snapshots per forward (block-diagonal graph of B copies; same metrics)")
    a = ap.parse_args(argv)
    if a.pickle:
        d = load_processed_pickle(a.pickle)
    elif a.fixture:
        z = np.load(a.fixture)
        d = {k: torch.from_numpy(z[k]) for k in z.files if z[k].ndim > 0}
    else:
        raise SystemExit("give --fixture or --pickle")
    dev = torch.device("cuda:0")
    n, f = d["node_data"].shape[:2]
    xs, ys = snapshot_windows(d["node_data"], a.num_timesteps_in, a.num_timesteps_out)
    _, (vx, vy) = split([x.to(dev) for x in xs], [y.to(dev) for y in ys], a.tr)
    if a.model == "RegionalTemporalGCN":
        model = rnn.RegionalTemporalGCN(f, n, a.num_timesteps_in, a.num_timesteps_out).to(dev)
        graph = model.prepare_graph(d["edge_index"].to(dev), [d[f"edge_{r}_index"].to(dev) for r in REGIONS],
                                    [d[f"edge_{r}_attr"].to(dev) for r in REGIONS])
    else:
        model = rnn.TemporalGCN(f, a.num_timesteps_in, a.num_timesteps_out).to(dev)
        graph = model.prepare_graph(d["edge_index"].to(dev), d["edge_attr"].to(dev), n)
    model.load_state_dict(torch.load(a.checkpoint, map_location=dev, weights_only=True))
    if a.snap_batch > 1:
        from .train import BatchedGraphs, WindowStore
        if a.model == "RegionalTemporalGCN":
            idx, att = [d[f"edge_{r}_index"].to(dev) for r in REGIONS], [d[f"edge_{r}_attr"].to(dev) for r in REGIONS]
            graphs = BatchedGraphs(lambda b: model.prepare_graph(d["edge_index"].to(dev), idx, att, copies=b))
        else:
            graphs = BatchedGraphs(lambda b: model.prepare_graph(d["edge_index"].to(dev), d["edge_attr"].to(dev), n, copies=b))
        mae, rmse, mape = predict_metrics_batched(model, WindowStore(vx, vy), graphs, a.snap_batch)
    else:
        mae, rmse, mape = predict_metrics(model, vx, vy, graph)
    print("MAE: {:.4f}, RMSE: {:.4f}, MAPE: {:.4f}".format(mae, rmse, mape))


if __name__ == "__main__":
    main()
