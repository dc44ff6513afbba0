"""CPU restatement of the reference's training / evaluation loop semantics.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Follows run.py:163-199 (train) and
run.py:202-226 (test):

* one forward + ``mean((out - y)**2)`` + ``backward()`` per snapshot, gradients
  ACCUMULATE over all train snapshots;
* ``RMSprop(lr, weight_decay)`` steps ONCE per epoch (run.py:145, 194-195);
* train() returns the LAST snapshot's loss (run.py:197-199);
* test() returns ``(sqrt(mean(se)), mean(se))`` over all test snapshots/nodes/horizon steps
  (run.py:226 -- the second value is an MSE although the reference prints it as "MAE").

Snapshot windows follow load_dataset.py:451-457: ``x_i = node_data[:, :, i:i+T]``,
``y_i = node_data[:, -1, i+T:i+T+O]``; the train/test split is the first ``int(ratio*n)``
snapshots vs. the rest (torch_geometric_temporal ``temporal_signal_split``).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Tuple

import torch


def make_windows(node_data: torch.Tensor, t_in: int, t_out: int):
    """node_data (N, F, n_steps) -> lists of x (N,F,T) and y (N,O) (load_dataset.py:451-457)."""
    n_steps = node_data.shape[2]
    xs, ys = [], []
    for i in range(n_steps - (t_in + t_out) + 1):
        xs.append(node_data[:, :, i:i + t_in])
        ys.append(node_data[:, -1, i + t_in:i + t_in + t_out])
    return xs, ys


def split(xs: List, ys: List, ratio: float):
    k = int(ratio * len(xs))
    return (xs[:k], ys[:k]), (xs[k:], ys[k:])


def train_epoch(params: Dict[str, torch.Tensor], forward: Callable, xs, ys, optimizer) -> Tuple[float, List[float]]:
    """One epoch of run.py::train().  ``forward(params, x) -> (pred, hidden)``; ``params`` are leaf
    tensors with requires_grad.  Returns (last loss, all losses)."""
    losses = []
    for x, y in zip(xs, ys):
        pred, _ = forward(params, x)
        loss = torch.mean((pred - y) ** 2)
        loss.backward()
        losses.append(float(loss.detach()))
    optimizer.step()
    optimizer.zero_grad()
    return losses[-1], losses


@torch.no_grad()
def evaluate(params, forward: Callable, xs, ys) -> Tuple[float, float]:
    """run.py::test(): (rmse, mse) over the whole split."""
    se = [((forward(params, x)[0] - y) ** 2) for x, y in zip(xs, ys)]
    m = torch.cat(se, dim=0).mean()
    return float(m.sqrt()), float(m)


@torch.no_grad()
def predict_metrics(params, forward: Callable, xs, ys) -> Tuple[float, float, float]:
    """predict.py:141-194: (MAE, RMSE, MAPE %) -- absolute error, squared error and absolute error over the
    95th percentile of the snapshot's targets (numpy percentile, snapshot skipped if the ratio has an inf)."""
    import numpy as np
    mae, mse, mape = [], [], []
    for x, y in zip(xs, ys):
        out = forward(params, x)[0]
        mae.append(np.abs((y - out).cpu()))
        mse.append(((y - out) ** 2).cpu())
        ratio = np.abs((y - out).cpu() / np.percentile(y.cpu(), q=95))
        if np.isinf(ratio).any() == 0:
            mape.append(ratio)
    return (float(torch.cat(mae, dim=0).mean()), float(torch.cat(mse, dim=0).mean().sqrt()),
            float(torch.cat(mape, dim=0).mean()) * 100)
