"""run.py counterpart: the reference's training / evaluation loop semantics around the HIP hot path.

Kept from the reference (run.py:163-243):
* train(): one forward + ``mean((out - y)**2)`` + ``backward()`` per snapshot, gradients ACCUMULATE over all
  train snapshots, ``optimizer.step()`` + ``zero_grad()`` ONCE per epoch, returns the LAST snapshot's loss;
* test(): ``(sqrt(mean(se)), mean(se))`` over all test snapshots (the reference prints the second as "MAE");
* RMSprop(lr, weight_decay) (run.py:145); ``range(epochs + 1)`` epochs; state_dict saved every 10 epochs
  as ``model_in{T}_out{O}_epoch{k}.pt`` (run.py:242-243).
Changed on purpose: snapshots and the static graph live on the GPU for the whole run (the reference copies
every batch host->device and syncs on ``.cpu()`` every step, run.py:172,180), and the graph is prepared once.

    python -m regtgcn_amd.train --model RegionalTemporalGCN --num_timesteps_in 6 --num_timesteps_out 1 \
        --tr 0.2 --epochs 5 --fixture tests/golden/tpims_fixture.npz
"""
from __future__ import annotations

import argparse
import os
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import functional as F_
from . import nn as rnn
from .data import snapshot_windows
from .dist import HaloPipeline, Shard, allreduce_gradients, allreduce_sum

REGIONS = ("IA", "KS", "KY", "OH", "WI")


def train_epoch(model, xs: Sequence[torch.Tensor], ys: Sequence[torch.Tensor], graph, optimizer, stepper=None) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """One epoch of run.py::train() on device-resident snapshots; returns (last loss, all losses) as device scalars.
    ``stepper`` (functional.FusedTrainStep): the same steps without autograd -- for graphs so small that the host work
    per step is the bound."""
    model.train()
    losses = []
    if stepper is not None:
        for x, y in zip(xs, ys):
            losses.append(stepper(x, y).clone())
        allreduce_gradients(list(model.parameters()))
        optimizer.step()
        stepper.zero_grad()
        return losses[-1][0], [l[0] for l in losses]
    # (the gradients of the epoch's snapshots add up in .grad, run.py:178-194: the model's backward does that addition itself, one
    # multi-tensor add instead of one autograd add per parameter -- functional.set_grad_accumulation_in_backward)
    prev = F_.set_grad_accumulation_in_backward(True)
    try:
        for x, y in zip(xs, ys):
            out, _ = model.forward_prepared(x, graph)
            loss = F_.mse_loss(out, y)               # torch.mean((out - y) ** 2), run.py:180, value and gradient in one kernel
            loss.backward()
            losses.append(loss.detach())
    finally:
        F_.set_grad_accumulation_in_backward(prev)
    allreduce_gradients(list(model.parameters()))
    optimizer.step()
    optimizer.zero_grad()
    return losses[-1], losses


@torch.no_grad()
def evaluate(model, xs, ys, graph) -> Tuple[float, float]:
    """run.py::test(): (rmse, mse)."""
    model.eval()
    se = [(model.forward_prepared(x, graph)[0] - y) ** 2 for x, y in zip(xs, ys)]
    m = torch.cat(se, dim=0).mean()
    return float(m.sqrt()), float(m)


# ---- snapshot batching ---------------------------------------------------------------------------------------------------------------
# run.py:170-192 runs one forward / backward per snapshot and ADDS the gradients of all train snapshots; the test loop (:208-216)
# averages squared errors over all test snapshots.  Both are additive over snapshots, and the graph is the same in every one of
# them, so B snapshots can run as ONE problem on the block-diagonal graph of B copies (graph.replicate_edges): M = B*N*T rows per
# launch instead of N*T.  At TPIMS size (N = 104) a per-snapshot step is ~40 dependent launches of 5-25 us -- latency, not work;
# batching turns the same launches into work.  Gradients = the same sum (another summation order), per-snapshot losses are
# still reported, the optimiser still steps once per epoch.

class WindowStore:
    """All sliding windows of a run, device-resident and stacked: X (S, N, F, T), Y (S, N, O) -- load_dataset.py:451-457 -- so that a
    batch of B consecutive snapshots is a VIEW (B*N, F, T) of X, no per-step copy."""

    def __init__(self, xs: Sequence[torch.Tensor], ys: Sequence[torch.Tensor]):
        self.X = torch.stack(list(xs)).contiguous() if len(xs) else torch.empty(0)
        self.Y = torch.stack(list(ys)).contiguous() if len(ys) else torch.empty(0)

    def __len__(self):
        return self.X.shape[0]

    def batch(self, i: int, b: int):
        x, y = self.X[i:i + b], self.Y[i:i + b]
        return x.reshape(-1, x.shape[2], x.shape[3]), y.reshape(-1, y.shape[2])


class BatchedGraphs:
    """prepare_graph(copies=b) per batch size b in use (the epoch's last batch may be shorter), built on first use."""

    def __init__(self, build):
        self._build, self._by_size = build, {}

    def get(self, b: int):
        if b not in self._by_size:
            self._by_size[b] = self._build(b)
        return self._by_size[b]


def train_epoch_batched(model, store: WindowStore, graphs: BatchedGraphs, optimizer, snap_batch: int) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """train_epoch() with ``snap_batch`` snapshots per forward / backward; returns (last snapshot's loss, all per-snapshot losses)."""
    model.train()
    losses = []
    n_o = store.Y.shape[1] * store.Y.shape[2]                 # entries of ONE snapshot: the divisor of run.py:180's mean
    prev = F_.set_grad_accumulation_in_backward(True)
    try:
        for i in range(0, len(store), snap_batch):
            b = min(snap_batch, len(store) - i)
            x, y = store.batch(i, b)
            out, _ = model.forward_prepared(x, graphs.get(b))
            F_.mse_loss(out, y, n_o).backward()               # = the sum of the b snapshot means: gradients add up as in run.py
            losses.append(((out.detach() - y) ** 2).view(b, -1).mean(dim=1))
    finally:
        F_.set_grad_accumulation_in_backward(prev)
    allreduce_gradients(list(model.parameters()))
    optimizer.step()
    optimizer.zero_grad()
    all_l = torch.cat(losses)
    return all_l[-1], list(all_l.unbind(0))


@torch.no_grad()
def evaluate_batched(model, store: WindowStore, graphs: BatchedGraphs, snap_batch: int) -> Tuple[float, float]:
    """run.py::test() with ``snap_batch`` snapshots per forward: (rmse, mse)."""
    model.eval()
    se = torch.zeros((), dtype=torch.float64, device=store.X.device)
    for i in range(0, len(store), snap_batch):
        b = min(snap_batch, len(store) - i)
        x, y = store.batch(i, b)
        se += ((model.forward_prepared(x, graphs.get(b))[0] - y) ** 2).sum(dtype=torch.float64)
    m = float(se) / float(store.Y.numel())
    return m ** 0.5, m


def train_epoch_sharded(model, xs: Sequence[torch.Tensor], ys: Sequence[torch.Tensor], shard: Shard, pipe: HaloPipeline,
                        optimizer, global_nodes: int, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """train_epoch() for one region shard of a multi-GPU run: ``xs[i]`` (n_local, F, T) / ``ys[i]`` (n_local, O) are this
    rank's rows of snapshot i.  The loss is the mean over the GLOBAL graph (run.py:180), so the per-rank partial sums are
    scaled by 1/(global_nodes*O) before backward and the gradients of all ranks add up to the single-GPU gradient (one
    flat all-reduce before the optimiser step).  Halo rows of snapshot i+1 travel while snapshot i computes.
    Returns (last global loss, all global losses) as device tensors."""
    model.train()
    losses = []
    if len(xs):
        pipe.submit(0, xs[0])
    prev = F_.set_grad_accumulation_in_backward(True)
    try:
        for i, (x, y) in enumerate(zip(xs, ys)):
            buf = pipe.acquire(i % 2)
            if i + 1 < len(xs):
                pipe.submit((i + 1) % 2, xs[i + 1])
            out, _ = model.forward_packed(buf, shard.graph)
            loss = F_.mse_loss(out, y, global_nodes * y.shape[1])
            loss.backward()
            pipe.release(i % 2)
            losses.append(loss.detach())
    finally:
        F_.set_grad_accumulation_in_backward(prev)
    allreduce_gradients(list(model.parameters()), group)
    tot = allreduce_sum(torch.stack(losses), group)
    optimizer.step()
    optimizer.zero_grad()
    return tot[-1], tot


@torch.no_grad()
def evaluate_sharded(model, xs, ys, shard: Shard, pipe: HaloPipeline, global_nodes: int, group=None) -> Tuple[float, float]:
    """run.py::test() over region shards: (rmse, mse) of the global graph."""
    model.eval()
    se = torch.zeros((), dtype=torch.float32, device=xs[0].device)
    pipe.submit(0, xs[0])
    for i, (x, y) in enumerate(zip(xs, ys)):
        buf = pipe.acquire(i % 2)
        if i + 1 < len(xs):
            pipe.submit((i + 1) % 2, xs[i + 1])
        se += ((model.forward_packed(buf, shard.graph)[0] - y) ** 2).sum()
        pipe.release(i % 2)
    m = allreduce_sum(se, group) / float(global_nodes * ys[0].shape[1] * len(xs))
    return float(m.sqrt()), float(m)


def split(xs, ys, ratio: float):
    k = int(ratio * len(xs))          # temporal_signal_split: first int(ratio*n) snapshots train, rest test
    return (xs[:k], ys[:k]), (xs[k:], ys[k:])


# models of run.py:115-136 that are built on the TGCN cell (the hot path and the SURVEY 8(f) baselines); the others
# (SpatialGCN, TemporalGConvLSTM, StackedGRU, STAEformer, STID, STNorm) are out of scope
MODELS = ("RegionalTemporalGCN", "RandomTemporalGCN", "TemporalGCN", "ConvStackedTemporalGCN", "GraphSAGETemporalGCN", "GAT", "GATTemporal")


def build_parser() -> argparse.ArgumentParser:
    """run.py's flag set (run.py:22-45) -- the reference's own launch lines parse unchanged, e.g. scripts/RegionalTemporalGCN.sh:1 --
    plus this package's data-source flags.  Flags run.py parses and never reads (--momentum, --bs, --checkpoint_path: RMSprop is
    built from lr / decay only, run.py:145; snapshots are not batched, run.py:170) are accepted and ignored here too."""
    ap = argparse.ArgumentParser(description="RegT-GCN training loop (reference run.py flags)")
    ap.add_argument("--seed", default=42, type=int)
    ap.add_argument("--epochs", default=30, type=int)
    ap.add_argument("--lr", default=1e-3, type=float)
    ap.add_argument("--decay", default=1e-4, type=float)
    ap.add_argument("--momentum", default=0.9, type=float, help="accepted, unused (as in run.py)")
    ap.add_argument("--bs", "--batch_size", default=32, type=int, dest="bs", help="accepted, unused (as in run.py)")
    ap.add_argument("--tr", "--train_ratio", default=0.8, type=float, dest="tr")
    ap.add_argument("--tf", "--train_feature", default="available", type=str, dest="tf", help="occrate / available: the target column (run.py:31)")
    ap.add_argument("--edge_cut", default=None, type=str, help="dataloading_type 1 only in run.py; accepted, unused")
    ap.add_argument("--dataset_path", default=None, type=str, help="the reference's dataset/ directory (= --dataset_root)")
    ap.add_argument("--checkpoint_path", default="../checkpoints/", type=str, help="accepted, unused (as in run.py)")
    ap.add_argument("--dataloading_type", default=2, type=int, help="2 (TruckParkingDataset2 semantics) is what this loader implements")
    ap.add_argument("--decomp_type", default=None, type=str, help="regional: the five state link files; random: run.py's random decomposition "
                                                                  "(load_dataset.py:324-329) of the full graph into five overlapping parts")
    ap.add_argument("--num_timesteps_in", default=8, type=int)
    ap.add_argument("--num_timesteps_out", default=4, type=int)
    ap.add_argument("--model", default="TemporalGCN", choices=list(MODELS))
    ap.add_argument("--is_preprocessed", action="store_true", help="accepted: the data source decides (a .pkl / .npz is preprocessed by definition)")
    ap.add_argument("--is_pretrained", action="store_true")
    ap.add_argument("--pretrained_model", default="")
    ap.add_argument("--pretrained_model_epoch", default="0")
    ap.add_argument("--logs", action="store_true", help="also write the epoch lines to ./logs/<date>.txt (run.py:48-49)")
    # this package's own flags
    ap.add_argument("--fixture", help=".npz with node_data (N,F,steps), edge_index, edge_attr, edge_<R>_index/attr")
    ap.add_argument("--dataset_root", help="the reference's dataset/ directory (read through regtgcn_amd.etl)")
    ap.add_argument("--max_steps", type=int, default=None, help="with --dataset_root: use the first MAX_STEPS timesteps")
    ap.add_argument("--out_dir", default="pretrained")
    ap.add_argument("--fused_step", action="store_true", help="train through functional.FusedTrainStep (no autograd; faster on small graphs)")
    ap.add_argument("--snap_batch", type=int, default=1,
                    help="snapshots per forward / backward (block-diagonal graph of B copies; same accumulated gradients and metrics, "
                         "run.py:170-192 / 208-216 are additive over snapshots).  1 = one launch sequence per snapshot, as run.py")
    return ap


def random_decomposition(edge_index: torch.Tensor, edge_attr: torch.Tensor, parts: int = 5, seed: int = 42):
    """run.py --decomp_type random.  The reference reads five files links/0322/link{1..5}_data.csv there (load_dataset.py:324-329)
    that its repository does not ship; what they hold is a split of the full graph's edges into five parts that ignores the
    states.  Here the five parts are drawn (seeded) from the full edge list, so a node may receive edges in several of them --
    the "overlapping" layout of graph.prepare_graph, the general form of the regional embedding."""
    g = torch.Generator().manual_seed(seed)
    e = edge_index.shape[1]
    idx, att = [], []
    for _ in range(parts):
        pick = torch.randperm(e, generator=g)[: max(1, e // parts)]
        idx.append(edge_index[:, pick].contiguous())
        att.append(edge_attr[pick].contiguous())
    return idx, att


def main(argv=None):
    a = build_parser().parse_args(argv)
    torch.manual_seed(a.seed)
    dev = torch.device("cuda:0")
    if a.dataloading_type != 2:
        raise SystemExit("--dataloading_type 2 (TruckParkingDataset2: full graph + five regional graphs) is the loader this package implements")
    root = a.dataset_root or (a.dataset_path if a.dataset_path and os.path.isdir(os.path.join(a.dataset_path, "nodes")) else None)
    if root:
        from . import etl
        d = {k: v.numpy() for k, v in etl.load_tpims(root, a.max_steps, train_feature=a.tf).as_dict().items()}
    elif a.fixture:
        d = np.load(a.fixture)
        if a.tf.lower() != "occrate":
            print("note: a fixture's target column is fixed when it is built (tests/golden/tpims_fixture.npz: OCCRATE); --tf is not applied")
    else:
        raise SystemExit("give --fixture, --dataset_root or a --dataset_path that holds the reference's dataset/ directory")
    node_data = torch.from_numpy(d["node_data"])
    n, f = node_data.shape[:2]
    xs, ys = snapshot_windows(node_data, a.num_timesteps_in, a.num_timesteps_out)
    xs, ys = [x.to(dev) for x in xs], [y.to(dev) for y in ys]
    (tx, ty), (vx, vy) = split(xs, ys, a.tr)
    ei = torch.from_numpy(d["edge_index"]).to(dev)
    ea = torch.from_numpy(d["edge_attr"]).to(dev)
    if a.model in ("RegionalTemporalGCN", "RandomTemporalGCN"):             # run.py:115-116: one class for both decompositions
        model = rnn.RegionalTemporalGCN(f, n, a.num_timesteps_in, a.num_timesteps_out).to(dev)
        if (a.decomp_type or "regional").lower() == "random":
            r_idx, r_att = random_decomposition(ei, ea, len(REGIONS), a.seed)
        else:
            r_idx = [torch.from_numpy(d[f"edge_{r}_index"]).to(dev) for r in REGIONS]
            r_att = [torch.from_numpy(d[f"edge_{r}_attr"]).to(dev) for r in REGIONS]
        graph = model.prepare_graph(ei, r_idx, r_att)
    elif a.model == "TemporalGCN":
        model = rnn.TemporalGCN(f, a.num_timesteps_in, a.num_timesteps_out).to(dev)
        graph = model.prepare_graph(ei, ea, n)
    elif a.model == "ConvStackedTemporalGCN":                               # run.py:125-126
        model = rnn.ConvStackedTemporalGCN(f, a.num_timesteps_in, a.num_timesteps_out).to(dev)
        graph = model.prepare_graph(ei, ea, n)
    elif a.model == "GraphSAGETemporalGCN":                                 # run.py:127-128
        model = rnn.GraphSAGETemporalGCN(f, n, a.num_timesteps_in, a.num_timesteps_out).to(dev)
        graph = model.prepare_graph(ei, n)
    else:                                                                   # 'GAT' in run.py:129-130 (class GATTemporal)
        model = rnn.GATTemporal(f, n, a.num_timesteps_in, a.num_timesteps_out).to(dev)
        graph = model.prepare_graph(ei, n)
    if a.is_pretrained:
        model.load_state_dict(torch.load(a.pretrained_model, map_location=dev, weights_only=True))
    opt = torch.optim.RMSprop(model.parameters(), lr=a.lr, weight_decay=a.decay)
    out_dir = os.path.join(a.out_dir, a.tf, a.model)                         # run.py:138, 243: pretrained/<tf>/<model>/
    os.makedirs(out_dir, exist_ok=True)
    log = None
    if a.logs:
        import datetime
        os.makedirs("logs", exist_ok=True)
        log = open(os.path.join("logs", datetime.datetime.now().strftime("%y-%m-%d_%H-%M") + ".txt"), "a")
    batched = None
    if a.snap_batch > 1:
        if a.fused_step:
            raise SystemExit("--snap_batch and --fused_step are alternatives (both remove per-snapshot host work)")
        # every operator of these models is local to a node's in-neighbours (gcn_norm, ChebConv.__norm__, SAGE mean, GAT softmax), so
        # B disjoint copies of the graph are B independent snapshots
        if a.model in ("TemporalGCN", "ConvStackedTemporalGCN"):
            graphs = BatchedGraphs(lambda b: model.prepare_graph(ei, ea, n, copies=b))
        elif a.model in ("GraphSAGETemporalGCN", "GAT", "GATTemporal"):
            graphs = BatchedGraphs(lambda b: model.prepare_graph(ei, n, copies=b))
        else:
            graphs = BatchedGraphs(lambda b: model.prepare_graph(ei, r_idx, r_att, copies=b))
        batched = (WindowStore(tx, ty), WindowStore(vx, vy), graphs)
    stepper = None
    if a.fused_step:
        if a.model not in ("RegionalTemporalGCN", "RandomTemporalGCN", "TemporalGCN") or getattr(graph, "overlap", False):
            raise SystemExit("--fused_step covers RegionalTemporalGCN (regional decomposition) / TemporalGCN")
        from .functional import FusedTrainStep
        stepper = FusedTrainStep(model, graph, f, a.num_timesteps_in)
    for epoch in range(a.epochs + 1):
        if batched:
            last, _ = train_epoch_batched(model, batched[0], batched[2], opt, a.snap_batch)
            rmse, mse = evaluate_batched(model, batched[1], batched[2], a.snap_batch)
        else:
            last, _ = train_epoch(model, tx, ty, graph, opt, stepper)
            rmse, mse = evaluate(model, vx, vy, graph)
        line = "Train Loss: {:.4f}, Test RMSE: {:.4f}, MAE: {:.4f}".format(float(last), rmse, mse)   # run.py:236 format
        print(line)
        if log:
            log.write(line + "\n")
            log.flush()
        if epoch % 10 == 0:
            torch.save(model.state_dict(), os.path.join(out_dir, "model_in{}_out{}_epoch{}.pt".format(
                a.num_timesteps_in, a.num_timesteps_out, int(a.pretrained_model_epoch) + epoch)))


if __name__ == "__main__":
    main()
