"""Generate golden vectors by running the REFERENCE's own model files.

TEST INFRASTRUCTURE; runs only in the build container (needs ``/root/reference``; that tree
never travels to the GPU box, the ``.npz`` files written here do).

What is imported from the reference: ``models/RegionalTemporalGCN.py``, ``models/TemporalGCN.py``
and ``models/utils.py`` (TGCN) -- unmodified, via ``sys.path``.  Those files import
``torch_geometric.nn`` / ``torch_scatter``, which are un-vendored third-party packages absent
from this image; for the import to resolve, this script registers
stand-in modules whose ``ChebConv`` / ``GCNConv`` are ``nn.Module`` wrappers (PyG parameter
names) around the restated operators in ``oracle/graph_ops.py``.  Consequence, stated in
oracle/__init__.py and DESIGN.md: the goldens pin the *orchestration* (reference code) on top
of operator arithmetic that is itself only pinned by dense known answers.

Outputs (tests/golden/):
  golden_cell.npz            G2  TGCN cell, C=16, forward + parameter grads (params stored)
  golden_regt_*.npz          G3  RegionalTemporalGCN on the TPIMS fixture, seeded params
  golden_regt_ckpt.npz       G3  same with the reference's shipped checkpoint in6/out1
  golden_tgcn_*.npz          G4  TemporalGCN on the TPIMS fixture
  golden_loop.npz            G5  3-snapshot accumulate-then-RMSprop trajectory (run.py semantics)
  golden_convstack_*.npz     G6  ConvStackedTemporalGCN on the TPIMS fixture
  golden_sage_*.npz          G7  GraphSAGETemporalGCN (TGCN cell with SAGEConv gates) on the TPIMS fixture
  golden_gat_*.npz           G8  GATTemporal (TGCN cell with GATConv gates) on the TPIMS fixture
  ref_ckpt_in6_out1_epoch50.pt   data fixture: the checkpoint used by golden_regt_ckpt
"""
from __future__ import annotations

import os
import shutil
import sys
import types

import numpy as np
import torch
import torch.nn as nn

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import graph_ops, model as omodel  # noqa: E402

REGIONS = ("IA", "KS", "KY", "OH", "WI")


# ---- stand-ins for the absent third-party packages ------------------------------------------------

class _Lin(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        nn.init.xavier_uniform_(self.weight)


class GCNConv(nn.Module):
    def __init__(self, in_channels, out_channels, improved=False, cached=False, add_self_loops=True, **kw):
        super().__init__()
        assert not improved and add_self_loops
        self.lin = _Lin(in_channels, out_channels)
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index, edge_weight=None):
        return graph_ops.gcn_conv(x, edge_index, edge_weight, self.lin.weight, self.bias)


class ChebConv(nn.Module):
    def __init__(self, in_channels, out_channels, K, normalization="sym", bias=True, **kw):
        super().__init__()
        assert K == 2 and normalization == "sym"
        self.lins = nn.ModuleList([_Lin(in_channels, out_channels) for _ in range(K)])
        self.bias = nn.Parameter(torch.zeros(out_channels))

    def forward(self, x, edge_index, edge_weight=None, batch=None, lambda_max=None):
        return graph_ops.cheb_conv(x, edge_index, edge_weight, self.lins[0].weight, self.lins[1].weight, self.bias)


class _LinB(nn.Module):
    def __init__(self, i, o):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(o, i))
        self.bias = nn.Parameter(torch.zeros(o))
        nn.init.xavier_uniform_(self.weight)


class SAGEConv(nn.Module):
    """PyG parameter names (lin_l.weight, lin_l.bias, lin_r.weight) around oracle.graph_ops.sage_conv."""

    def __init__(self, in_channels, out_channels, **kw):
        super().__init__()
        assert not kw, kw
        self.lin_l = _LinB(in_channels, out_channels)
        self.lin_r = _Lin(in_channels, out_channels)

    def forward(self, x, edge_index, size=None):
        assert size is None
        return graph_ops.sage_conv(x, edge_index, self.lin_l.weight, self.lin_l.bias, self.lin_r.weight)


class GATConv(nn.Module):
    """PyG (>= 2.5) parameter names (att_src, att_dst, bias, lin.weight) around oracle.graph_ops.gat_conv."""

    def __init__(self, in_channels, out_channels, **kw):
        super().__init__()
        assert not kw, kw
        self.att_src = nn.Parameter(torch.empty(1, 1, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, 1, out_channels))
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.lin = _Lin(in_channels, out_channels)
        nn.init.xavier_uniform_(self.att_src)
        nn.init.xavier_uniform_(self.att_dst)

    def forward(self, x, edge_index, edge_attr=None, size=None):
        assert edge_attr is None and size is None
        return graph_ops.gat_conv(x, edge_index, self.lin.weight, self.att_src, self.att_dst, self.bias)


def install_standins():
    tg = types.ModuleType("torch_geometric")
    tgnn = types.ModuleType("torch_geometric.nn")
    inits = types.ModuleType("torch_geometric.nn.inits")
    tgnn.ChebConv, tgnn.GCNConv, tgnn.SAGEConv, tgnn.GATConv = ChebConv, GCNConv, SAGEConv, GATConv
    inits.glorot = lambda t: None
    tgnn.inits = inits
    tg.nn = tgnn
    ts = types.ModuleType("torch_scatter")
    ts.scatter_mean = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("unused on the hot path"))
    sys.modules.update({"torch_geometric": tg, "torch_geometric.nn": tgnn,
                        "torch_geometric.nn.inits": inits, "torch_scatter": ts})


def load_reference():
    install_standins()
    sys.path.insert(0, REF)
    from models.RegionalTemporalGCN import RegionalTemporalGCN  # noqa
    from models.TemporalGCN import TemporalGCN  # noqa
    from models.utils import TGCN  # noqa
    return RegionalTemporalGCN, TemporalGCN, TGCN


def load_reference_convstack():
    install_standins()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.ConvStackedTemporalGCN import ConvStackedTemporalGCN  # noqa
    return ConvStackedTemporalGCN


def load_reference_zero_hidden():
    install_standins()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.GraphSAGETemporalGCN import GraphSAGETemporalGCN  # noqa
    from models.GATTemporal import GATTemporal  # noqa
    return GraphSAGETemporalGCN, GATTemporal


# ---- helpers -------------------------------------------------------------------------------------

def fixture():
    d = np.load(os.path.join(OUT, "tpims_fixture.npz"))
    t = {k: torch.from_numpy(d[k]) for k in d.files if d[k].ndim > 0}
    return t


def grads_summary(named_grads, full_limit=8192):
    """Full gradient for small tensors; for big ones row/col sums + strided sample + norm."""
    out = {}
    for name, g in named_grads.items():
        key = name.replace(".", "__")
        if g is None:
            out[f"gnone__{key}"] = np.zeros(0, dtype=np.float32)
            continue
        g = g.detach().to(torch.float32)
        if g.numel() <= full_limit:
            out[f"g__{key}"] = g.numpy()
        else:
            out[f"grow__{key}"] = g.sum(dim=1).numpy()
            out[f"gcol__{key}"] = g.sum(dim=0).numpy()
            out[f"gsmp__{key}"] = g.flatten()[::97].clone().numpy()
            out[f"gnrm__{key}"] = np.array([float(g.norm())], dtype=np.float32)
    return out


def run_module(mod, args, y):
    mod.zero_grad()
    pred, hidden = mod(*args)
    loss = torch.mean((pred - y) ** 2)
    loss.backward()
    return pred.detach(), hidden.detach(), float(loss), {n: p.grad for n, p in mod.named_parameters()}


def regt_args(fx, x):
    return [x, fx["edge_index"]] + [fx[f"edge_{r}_index"] for r in REGIONS] + [fx[f"edge_{r}_attr"] for r in REGIONS]


def param_checksum(sd):
    return np.array([float(sum(v.double().abs().sum() for v in sd.values()))])


# ---- the five golden families ---------------------------------------------------------------------

def golden_cell(TGCN):
    torch.manual_seed(7)
    n, f, c = 10, 8, 16
    cell = TGCN(in_channels=f, out_channels=c)
    with torch.no_grad():
        for name, p in cell.named_parameters():
            p.copy_(torch.randn_like(p) * 0.3)
    g = torch.Generator().manual_seed(11)
    src = torch.randint(0, n, (30,), generator=g)
    dst = torch.randint(0, n, (30,), generator=g)
    src[3], dst[3] = 4, 4            # a pre-existing self loop
    src[5], dst[5] = src[6], dst[6]  # a duplicate edge
    dst[dst == 9] = 0                # node 9 has no in-edge
    src[src == 8] = 1                # node 8 has no out-edge
    ei = torch.stack([src, dst])
    ew = torch.rand(30, generator=g) * 5 + 0.5
    x = torch.rand(n, f, generator=g)
    h = torch.randn(n, c, generator=g)
    out = {"edge_index": ei.numpy(), "edge_weight": ew.numpy(), "x": x.numpy(), "h": h.numpy()}
    for k, v in cell.state_dict().items():
        out["p__" + k.replace(".", "__")] = v.numpy()
    for tag, w in (("unit", None), ("weighted", ew)):
        cell.zero_grad()
        o = cell(x, ei, w, h)
        (o ** 2).sum().backward()
        out[f"out_{tag}"] = o.detach().numpy()
        for name, p in cell.named_parameters():
            out[f"g_{tag}__" + name.replace(".", "__")] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "golden_cell.npz"), **out)


def golden_regt(RegT, fx, t_in, t_out, seed, tag, ckpt=None, window=0):
    n = fx["node_data"].shape[0]
    x = fx["node_data"][:, :, window:window + t_in].contiguous()
    y = fx["node_data"][:, -1, window + t_in:window + t_in + t_out].contiguous()
    mod = RegT(node_features=8, num_nodes=n, periods=t_in, output_dim=t_out)
    if ckpt is None:
        sd = omodel.init_params("RegionalTemporalGCN", 8, t_in, t_out, num_nodes=n, seed=seed)
    else:
        sd = torch.load(ckpt, map_location="cpu", weights_only=True)
    missing = mod.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    pred, hidden, loss, grads = run_module(mod, regt_args(fx, x), y)
    out = {"t_in": t_in, "t_out": t_out, "seed": seed, "window": window, "pred": pred.numpy(),
           "hidden": hidden.numpy(), "loss": np.array([loss]), "param_checksum": param_checksum(sd)}
    out.update(grads_summary(grads))
    np.savez_compressed(os.path.join(OUT, f"golden_regt_{tag}.npz"), **out)
    return pred, hidden, loss


def golden_tgcn(TG, fx, t_in, t_out, seed, tag, window=0):
    x = fx["node_data"][:, :, window:window + t_in].contiguous()
    y = fx["node_data"][:, -1, window + t_in:window + t_in + t_out].contiguous()
    mod = TG(node_features=8, periods=t_in, output_dim=t_out)
    sd = omodel.init_params("TemporalGCN", 8, t_in, t_out, seed=seed)
    mod.load_state_dict(sd, strict=True)
    mod.zero_grad()
    pred, hidden = mod(x=x, edge_index=fx["edge_index"], edge_attr=fx["edge_attr"])  # keyword call, run.py:188
    loss = torch.mean((pred - y) ** 2)
    loss.backward()
    grads = {n: p.grad for n, p in mod.named_parameters()}
    out = {"t_in": t_in, "t_out": t_out, "seed": seed, "window": window, "pred": pred.detach().numpy(),
           "hidden": hidden.detach().numpy(), "loss": np.array([float(loss)]), "param_checksum": param_checksum(sd)}
    out.update(grads_summary(grads))
    np.savez_compressed(os.path.join(OUT, f"golden_tgcn_{tag}.npz"), **out)


def golden_convstack(CS, fx, t_in, t_out, seed, tag, window=0):
    """models/ConvStackedTemporalGCN.py on the TPIMS fixture, positional call of run.py:214."""
    x = fx["node_data"][:, :, window:window + t_in].contiguous()
    y = fx["node_data"][:, -1, window + t_in:window + t_in + t_out].contiguous()
    mod = CS(node_features=8, periods=t_in, output_dim=t_out)
    sd = omodel.init_params("ConvStackedTemporalGCN", 8, t_in, t_out, seed=seed)
    res = mod.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    mod.zero_grad()
    pred, hidden = mod(x, fx["edge_index"], fx["edge_attr"])
    loss = torch.mean((pred - y) ** 2)
    loss.backward()
    grads = {n: p.grad for n, p in mod.named_parameters()}
    out = {"t_in": t_in, "t_out": t_out, "seed": seed, "window": window, "pred": pred.detach().numpy(),
           "hidden": hidden.detach().numpy(), "loss": np.array([float(loss)]), "param_checksum": param_checksum(sd)}
    out.update(grads_summary(grads))
    np.savez_compressed(os.path.join(OUT, f"golden_convstack_{tag}.npz"), **out)
    return float(loss)


def golden_zero_hidden(Mod, name, fx, t_in, t_out, seed, tag, window=0):
    """models/GraphSAGETemporalGCN.py / models/GATTemporal.py on the TPIMS fixture, positional call of
    run.py:214 ``model(batch.x, batch.edge_index, batch.edge_attr)``."""
    n = fx["node_data"].shape[0]
    x = fx["node_data"][:, :, window:window + t_in].contiguous()
    y = fx["node_data"][:, -1, window + t_in:window + t_in + t_out].contiguous()
    mod = Mod(node_features=8, num_nodes=n, periods=t_in, output_dim=t_out)
    sd = omodel.init_params(name, 8, t_in, t_out, num_nodes=n, seed=seed)
    res = mod.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert list(sd.keys()) == list(mod.state_dict().keys()), (list(sd.keys()), list(mod.state_dict().keys()))
    mod.zero_grad()
    pred, hidden = mod(x, fx["edge_index"], fx["edge_attr"])
    loss = torch.mean((pred - y) ** 2)
    loss.backward()
    grads = {n_: p.grad for n_, p in mod.named_parameters()}
    out = {"t_in": t_in, "t_out": t_out, "seed": seed, "window": window, "pred": pred.detach().numpy(),
           "hidden": hidden.detach().numpy(), "loss": np.array([float(loss)]), "param_checksum": param_checksum(sd)}
    out.update(grads_summary(grads))
    short = "sage" if name == "GraphSAGETemporalGCN" else "gat"
    np.savez_compressed(os.path.join(OUT, f"golden_{short}_{tag}.npz"), **out)
    return float(loss)


def golden_loop(RegT, fx, t_in=6, t_out=1, seed=5, n_train=3, n_test=2, epochs=2):
    """run.py:163-226 semantics driven on the reference module: accumulate, one RMSprop step/epoch."""
    n = fx["node_data"].shape[0]
    mod = RegT(node_features=8, num_nodes=n, periods=t_in, output_dim=t_out)
    mod.load_state_dict(omodel.init_params("RegionalTemporalGCN", 8, t_in, t_out, num_nodes=n, seed=seed))
    opt = torch.optim.RMSprop(mod.parameters(), lr=1e-3, weight_decay=1e-4)  # run.py:145
    nd = fx["node_data"]
    wins = [(nd[:, :, i:i + t_in].contiguous(), nd[:, -1, i + t_in:i + t_in + t_out].contiguous())
            for i in range(n_train + n_test)]
    losses, metrics, sums = [], [], []
    for _ in range(epochs):
        mod.train()
        for x, y in wins[:n_train]:
            pred, _ = mod(*regt_args(fx, x))
            loss = torch.mean((pred - y) ** 2)
            loss.backward()
            losses.append(float(loss))
        opt.step()
        opt.zero_grad()
        mod.eval()
        with torch.no_grad():
            se = torch.cat([(mod(*regt_args(fx, x))[0] - y) ** 2 for x, y in wins[n_train:]], dim=0)
        metrics.append([float(se.mean().sqrt()), float(se.mean())])
        sums.append([float(p.detach().double().sum()) for _, p in mod.named_parameters()])
    np.savez_compressed(os.path.join(OUT, "golden_loop.npz"), t_in=t_in, t_out=t_out, seed=seed, n_train=n_train,
                        n_test=n_test, epochs=epochs, losses=np.array(losses), metrics=np.array(metrics),
                        param_sums=np.array(sums), names=np.array([n for n, _ in mod.named_parameters()]))


def main():
    RegT, TG, TGCN = load_reference()
    fx = fixture()
    golden_cell(TGCN)
    for t_in, t_out, seed in ((6, 1, 1), (12, 1, 2), (12, 3, 3), (6, 3, 4)):
        _, _, loss = golden_regt(RegT, fx, t_in, t_out, seed, f"in{t_in}_out{t_out}")
        print("regt", t_in, t_out, "loss", loss)
    ck = os.path.join(REF, "pretrained", "occrate", "RegionalTemporalGCN", "model_in6_out1_epoch50.pt")
    shutil.copyfile(ck, os.path.join(OUT, "ref_ckpt_in6_out1_epoch50.pt"))
    _, _, loss = golden_regt(RegT, fx, 6, 1, -1, "ckpt", ckpt=ck, window=7)
    print("regt ckpt loss", loss)
    for t_in, t_out, seed in ((6, 1, 6), (12, 3, 7)):
        golden_tgcn(TG, fx, t_in, t_out, seed, f"in{t_in}_out{t_out}")
    golden_loop(RegT, fx)
    CS = load_reference_convstack()
    for t_in, t_out, seed in ((6, 1, 8), (12, 3, 9)):
        print("convstack", t_in, t_out, "loss", golden_convstack(CS, fx, t_in, t_out, seed, f"in{t_in}_out{t_out}"))
    SAGE, GATT = load_reference_zero_hidden()
    for t_in, t_out, seed in ((6, 1, 10), (12, 3, 11)):
        print("graphsage", t_in, t_out, "loss", golden_zero_hidden(SAGE, "GraphSAGETemporalGCN", fx, t_in, t_out, seed, f"in{t_in}_out{t_out}"))
        print("gat", t_in, t_out, "loss", golden_zero_hidden(GATT, "GATTemporal", fx, t_in, t_out, seed + 2, f"in{t_in}_out{t_out}"))
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()
