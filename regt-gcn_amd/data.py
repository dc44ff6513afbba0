"""Synthetic regional graphs of the BASELINE.json shapes and snapshot windows.

``synthetic_regional_graph`` follows SURVEY.md section 8(d): contiguous region blocks, source uniform
over all nodes, destination uniform inside the source's region with probability ``p_intra`` (else
uniform over all nodes -> cross-region edges that only the full graph holds), no self loops, no
duplicates, exactly E directed edges, ``edge_attr ~ U(75, 3000)`` (the TPIMS DIST range), regional
graphs = intra-region subset with the same weights (global node ids, like the reference's link
files).  ``snapshot_windows`` restates load_dataset.py:451-457.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import numpy as np
import torch


@dataclass
class RegionalGraph:
    num_nodes: int
    num_regions: int
    edge_index: torch.Tensor            # (2,E) int64
    edge_attr: torch.Tensor             # (E,) float32
    region_index: List[torch.Tensor]    # R x (2,E_r) int64, global ids
    region_attr: List[torch.Tensor]     # R x (E_r,) float32
    region_bounds: np.ndarray           # (R+1,) node id boundaries

    def to(self, device):
        return RegionalGraph(self.num_nodes, self.num_regions, self.edge_index.to(device), self.edge_attr.to(device),
                             [t.to(device) for t in self.region_index], [t.to(device) for t in self.region_attr],
                             self.region_bounds)


def synthetic_regional_graph(num_nodes: int, num_edges: int, num_regions: int, seed: int = 42,
                             p_intra: float = 0.95) -> RegionalGraph:
    rng = np.random.default_rng(seed)
    bounds = np.linspace(0, num_nodes, num_regions + 1).astype(np.int64)
    region_of = np.searchsorted(bounds, np.arange(num_nodes), side="right") - 1
    keys = np.zeros(0, dtype=np.int64)
    while keys.size < num_edges:
        m = int((num_edges - keys.size) * 1.1) + 1024
        src = rng.integers(0, num_nodes, size=m, dtype=np.int64)
        r = region_of[src]
        lo, hi = bounds[r], bounds[r + 1]
        dst_in = lo + (rng.random(m) * (hi - lo)).astype(np.int64)
        dst_any = rng.integers(0, num_nodes, size=m, dtype=np.int64)
        dst = np.where(rng.random(m) < p_intra, dst_in, dst_any)
        ok = src != dst
        new = src[ok] * num_nodes + dst[ok]
        keys = np.unique(np.concatenate([keys, new]))
    keys = rng.permutation(keys)[:num_edges]
    src, dst = keys // num_nodes, keys % num_nodes
    attr = rng.uniform(75.0, 3000.0, size=num_edges).astype(np.float32)
    ei = torch.from_numpy(np.stack([src, dst]))
    ea = torch.from_numpy(attr)
    rs, rd = region_of[src], region_of[dst]
    r_idx, r_attr = [], []
    for r in range(num_regions):
        m = (rs == r) & (rd == r)
        r_idx.append(torch.from_numpy(np.stack([src[m], dst[m]])))
        r_attr.append(torch.from_numpy(attr[m]))
    return RegionalGraph(num_nodes, num_regions, ei, ea, r_idx, r_attr, bounds)


def synthetic_snapshots(num_nodes: int, num_features: int, periods: int, horizon: int, count: int, seed: int = 42):
    """``count`` independent (x (N,F,T), y (N,O)) pairs, x,y ~ U(0,1) like the min-max-scaled TPIMS features."""
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(num_nodes, num_features, periods, generator=g), torch.rand(num_nodes, horizon, generator=g))
            for _ in range(count)]


def snapshot_windows(node_data: torch.Tensor, t_in: int, t_out: int) -> Tuple[list, list]:
    """node_data (N,F,steps) -> x_i = node_data[:, :, i:i+T], y_i = node_data[:, -1, i+T:i+T+O]."""
    steps = node_data.shape[2]
    xs, ys = [], []
    for i in range(steps - (t_in + t_out) + 1):
        xs.append(node_data[:, :, i:i + t_in].contiguous())
        ys.append(node_data[:, -1, i + t_in:i + t_in + t_out].contiguous())
    return xs, ys
