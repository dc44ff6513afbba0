"""TPIMS ETL counterpart: the reference's ``dataset/`` directory -> the tensors the
training loop consumes, and the reference's processed 13-tuple pickle in both directions.

What the reference does (load_dataset.py:309-437, ``TruckParkingDataset2.process``) and what is kept here:

* node index space = the unique ``SITE_ID`` s of ``data/tpims_location.csv`` without the IL / MI / MN / IN sites
  (:341-343); the link files address it by ``SRC_IDX`` / ``DST_IDX`` (:130, :319-323, :349-360);
* one (N, 8) feature matrix per 10-minute timestep with the columns WEEKID, DAYID, HOURID, <distance column>, OWNER,
  AMENITY, CAPACITY, OCCRATE (:413-415), min-max scaled to [0, 1] **per timestep across the nodes** (:429-430); a
  constant column maps to 0 (sklearn's MinMaxScaler);
* the processed file is a ``torch.save`` d 13-tuple (edge_index, edge_attr, 5 x (edge_<R>_index, edge_<R>_attr),
  node_data_list) (:434-436); sliding windows are cut by ``data.snapshot_windows`` (:451-457).

What differs, on purpose: the reference rebuilds every timestep from one large raw table (``tpims_data_<size>.csv``,
not shipped with the repository) with a pandas query per step; this reader consumes the per-timestep files the
repository does ship (``nodes/0322/node_data_*.csv``, column names of load_dataset.py:126) with one pass of the ``csv``
module.  Sites without a row at a timestep get zeros (the reference's fill rule indexes a list past its end and falls
into its ``IndexError`` branch, :406-409, which is zeros too).  The full graph is ``links/0322/link_data.csv`` when
present, else the union of the five regional link files.

CPU-side, runs once per dataset; nothing here touches the GPU.
"""
from __future__ import annotations

import csv
import io
import os
import tarfile
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

REGIONS = ("IA", "KS", "KY", "OH", "WI")
EXCLUDED_STATES = ("IL", "MI", "MN", "IN")          # load_dataset.py:342
# columns of a node_data_*.csv row (load_dataset.py:126) that make up the 8 features, in feature order
FEATURE_COLUMNS = (3, 4, 5, 6, 8, 9, 10, 12)        # WEEKID, DAYID, HOURID, TRAVEL_TIME, OWNER, AMENITY, CAPACITY, OCCRATE
# the last feature (= the prediction target, load_dataset.py:453) is chosen by run.py's --tf / --train_feature
# (load_dataset.py:417-419: ``self.train_feature.upper()``): column 11 = AVAILABLE, column 12 = OCCRATE
TARGET_COLUMN = {"occrate": 12, "available": 11}


def feature_columns(train_feature: str = "occrate") -> Tuple[int, ...]:
    tf = train_feature.lower()
    if tf not in TARGET_COLUMN:
        raise ValueError(f"train_feature must be 'occrate' or 'available' (run.py --tf), got {train_feature!r}")
    return FEATURE_COLUMNS[:-1] + (TARGET_COLUMN[tf],)


@dataclass
class TpimsData:
    site_ids: List[str]
    node_data: torch.Tensor                 # (N, 8, steps) float32, min-max scaled per timestep
    edge_index: torch.Tensor                # (2, E) int64
    edge_attr: torch.Tensor                 # (E,) float32   (DIST)
    region_index: List[torch.Tensor]        # 5 x (2, E_r)
    region_attr: List[torch.Tensor]         # 5 x (E_r,)

    def as_dict(self) -> Dict[str, torch.Tensor]:
        """Layout of tests/golden/tpims_fixture.npz, consumed by train.py / evaluate.py."""
        d = {"node_data": self.node_data, "edge_index": self.edge_index, "edge_attr": self.edge_attr}
        for r, i, a in zip(REGIONS, self.region_index, self.region_attr):
            d[f"edge_{r}_index"], d[f"edge_{r}_attr"] = i, a
        return d


def read_sites(location_csv: str) -> List[str]:
    """Unique SITE_IDs in file order, without the excluded states."""
    with open(location_csv, newline="") as f:
        rows = list(csv.DictReader(f))
    seen, out = set(), []
    for r in rows:
        s = r.get("SITE_ID") or ""
        if not s or s.startswith(EXCLUDED_STATES) or s in seen:
            continue
        seen.add(s)
        out.append(s)
    return out


def _parse_links(rows: Iterable[Sequence[str]], num_nodes: int) -> Tuple[torch.Tensor, torch.Tensor]:
    src, dst, dist = [], [], []
    for x in rows:
        if len(x) < 5:
            continue
        try:
            s, d, w = int(x[0]), int(x[2]), float(x[4])
        except ValueError:                   # header line
            continue
        if not (0 <= s < num_nodes and 0 <= d < num_nodes):
            raise ValueError(f"link ({s} -> {d}) outside the {num_nodes}-site index space")
        src.append(s); dst.append(d); dist.append(w)
    return torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1), torch.tensor(dist, dtype=torch.float32)


def read_links(root: str, num_nodes: int, names: Sequence[str] = REGIONS) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    """``link_<R>_data.csv`` (SRC_IDX, SRC, DST_IDX, DST, DIST) from ``links/0322/`` or from ``tpims_link_0322.tar.xz``;
    key "" holds ``link_data.csv`` (the full graph) when it exists."""
    out: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
    want = {f"link_{r}_data.csv": r for r in names}
    want["link_data.csv"] = ""
    link_dir = os.path.join(root, "links", "0322")
    if os.path.isdir(link_dir):
        for fn, key in want.items():
            p = os.path.join(link_dir, fn)
            if os.path.exists(p):
                with open(p, newline="") as f:
                    out[key] = _parse_links(csv.reader(f), num_nodes)
    tar = os.path.join(root, "tpims_link_0322.tar.xz")
    if len(out) < len(names) and os.path.exists(tar):
        with tarfile.open(tar) as tf:
            for m in tf.getmembers():
                key = want.get(os.path.basename(m.name))
                if key is not None and key not in out and m.isfile():
                    out[key] = _parse_links(csv.reader(io.TextIOWrapper(tf.extractfile(m))), num_nodes)
    missing = [r for r in names if r not in out]
    if missing:
        raise FileNotFoundError(f"no link file for region(s) {missing} under {root}")
    return out


def minmax_per_timestep(raw: np.ndarray) -> np.ndarray:
    """(N, F) -> [0, 1] per column (sklearn MinMaxScaler(feature_range=(0, 1)).fit_transform, load_dataset.py:430)."""
    lo, hi = raw.min(axis=0), raw.max(axis=0)
    return (raw - lo) / np.where(hi > lo, hi - lo, 1.0)


def read_node_steps(node_dir: str, site_ids: Sequence[str], max_steps: Optional[int] = None, start: int = 0,
                    train_feature: str = "occrate") -> torch.Tensor:
    """All ``node_data_*.csv`` of ``node_dir`` in name (= time) order -> (N, 8, steps) float32; the eighth feature is OCCRATE or
    AVAILABLE (``train_feature``)."""
    FEATURE_COLUMNS = feature_columns(train_feature)
    idx = {s: i for i, s in enumerate(site_ids)}
    files = sorted(fn for fn in os.listdir(node_dir) if fn.startswith("node_data_") and fn.endswith(".csv"))
    files = files[start:None if max_steps is None else start + max_steps]
    if not files:
        raise FileNotFoundError(f"no node_data_*.csv under {node_dir}")
    n = len(site_ids)
    data = np.zeros((n, len(FEATURE_COLUMNS), len(files)), dtype=np.float64)
    for t, fn in enumerate(files):
        raw = np.zeros((n, len(FEATURE_COLUMNS)), dtype=np.float64)
        with open(os.path.join(node_dir, fn), newline="") as f:
            for row in csv.reader(f):
                i = idx.get(row[1]) if len(row) > max(FEATURE_COLUMNS) else None
                if i is not None:
                    raw[i] = [float(row[c]) for c in FEATURE_COLUMNS]
        data[:, :, t] = minmax_per_timestep(np.nan_to_num(raw))
    return torch.from_numpy(data.astype(np.float32))


def load_tpims(root: str, max_steps: Optional[int] = None, start: int = 0, train_feature: str = "occrate") -> TpimsData:
    """``root`` = the reference's ``dataset/`` directory; ``train_feature`` = run.py's --tf (occrate / available)."""
    sites = read_sites(os.path.join(root, "data", "tpims_location.csv"))
    links = read_links(root, len(sites))
    node_data = read_node_steps(os.path.join(root, "nodes", "0322"), sites, max_steps, start, train_feature)
    r_idx = [links[r][0] for r in REGIONS]
    r_att = [links[r][1] for r in REGIONS]
    if "" in links:
        edge_index, edge_attr = links[""]
    else:
        edge_index, edge_attr = torch.cat(r_idx, dim=1), torch.cat(r_att)
    return TpimsData(list(sites), node_data, edge_index, edge_attr, r_idx, r_att)


def save_processed_tuple(d: TpimsData, path: str) -> None:
    """Write the reference's ``tpims_data_<size>.pkl`` layout (load_dataset.py:434-436): what its own
    ``TruckParkingDataset2(preprocessed=True).get()`` loads."""
    steps = [d.node_data[:, :, t].double().contiguous() for t in range(d.node_data.shape[2])]
    flat: List[torch.Tensor] = [d.edge_index, d.edge_attr]
    for i, a in zip(d.region_index, d.region_attr):
        flat += [i, a]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    torch.save(tuple(flat) + (steps,), path)


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="TPIMS dataset directory -> processed 13-tuple pickle / .npz")
    ap.add_argument("--root", required=True, help="the reference's dataset/ directory")
    ap.add_argument("--max_steps", type=int, default=None)
    ap.add_argument("--tf", "--train_feature", default="occrate", dest="tf", choices=sorted(TARGET_COLUMN), help="target column (run.py --tf)")
    ap.add_argument("--out", required=True, help="*.pkl (reference layout) or *.npz (fixture layout)")
    a = ap.parse_args(argv)
    d = load_tpims(a.root, a.max_steps, train_feature=a.tf)
    if a.out.endswith(".npz"):
        np.savez_compressed(a.out, **{k: v.numpy() for k, v in d.as_dict().items()})
    else:
        save_processed_tuple(d, a.out)
    print(f"{len(d.site_ids)} sites, {d.node_data.shape[2]} timesteps, {d.edge_index.shape[1]} edges -> {a.out}")


if __name__ == "__main__":
    main()
