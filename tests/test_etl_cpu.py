"""TPIMS ETL counterpart (regt-gcn_amd/etl.py): a synthetic dataset directory in the reference's layout, the processed
13-tuple round trip, and -- when the reference tree is mounted -- agreement with the committed fixture, which an
independent script (oracle/make_fixtures.py) built from the same files."""
import os

import numpy as np
import pytest
import torch

import regtgcn_amd as R
from conftest import GOLDEN

REF_DATASET = "/root/reference/dataset"


def _write_dataset(root):
    os.makedirs(root / "data")
    os.makedirs(root / "links" / "0322")
    os.makedirs(root / "nodes" / "0322")
    sites = ["IA001", "KS001", "IL999", "KY001", "OH001", "WI001", "IA002"]        # the IL site must be dropped
    with open(root / "data" / "tpims_location.csv", "w") as f:
        f.write("SITE_ID,NAME\n")
        for s in sites + ["IA001"]:                                               # duplicate row
            f.write(f"{s},x\n")
    kept = [s for s in sites if not s.startswith("IL")]
    pairs = {"IA": [(0, 5, 100.0), (5, 0, 120.0)], "KS": [(1, 1, 5.0)], "KY": [(2, 3, 7.5)], "OH": [(3, 2, 9.0)], "WI": [(4, 0, 11.0)]}
    for r, rows in pairs.items():
        with open(root / "links" / "0322" / f"link_{r}_data.csv", "w") as f:
            for s, d, w in rows:
                f.write(f"{s},{kept[s]},{d},{kept[d]},{w}\n")
    rng = np.random.default_rng(0)
    raws = []
    for t in range(4):
        raw = np.zeros((len(kept), 13))
        with open(root / "nodes" / "0322" / f"node_data_2022-03-01T00-{t}0-00Z.csv", "w") as f:
            for i, s in enumerate(kept):
                if t == 2 and i == 3:
                    continue                                                       # a site without a row at this step
                v = rng.uniform(1, 50, size=13)
                raw[i] = v
                f.write(",".join([str(i + 1), s, "2022-03-01T00:00:00Z"] + [repr(float(x)) for x in v[3:]]) + "\n")
            f.write("99,IL999,2022-03-01T00:00:00Z," + ",".join(["1"] * 10) + "\n")    # row of an excluded site
        raws.append(raw)
    return kept, pairs, raws


def test_etl_reads_reference_layout(tmp_path):
    kept, pairs, raws = _write_dataset(tmp_path)
    d = R.etl.load_tpims(str(tmp_path))
    assert d.site_ids == kept
    assert d.node_data.shape == (6, 8, 4) and d.node_data.dtype == torch.float32
    for t, raw in enumerate(raws):
        cols = raw[:, list(R.etl.FEATURE_COLUMNS)]
        want = (cols - cols.min(0)) / np.where(cols.max(0) > cols.min(0), cols.max(0) - cols.min(0), 1.0)
        np.testing.assert_allclose(d.node_data[:, :, t].numpy(), want, atol=1e-6)
    assert float(d.node_data.min()) == 0.0 and float(d.node_data.max()) == 1.0
    assert d.edge_index.shape[1] == sum(len(v) for v in pairs.values())            # union of the regional files
    for r, i, a in zip(R.etl.REGIONS, d.region_index, d.region_attr):
        assert i.t().tolist() == [[s, t] for s, t, _ in pairs[r]]
        assert a.tolist() == pytest.approx([w for _, _, w in pairs[r]])
    with pytest.raises(ValueError):
        R.etl._parse_links([["0", "a", "77", "b", "1.0"]], 6)


def test_etl_available_target_column(tmp_path):
    """run.py --tf available (load_dataset.py:417-419): the eighth feature -- the prediction target -- is the AVAILABLE column."""
    kept, pairs, raws = _write_dataset(tmp_path)
    occ = R.etl.load_tpims(str(tmp_path), train_feature="occrate")
    av = R.etl.load_tpims(str(tmp_path), train_feature="available")
    assert torch.equal(occ.node_data[:, :7], av.node_data[:, :7])
    for t, raw in enumerate(raws):
        col = raw[:, 11]
        want = (col - col.min()) / (col.max() - col.min())
        np.testing.assert_allclose(av.node_data[:, 7, t].numpy(), want, atol=1e-6)
    assert not torch.equal(occ.node_data[:, 7], av.node_data[:, 7])


def test_processed_tuple_round_trip(tmp_path):
    _write_dataset(tmp_path)
    d = R.etl.load_tpims(str(tmp_path), max_steps=3)
    p = tmp_path / "processed" / "tpims_data_small.pkl"
    R.etl.save_processed_tuple(d, str(p))
    t = torch.load(str(p), weights_only=False)
    assert isinstance(t, tuple) and len(t) == 13 and len(t[12]) == 3 and t[12][0].dtype == torch.float64
    back = R.evaluate.load_processed_pickle(str(p))
    for k, v in d.as_dict().items():
        assert torch.equal(back[k].to(v.dtype), v), k
    xs, ys = R.data.snapshot_windows(back["node_data"], 2, 1)
    assert len(xs) == 1 and xs[0].shape == (6, 8, 2) and ys[0].shape == (6, 1)


@pytest.mark.skipif(not os.path.isdir(REF_DATASET), reason="reference dataset directory not mounted")
def test_etl_agrees_with_committed_fixture():
    fx = np.load(os.path.join(GOLDEN, "tpims_fixture.npz"))
    steps = fx["node_data"].shape[2]
    d = R.etl.load_tpims(REF_DATASET, max_steps=steps)
    assert d.node_data.shape == fx["node_data"].shape
    np.testing.assert_allclose(d.node_data.numpy(), fx["node_data"], atol=1e-7)
    n_union = int(fx["n_union_edges"])                                              # the fixture appends synthetic cross edges
    assert np.array_equal(d.edge_index.numpy(), fx["edge_index"][:, :n_union])
    for r, i, a in zip(R.etl.REGIONS, d.region_index, d.region_attr):
        assert np.array_equal(i.numpy(), fx[f"edge_{r}_index"]) and np.allclose(a.numpy(), fx[f"edge_{r}_attr"])
