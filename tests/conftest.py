"""pytest configuration: `gpu` marker, repo-root import path, golden-fixture helpers."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REGIONS = ("IA", "KS", "KY", "OH", "WI")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_global_rng():
    """Tests that draw from torch's global generator see the same numbers on every run (tolerances are a few ulp)."""
    torch.manual_seed(1234)
    np.random.seed(1234)


def load_npz(name):
    d = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: d[k] for k in d.files}


@pytest.fixture(scope="session")
def tpims():
    d = load_npz("tpims_fixture.npz")
    return {k: torch.from_numpy(v) for k, v in d.items() if v.ndim > 0}


def region_lists(fx):
    return [fx[f"edge_{r}_index"] for r in REGIONS], [fx[f"edge_{r}_attr"] for r in REGIONS]


def check_grads_against_golden(golden, grads, atol, rtol=1e-4):
    """`grads`: dict name -> tensor|None; golden holds full / summarised grads (oracle/make_goldens.py)."""
    checked = 0
    for key, ref in golden.items():
        if "__" not in key or not key.startswith(("g__", "grow__", "gcol__", "gsmp__", "gnrm__", "gnone__")):
            continue
        kind, rest = key.split("__", 1)
        name = rest.replace("__", ".")
        g = grads[name]
        if kind == "gnone":
            assert g is None or float(g.abs().max()) == 0.0, name
            continue
        g = g.detach().cpu().to(torch.float32)
        if kind == "g":
            got = g.numpy()
        elif kind == "grow":
            got = g.sum(dim=1).numpy()
        elif kind == "gcol":
            got = g.sum(dim=0).numpy()
        elif kind == "gsmp":
            got = g.flatten()[::97].numpy()
        else:
            got = np.array([float(g.norm())], dtype=np.float32)
        np.testing.assert_allclose(got, ref, atol=atol, rtol=rtol, err_msg=f"{kind} {name}")
        checked += 1
    assert checked > 10
