"""The A/B switches of DESIGN.md section 6b select other kernels for the same arithmetic: each must reproduce the default
path's forward + backward on one seeded problem (tools/ab_libs.py runs every configuration in its own process -- the
switches are read once per process)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "regt-gcn_amd", "lib")


def _ab(spec_b, mode, nodes, feat):
    env = dict(os.environ, AB_F=str(feat))
    env.pop("REGT_LIB_DIR", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_libs.py"), LIB, LIB + ":" + spec_b, str(mode), str(nodes)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = re.findall(r"^(\S+)\s+max\|a-b\| (\S+)\s+max\|a\| (\S+)$", out.stdout, flags=re.M)
    assert len(rows) > 10, out.stdout[-2000:]
    return [(name, float(d), float(a)) for name, d, a in rows]


@pytest.mark.parametrize("switch,mode,rel", [
    # same kernels, descriptors from the LDS table: bit-identical -- except that the generated-operand candidate data gradient
    # (round 4) has no table form, so this switch also restores cell_bwd + dgrad_candidate: same dhp / dzp / drp / dh to the bit,
    # the attention gradient summed in another fixed order (1e-5 of its scale)
    ("REGT_GEMM_DESC=table", 0, 0.0),
    # bf16: also turns the fragment-order weights off, and with them the bf16 rows of x / A_hat x / L~ x and the fused kernels
    # (x is then aggregated unrounded and rounded at LDS staging instead of once while it is packed): same arithmetic, one
    # rounding point moved -- held to the bf16 bar of tests/test_gpu_bf16.py (8 u)
    ("REGT_GEMM_DESC=table", 2, 8 * 2.0 ** -9),
    ("REGT_FP32_CORE=wide", 0, 1e-5),        # two-workgroup kernels: a node's two partial sums are added in another order
    # (round 5: every environment switch the library still reads has a line here -- REGT_GEMM_MODE is the `mode` column,
    # REGT_FUSED_TRACE a developer trace, REGT_TGCN_COLLAPSE is held by tests/test_gpu_ops.py on the TemporalGCN model)
    ("REGT_SIDE_STREAM=0", 0, 1e-6),         # every kernel on the launch stream: same kernels, same reductions
    ("REGT_HIPGRAPH=1", 0, 1e-6),            # forward / backward replayed as captured graphs
    ("REGT_SPMM_ROWS=1", 0, 1e-6),           # row-block aggregation kernels: same CSR order
    ("REGT_SPMM_PL=8", 0, 1e-6),             # 128-byte column panels
    ("REGT_DGRAD1_GEN=0", 0, 1e-5),          # cell_bwd + dgrad_candidate as two launches (attention gradient: another fixed order)
    ("REGT_XBF=0", 2, 8 * 2.0 ** -9),        # bf16 without the bf16-row layout / fused forward (x rounded at another point)
    ("REGT_FUSED_BWD=0", 2, 8 * 2.0 ** -9),  # bf16 with the three data-gradient launches
])
def test_switch_reproduces_default(switch, mode, rel):
    for name, diff, scale in _ab(switch, mode, 3000, 32):
        # (the attention gradient is a difference of nearly equal terms: 6 x the bar, as in tests/test_gpu_bf16.py)
        r = 6 * rel if (mode == 2 and name == "g:tgnn._attention") else rel
        if mode == 0 and switch == "REGT_GEMM_DESC=table" and name == "g:tgnn._attention":
            r = 1e-5
        assert diff <= r * scale + (0.0 if r == 0.0 else 1e-9), (switch, mode, name, diff, scale)


def test_every_runtime_option_is_known_and_restores():
    """regt_set_option(name, v) returns the previous setting for every documented name and rejects unknown ones."""
    sys.path.insert(0, ROOT)
    import regtgcn_amd as R
    lib = R.load_library()
    defaults = {b"xbf": 1, b"fused_bwd": 1, b"fused_rows": 1, b"embed_kernel": 1, b"spmm_rows": 0, b"dgrad1_gen": 1, b"wgrad_ring": 6, b"wgrad_tile": 256,
                b"wgrad_ring256": 2, b"wgrad_bnw64": 1, b"wgrad_wave": 1, b"wgrad_pairs": 2, b"tgcn_collapse": 1}
    for name, dflt in defaults.items():
        prev = lib.regt_set_option(name, dflt)
        assert prev == dflt, (name, prev, dflt)                  # (the suite leaves every option at its default)
        assert lib.regt_set_option(name, dflt) == dflt
    assert lib.regt_set_option(b"no_such_option", 1) == -1
