"""Two region-shard processes on the one GPU of the test box (gloo carries the collectives, staged through the host;
the compute is the HIP path): HaloPipeline + all-to-all halo exchange + gradient all-reduce reproduce the parameters
and losses of single-GPU training on the global graph.  With RCCL peers the same code runs with backend 'nccl'."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import model as M

pytestmark = pytest.mark.gpu

WORLD, N_PER, REG_PER, T, O, SNAPS, EPOCHS = 2, 1200, 2, 6, 2, 3, 2
# (F, GEMM arithmetic): the fp32 path of configs[3] and the bf16 path of configs[4] (bf16 rows packed and exchanged as bf16,
# regt_forward_packed_bf16, fused forward / backward kernels)
CASES = {"fp32": (8, 0), "bf16": (64, 2)}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _optimizer(model):
    # plain SGD: RMSprop's first step is lr * g / (0.1 |g|), which turns rounding-level differences of near-zero gradient
    # elements into +-lr -- fine for training, useless for an equality test of two summation orders
    return torch.optim.SGD(model.parameters(), lr=1e-2, weight_decay=1e-4)


def _problem(R, F):
    n = N_PER * WORLD
    g = R.data.synthetic_regional_graph(n, 9000 * WORLD, REG_PER * WORLD, seed=11, p_intra=0.8)
    snaps = R.data.synthetic_snapshots(n, F, T, O, SNAPS, seed=11)
    p = M.init_params("RegionalTemporalGCN", F, T, O, num_nodes=n, num_regions=REG_PER * WORLD, seed=12)
    return n, g, snaps, p


def _worker(rank, port, q, case):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    try:
        import regtgcn_amd as R
        torch.cuda.set_device(0)
        F, mode = CASES[case]
        R.load_library().regt_set_gemm_mode(mode)
        n, g, snaps, p = _problem(R, F)
        model = R.RegionalTemporalGCN(F, n, T, O, num_regions=REG_PER * WORLD)   # num_nodes only sizes unused parameters
        model.load_state_dict(p)
        model = model.cuda()
        bounds = np.arange(WORLD + 1, dtype=np.int64) * N_PER
        owner = [r // REG_PER for r in range(REG_PER * WORLD)]
        sh = R.dist.build_shard(g.edge_index, g.region_index, g.region_attr, n, bounds, owner, rank, WORLD, "cuda")      # own rows only
        # ... which is, bit for bit, the slice of the globally normalised operator (the form every rank used to build by itself)
        ref = R.dist.build_shard(g.edge_index, g.region_index, g.region_attr, n, bounds, owner, rank, WORLD, "cuda", method="global")
        for name in ("rowptr", "col", "val", "m_rowptr", "m_col", "m_val_a", "m_val_l", "node_region"):
            assert torch.equal(getattr(sh.graph, name), getattr(ref.graph, name)), f"own-rows shard differs from the global build in {name}"
        assert torch.equal(sh.send_idx, ref.send_idx) and np.array_equal(sh.topo.halo_ids(), ref.topo.halo_ids())
        assert sh.topo.send_splits == ref.topo.send_splits and sh.topo.recv_splits == ref.topo.recv_splits
        pipe = R.dist.HaloPipeline(sh, T, F, torch.device("cuda", 0), dtype=torch.bfloat16 if mode == 2 else torch.float32)
        lo, hi = sh.topo.node_lo, sh.topo.node_hi
        xs = [x[lo:hi].contiguous().cuda() for x, _ in snaps]
        ys = [y[lo:hi].contiguous().cuda() for _, y in snaps]
        opt = _optimizer(model)
        losses = []
        for _ in range(EPOCHS):
            _, tot = R.train.train_epoch_sharded(model, xs, ys, sh, pipe, opt, n)
            losses.append(tot.cpu().numpy())
        rmse, mse = R.train.evaluate_sharded(model, xs, ys, sh, pipe, n)
        sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        q.put((rank, "ok", np.stack(losses), (rmse, mse), sd))
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put((rank, f"FAIL {type(e).__name__}: {e}\n{traceback.format_exc()}", None, None, None))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["fp32", "bf16"])
def test_two_shard_training_matches_single_gpu(case):
    import regtgcn_amd as R
    F, mode = CASES[case]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q, case)) for r in range(WORLD)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), [r[1] for r in res]

    lib = R.load_library()
    prev = lib.regt_set_gemm_mode(mode)
    n, g, snaps, p = _problem(R, F)
    model = R.RegionalTemporalGCN(F, n, T, O, num_regions=REG_PER * WORLD)
    model.load_state_dict(p)
    model = model.cuda()
    graph = model.prepare_graph(g.edge_index.cuda(), [i.cuda() for i in g.region_index], [a.cuda() for a in g.region_attr])
    xs = [x.cuda() for x, _ in snaps]
    ys = [y.cuda() for _, y in snaps]
    opt = _optimizer(model)
    want_losses = []
    for _ in range(EPOCHS):
        _, all_l = R.train.train_epoch(model, xs, ys, graph, opt)
        want_losses.append(torch.stack(all_l).cpu().numpy())
    want_rmse, want_mse = R.train.evaluate(model, xs, ys, graph)
    lib.regt_set_gemm_mode(prev)

    # bf16: the two shards add their gradient parts in another order than the single GPU sums its rows; a parameter that moves by
    # one fp32 ulp can flip the bf16 rounding of a weight -- the comparison is held to the bf16 bar of tests/test_gpu_bf16.py (8 u)
    rt, at = (2e-5, 3e-5) if mode == 0 else (8 * 2.0 ** -9, 8 * 2.0 ** -9)
    np.testing.assert_array_equal(res[0][2], res[1][2])                       # both ranks report the global losses
    np.testing.assert_allclose(res[0][2], np.stack(want_losses), rtol=rt, atol=1e-7)
    np.testing.assert_allclose(res[0][3], (want_rmse, want_mse), rtol=rt)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        np.testing.assert_array_equal(res[0][4][k], res[1][4][k], err_msg=f"ranks diverged on {k}")
        np.testing.assert_allclose(res[0][4][k], v, atol=at * (1.0 if mode == 0 else float(np.abs(v).max()) + 1e-3), rtol=1e-4 if mode == 0 else rt, err_msg=k)
