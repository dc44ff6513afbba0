"""Region-sharded path on CPU with the gloo backend (world_size 2): index logic of the shard topology,
the boundary-row exchange and the flat gradient all-reduce.  The compute in these tests is the oracle's
(the product has no CPU compute path); what is under test is regt-gcn_amd/dist.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import regtgcn_amd as R
from oracle import graph_ops as G


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_per, e_per, regions_per, width, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = n_per * world
        g = R.data.synthetic_regional_graph(n, e_per * world, regions_per * world, seed=7, p_intra=0.8)
        bounds = np.arange(world + 1, dtype=np.int64) * n_per
        topo = R.dist.shard_topology(g.edge_index.numpy(), bounds, rank, world)
        # the same lists from this rank's in-edges alone (need: local; send: one all-gather + one all-to-all between the ranks) --
        # what build_shard uses so that no rank walks the global edge list -- bit for bit
        ei_np = g.edge_index.numpy()
        mine_np = (ei_np[1] >= bounds[rank]) & (ei_np[1] < bounds[rank + 1])
        t2 = R.dist.topology_from_sources(ei_np[0][mine_np], bounds, rank, world)
        assert (t2.node_lo, t2.node_hi) == (topo.node_lo, topo.node_hi)
        assert all(np.array_equal(a, b) for a, b in zip(t2.need, topo.need)), "need lists differ"
        assert all(np.array_equal(a, b) for a, b in zip(t2.send, topo.send)), "send lists differ"
        gen = torch.Generator().manual_seed(3)
        x_glob = torch.randn(n, width, generator=gen)                 # "packed" rows of the global graph
        lo, hi = topo.node_lo, topo.node_hi
        xp = torch.zeros(topo.x_rows, width)
        xp[:n_per] = x_glob[lo:hi]
        R.dist.exchange_boundary_rows(xp, topo, torch.from_numpy(topo.send_index()))
        # every halo slot holds the row of the node the topology says it holds
        halo = topo.halo_ids()
        assert halo.size == topo.halo_rows > 0 and np.all((halo < lo) | (halo >= hi))
        assert torch.equal(xp[n_per:], x_glob[torch.from_numpy(halo)]), f"rank {rank}: halo rows wrong"
        assert sum(topo.send_splits) == topo.send_index().size and topo.send_splits[rank] == 0
        # local rows of A_hat x computed from the extended input == rows of the global product
        src, dst, w = G.gcn_norm_edges(g.edge_index, None, n, torch.float32)
        want = G.propagate(src, dst, w, x_glob, n)[lo:hi]
        mine = (dst >= lo) & (dst < hi)
        cols = torch.from_numpy(topo.remap_columns(src[mine].numpy(), bounds))
        got = torch.zeros(n_per, width).index_add_(0, dst[mine] - lo, w[mine].view(-1, 1) * xp.index_select(0, cols))
        assert float((got - want).abs().max()) < 1e-5
        # flat gradient all-reduce
        p = [torch.nn.Parameter(torch.zeros(5, 3)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
        p[0].grad = torch.full((5, 3), float(rank + 1))
        p[1].grad = torch.arange(7, dtype=torch.float32) * (rank + 1)
        R.dist.allreduce_gradients(p)
        tot = sum(range(1, world + 1))
        assert torch.equal(p[0].grad, torch.full((5, 3), float(tot)))
        assert torch.equal(p[1].grad, torch.arange(7, dtype=torch.float32) * tot)
        assert p[2].grad is None
        q.put((rank, "ok", (topo.send_splits, topo.recv_splits)))
    except Exception as e:  # noqa: BLE001
        q.put((rank, f"FAIL {type(e).__name__}: {e}", 0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 600, 5000, 3, 24, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
    by_rank = {r[0]: r[2] for r in res}
    for a in range(world):
        for b in range(world):
            assert by_rank[a][0][b] == by_rank[b][1][a]            # what a sends to b is what b expects from a


def test_topology_single_rank_has_no_halo():
    g = R.data.synthetic_regional_graph(300, 2000, 3, seed=1)
    topo = R.dist.shard_topology(g.edge_index.numpy(), np.array([0, 300]), 0, 1)
    assert topo.halo_rows == 0 and topo.x_rows == 300 and topo.send_index().size == 0
    cols = topo.remap_columns(np.array([0, 5, 299]), np.array([0, 300]))
    assert cols.tolist() == [0, 5, 299]


def test_region_chunks_cover_rows_without_straddling():
    reg = np.array([0] * 10 + [1] * 3 + [2] * 20, dtype=np.int32)
    tab, creg = R.graph.region_chunks(reg, 6)
    assert tab[0, 0] == 0 and tab[-1, 1] == 33 * 6
    assert np.all(tab[1:, 0] == tab[:-1, 1])
    for (a, b), r in zip(tab, creg):
        assert np.all(reg[a // 6:(b + 5) // 6] == r)


def test_node_regions_fill_and_overlap():
    a = torch.tensor([[0, 1], [1, 0]])
    b = torch.tensor([[4, 5], [5, 4]])
    own = R.graph.node_regions([a, b], 7)
    assert own.tolist() == [0, 0, 0, 0, 1, 1, 1]
    with pytest.raises(R.graph.OverlappingRegions):
        R.graph.node_regions([a, torch.tensor([[3], [1]])], 7)
