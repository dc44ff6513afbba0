"""The RCCL calls of the sharded path on a real 'nccl' process group (world_size 1 is all a one-GPU box offers): the
halo all-to-all writes into a VIEW of the extended input from a side stream, the gradient all-reduce runs on a flat
buffer, barrier + max-reduce as in bench.py.  Catches API misuse (views, split lists, streams, device_id) before the
8-GPU run; the multi-rank data flow itself is covered by the gloo tests."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        import numpy as np
        import regtgcn_amd as R
        n_local, halo, t, f = 300, 40, 6, 8
        topo = R.dist.ShardTopology(0, 1, 0, n_local, [np.arange(halo, dtype=np.int64)], [np.arange(halo, dtype=np.int64)])
        xp = torch.zeros(n_local + halo, t * f, device=dev)
        xp[:n_local] = torch.rand(n_local, t * f, device=dev)
        send_idx = torch.arange(10, 10 + halo, device=dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            send = xp.index_select(0, send_idx)
            recv = xp[n_local:]                                     # contiguous view of the extended input
            dist.all_to_all_single(recv, send, topo.recv_splits, topo.send_splits)
            ev = torch.cuda.Event()
            ev.record(side)
        torch.cuda.current_stream(dev).wait_event(ev)
        assert torch.equal(xp[n_local:], xp[10:10 + halo])
        p = [torch.nn.Parameter(torch.zeros(5, 3, device=dev)), torch.nn.Parameter(torch.zeros(7, device=dev))]
        p[0].grad = torch.full((5, 3), 2.0, device=dev)
        p[1].grad = torch.arange(7, dtype=torch.float32, device=dev)
        R.dist.allreduce_gradients(p)
        assert torch.equal(p[0].grad, torch.full((5, 3), 2.0, device=dev))
        tot = R.dist.allreduce_sum(torch.tensor([1.5, 2.5], device=dev))
        assert tot.tolist() == [1.5, 2.5]
        # graph preparation of an own-rows shard (dist.build_shard): the id-list exchange (int64 all_gather + all_to_all_single
        # with split lists, here a rank exchanging with itself) and the all-reduce of the published D^-1/2 vector
        back = R.dist.exchange_need_lists([np.arange(3, 20, dtype=np.int64)], 1)
        assert len(back) == 1 and np.array_equal(back[0], np.arange(3, 20))
        assert R.dist.exchange_need_lists([np.zeros(0, dtype=np.int64)], 1)[0].size == 0
        dis = torch.rand(100_000, device=dev)
        keep = dis.clone()
        R.dist.allreduce_sum(dis)
        assert torch.equal(dis, keep)
        dist.barrier()
        tmax = torch.tensor([3.0], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        torch.cuda.synchronize()
        dist.destroy_process_group()
        q.put("ok")
    except Exception as e:  # noqa: BLE001
        import traceback
        q.put(f"FAIL {type(e).__name__}: {e}\n{traceback.format_exc()}")


def test_rccl_calls_of_the_sharded_path():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=240)
    p.join(timeout=60)
    assert res == "ok", res


def test_bench_shard_path_with_one_rank():
    """bench.py --force-shard-path: the code path N > 1 runs (packed input, HaloPipeline on a side stream, all-to-all,
    gradient all-reduce, barrier, max-reduce) on the real 'nccl' backend with a one-rank group."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    res = subprocess.run([sys.executable, "bench.py", "--force-shard-path", "--workload", "small", "--steps", "3", "--warmup", "1",
                          "--no-cpu-baseline", "--no-split-leg", "--no-tpims-leg"], cwd=root, env=env, capture_output=True, text=True,
                         timeout=400)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and np.isfinite(out["config"]["final_loss"])
    assert out["multi_gpu"]["world_size"] == 1 and out["multi_gpu"]["backend"] == "nccl" and out["multi_gpu"]["rank0_halo_rows"] == 0


@pytest.mark.parametrize("launcher", ["torchrun", "self"])
def test_bench_two_ranks_on_one_gpu(launcher):
    """The driver's N > 1 command (python -m torch.distributed.run ... bench.py --gpus 2) and the plain `python bench.py --gpus 2`
    (bench.py then starts the ranks itself, before it touches the GPU), with two ranks sharing this box's one GPU
    (REGT_BENCH_BACKEND=gloo: collectives staged through the host): exit code 0, ONE JSON line from rank 0, the N > 1 block."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    args = ["bench.py", "--gpus", "2", "--workload", "small", "--steps", "3", "--warmup", "1"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + args
    else:
        cmd = [sys.executable] + args
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["REGT_BENCH_BACKEND"] = "gloo"
    res = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "strong" and np.isfinite(out["config"]["final_loss"])
    mg = out["multi_gpu"]
    assert mg["world_size"] == 2 and mg["backend"] == "gloo" and mg["rank0_halo_rows"] > 0
    assert mg["rank0_halo_bytes_per_step"] == mg["rank0_halo_rows"] * 12 * 32 * 4
    assert mg["pack_and_exchange_side_stream"]["samples"] >= 3 and mg["grad_allreduce_ms"]["calls"] >= 1
