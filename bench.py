#!/usr/bin/env python3
"""Headline benchmark of the RegT-GCN hot path on MI355X (contract: see the task README / DESIGN.md section 6).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], "cfg-3"): synthetic regional graph, 100 000 nodes / 1 000 000
directed edges / 8 regions / 32 node features / T = 12 periods / horizon 1, fp32, per GPU.
A "step" is what the reference's run.py::train() does per snapshot: forward, mean((out-y)^2),
backward with gradients accumulating (run.py:178-191); the optimiser (RMSprop, run.py:145) steps once
per epoch, here once at the end of the K timed steps, inside the timed region.

N > 1: weak scaling, region-sharded (BASELINE.json configs[4] shape of growth).  The global graph has N*100k nodes and
N*8 regions; rank g owns 8 regions, exchanges the packed halo rows over RCCL every step (all-to-all, one step ahead on
a side stream) and all-reduces the gradient buffer once before the optimiser step.  ``value`` counts 100k-node shard
snapshots per second over all ranks (= N * steps / time).  ``--scaling strong`` instead splits the ONE 100k-node graph by
regions (configs[3]: 8 regions, one per GPU at N = 8); ``value`` is then whole-graph snapshots per second.

The JSON line carries ``roofline`` (dominant kernel, measured with HIP events recorded on the launch
stream by the library itself: regt_profile_*) and ``cpu_baseline`` (the oracle's eager-faithful CPU
path timed on the host cores of this box, rank 0, N = 1 only, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs, 2.4 GHz
PEAK_HBM_GBS = 8000.0             # HBM3E spec peak (6.3 TB/s is the measured achievable copy rate)

# HBM bytes per launch of the dominant kernels at cfg-3 on one GPU, from rocprofv3 PMC passes of this same command
# (profiles/r01_d_pmc_hbm_summary.txt): 2 x FETCH_SIZE (gfx950 reports half the bytes of 16-B/lane streaming reads --
# MI355X_MICROARCH.md, HBM section; calibrated here on cell_bwd, whose 3.7 GB of float4 reads show as 2.08e6 KB) + WRITE_SIZE.
PMC_TRAFFIC_CFG3 = {
    "gemm_gates": 2 * 1.255e6 * 1024 + 3.600e6 * 1024,        # h (twice: A operand and R-half epilogue) + Âx; ZR + q written
    "dgrad_gates": 2 * 2.558e6 * 1024 + 1.200e6 * 1024,
    "wgrad_Uzr": 2 * 2.433e6 * 1024 + 3.564e4 * 1024,
    "spmm": 2 * 2.055e5 * 1024 + 3.000e5 * 1024,              # profiles/r01_d_pmc_hbm_summary.txt (inside the step)
}

WORKLOADS = {
    # name: (nodes, edges, regions, F, T, O) per GPU
    "cfg3": (100_000, 1_000_000, 8, 32, 12, 1),
    "small": (20_000, 200_000, 8, 32, 12, 1),       # quick functional run
    "cfg3r64": (100_000, 1_000_000, 64, 32, 12, 1), # cfg-3 with the region count of the 8-GPU global graph (compose cost check)
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every GPU owns a full workload-sized shard (global graph N times larger, the default "
                         "the driver measures); strong = the ONE workload graph split by regions across the GPUs")
    ap.add_argument("--force-shard-path", action="store_true",
                    help="N = 1 only: run the region-shard code path (packed input, halo pipeline, RCCL calls) with a 1-rank "
                         "process group -- a rehearsal of what N > 1 executes, not a measurement")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-split-leg", action="store_true", help="skip the secondary bf16x3-split measurement")
    ap.add_argument("--no-tpims-leg", action="store_true", help="skip the secondary TPIMS-scale (configs[1]) measurement")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-stage HIP events")
    ap.add_argument("--cpu-baseline-only", action="store_true")
    return ap.parse_args()


def cpu_baseline(nodes, edges, regions, F, T, O, seed=42):
    """Oracle (op-for-op restatement of the reference CPU path) on the host cores.

    A full T=12 step at this size needs > 57 GB and ~90 s, so the bounded sample
    is ONE period (T=1) of the same graph, forward + loss + backward; periods are independent in
    the reference (RegionalTemporalGCN.py:135-148), so the step time is T x the period time."""
    import regtgcn_amd as R
    from oracle import model as M
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(avail, 16)          # a 1-GPU box owns a 16-core share of the host (more threads only oversubscribe)
    torch.set_num_threads(cores)
    g = R.data.synthetic_regional_graph(nodes, edges, regions, seed=seed)
    (x, y), = R.data.synthetic_snapshots(nodes, F, 1, O, 1, seed=seed)
    p = M.init_params("RegionalTemporalGCN", F, 1, O, num_nodes=nodes, num_regions=regions, seed=seed)
    p = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    t0 = time.perf_counter()
    pred, _ = M.regional_temporal_gcn(p, x, g.edge_index, g.region_index, g.region_attr)
    loss = torch.mean((pred - y) ** 2)
    loss.backward()
    dt = time.perf_counter() - t0
    return {"value": 1.0 / (dt * T), "unit": "snapshots/s", "cores": cores, "kind": "port",
            "sample": f"1 of {T} periods (T=1 forward+loss+backward, {dt:.2f} s) of the same {nodes}-node/{edges}-edge/"
                      f"{regions}-region graph; periods are independent, step time = {T} x period time"}


def tpims_leg(dev, steps=300, warmup=30, with_cpu=True):
    """BASELINE.json configs[1]: the reference's own graph (TPIMS, 104 sites after its site filter, 5 regions, F = 8), T = 12,
    horizon 1 -- the other configuration the metric names.  Same step semantics as the headline loop; this size is bound
    by the latency of ~45 dependent small kernels, not by any roofline."""
    import regtgcn_amd as R
    z = np.load(os.path.join(ROOT, "tests", "golden", "tpims_fixture.npz"))
    fx = {k: torch.from_numpy(z[k]) for k in z.files if z[k].ndim > 0}
    regs = ("IA", "KS", "KY", "OH", "WI")
    n, T, O = fx["node_data"].shape[0], 12, 1
    torch.manual_seed(42)
    model = R.RegionalTemporalGCN(8, n, T, O).to(dev)
    graph = model.prepare_graph(fx["edge_index"].to(dev), [fx[f"edge_{r}_index"].to(dev) for r in regs],
                                [fx[f"edge_{r}_attr"].to(dev) for r in regs])
    xs, ys = R.data.snapshot_windows(fx["node_data"], T, O)
    xs, ys = [x.to(dev) for x in xs], [y.to(dev) for y in ys]
    opt = torch.optim.RMSprop(model.parameters(), lr=1e-3, weight_decay=1e-4)

    def run(k):
        for i in range(k):
            pred, _ = model.forward_prepared(xs[i % len(xs)], graph)
            torch.mean((pred - ys[i % len(xs)]) ** 2).backward()
        opt.step()
        opt.zero_grad(set_to_none=False)
        torch.cuda.synchronize()

    run(warmup)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    # the same steps through functional.FusedTrainStep (three C-ABI calls + one axpy per step, no autograd)
    stepper = R.functional.FusedTrainStep(model, graph, 8, T)

    def run_fused(k):
        for i in range(k):
            stepper(xs[i % len(xs)], ys[i % len(xs)])
        opt.step()
        stepper.zero_grad()
        torch.cuda.synchronize()

    run_fused(warmup)
    t0 = time.perf_counter()
    run_fused(steps)
    dtf = time.perf_counter() - t0
    cpu = None
    if with_cpu:                                # the oracle (CPU restatement of the reference path) on the same snapshots
        from oracle import model as M
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        p = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        ri, rw = [fx[f"edge_{r}_index"] for r in regs], [fx[f"edge_{r}_attr"] for r in regs]
        xc, yc = [x.cpu() for x in xs[:4]], [y.cpu() for y in ys[:4]]
        k_cpu = 12
        for i in range(k_cpu + 2):
            if i == 2:
                t1 = time.perf_counter()
            pr, _ = M.regional_temporal_gcn(p, xc[i % 4], fx["edge_index"], ri, rw)
            torch.mean((pr - yc[i % 4]) ** 2).backward()
        cpu = {"value": k_cpu / (time.perf_counter() - t1), "unit": "snapshots/s", "kind": "port", "cores": torch.get_num_threads(),
               "sample": f"{k_cpu} forward+loss+backward steps of the oracle on the same snapshots"}
    return {"value": steps / dt, "unit": "snapshots/s", "ms_per_step": 1e3 * dt / steps, "steps": steps,
            "fused_train_step": {"value": steps / dtf, "unit": "snapshots/s", "ms_per_step": 1e3 * dtf / steps},
            "cpu_baseline": cpu,
            "workload": f"TPIMS fixture: {n} nodes / {fx['edge_index'].shape[1]} edges / 5 regions, F=8, T={T}, O={O} (BASELINE configs[1]); "
                        "launch-latency-bound"}


def stage_flops(stage, M, C, F):
    return {
        "gemm_gates": 2.0 * M * 2 * C * (C + F), "gemm_candidate": 2.0 * M * C * (C + F),
        "gemm_regional": 2.0 * M * C * 2 * F, "dgrad_candidate": 2.0 * M * C * C, "dgrad_gates": 2.0 * M * C * 2 * C,
        "wgrad_Uh": 2.0 * M * C * C, "wgrad_Uzr": 2.0 * M * 2 * C * C, "wgrad_Gh": 2.0 * M * C * F,
        "wgrad_Gzr": 2.0 * M * 2 * C * F, "wgrad_A0": 2.0 * M * C * F, "wgrad_Ar": 2.0 * M * C * F, "wgrad_A0_Ar": 2.0 * M * C * 2 * F,
    }.get(stage)


def main():
    args = parse()
    nodes, edges, regions, F, T, O = WORKLOADS[args.workload]
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(nodes, edges, regions, F, T, O)))
        return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU path")
    backend = os.environ.get("REGT_BENCH_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 flow on a 1-GPU box
    if backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    force_shard = args.force_shard_path and world == 1
    if force_shard:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        os.environ["REGT_DIST_FORCE"] = "1"          # collectives run even though the group has one rank
    if world > 1 or force_shard:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import regtgcn_amd as R
    from regtgcn_amd import _lib
    lib = R.load_library()

    # ---- data: global graph (weak: world x the workload shape; strong: the workload itself), this rank's shard ------
    weak = args.scaling == "weak" or world == 1
    if weak:
        gnodes, gedges, gregions = nodes * world, edges * world, regions * world
    else:
        if regions % world:
            raise SystemExit(f"--scaling strong needs the {regions} regions to divide evenly over {world} GPUs")
        gnodes, gedges, gregions = nodes, edges, regions
    g = R.data.synthetic_regional_graph(gnodes, gedges, gregions, seed=42)
    rpg = gregions // world                                                  # regions per GPU
    owner_bounds = np.asarray(g.region_bounds[::rpg], dtype=np.int64)        # contiguous region blocks
    nodes = int(owner_bounds[rank + 1] - owner_bounds[rank])                 # this rank's node count from here on
    C = R.nn.HIDDEN
    torch.manual_seed(42)                     # same random-init weights on every rank (run.py:71)
    model = R.RegionalTemporalGCN(node_features=F, num_nodes=nodes, periods=T, output_dim=O, num_regions=gregions)
    model = model.to(dev)
    n_snap = 4
    # the global snapshot is the concatenation of per-rank row blocks, each drawn from its own seeded stream, so a rank
    # only ever materialises its own 100k rows (8 ranks x the 800k-node tensor would be ~40 GB of host memory)
    snaps = R.data.synthetic_snapshots(nodes, F, T, O, n_snap, seed=42 + 1000 * rank)
    xs = [x.to(dev) for x, _ in snaps]
    ys = [y.to(dev) for _, y in snaps]
    del snaps
    if world == 1 and not force_shard:
        graph = R.prepare_graph(g.edge_index.to(dev), None, [t.to(dev) for t in g.region_index],
                                [t.to(dev) for t in g.region_attr], nodes)
        shard = None
    else:
        region_owner = [r // rpg for r in range(gregions)]
        shard = R.dist.build_shard(g.edge_index, g.region_index, g.region_attr, gnodes, owner_bounds, region_owner,
                                   rank, world, dev)
        graph = shard.graph
        pipe = R.dist.HaloPipeline(shard, T, F, dev)
        pipe.submit(0, xs[0])
    opt = torch.optim.RMSprop(model.parameters(), lr=1e-3, weight_decay=1e-4)   # run.py:145
    params = list(model.parameters())
    inv_count = 1.0 / float(gnodes * O)

    def step(i):
        x, y = xs[i % n_snap], ys[i % n_snap]
        if shard is None:
            pred, _ = model.forward_prepared(x, graph)
        else:
            # the halo rows of snapshot i were exchanged while step i-1 computed; start snapshot i+1's exchange now
            xp_ext = pipe.acquire(i % 2)
            pipe.submit((i + 1) % 2, xs[(i + 1) % n_snap])
            pred, _ = model.forward_packed(xp_ext, graph)
        loss = ((pred - y) ** 2).sum() * inv_count        # mean over the GLOBAL graph (run.py:180)
        loss.backward()
        if shard is not None:
            pipe.release(i % 2)
        return loss

    def epoch_end():
        R.dist.allreduce_gradients(params)
        opt.step()
        opt.zero_grad(set_to_none=False)

    def fence():
        torch.cuda.synchronize()
        if world > 1 or force_shard:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    for i in range(args.warmup):
        loss = step(i)        # keep the previous step's graph alive exactly as the timed loop does, so that both
                              # activation workspaces exist before timing starts (an 11 GB hipMalloc can take 0.3 s)
    if args.warmup:
        epoch_end()
    fence()
    alloc0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    profile = not args.no_profile
    if profile:
        lib.regt_profile_enable(1)
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    epoch_end()
    fence()
    dt = time.perf_counter() - t0
    stages = {}
    if profile:
        lib.regt_profile_enable(0)
        buf = (__import__("ctypes").c_char * 16384)()
        _lib.check(lib.regt_profile_collect(buf, 16384), "regt_profile_collect")
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            stages[name] = (int(cnt), float(ms))
    final_loss = float(loss.detach())
    # Secondary, opt-in arithmetic (never the headline): the same K steps with the flat GEMMs on the bf16 matrix pipe
    # through an exact 3-way bf16 split of both fp32 operands (gemm_split.h).  N = 1 only, after the timed region.
    split_dt = None
    if world == 1 and not args.no_split_leg:
        prev = lib.regt_set_gemm_mode(1)
        for i in range(max(args.warmup, 1)):
            loss_s = step(i)
        epoch_end()
        fence()
        t1 = time.perf_counter()
        for i in range(args.steps):
            loss_s = step(i)
        epoch_end()
        fence()
        split_dt = time.perf_counter() - t1
        lib.regt_set_gemm_mode(prev)
        del loss_s
    tpims = None
    if world == 1 and not args.no_tpims_leg:
        loss = None                         # let the headline model's workspaces go back to the pool
        tpims = tpims_leg(dev, with_cpu=not args.no_cpu_baseline)
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1 or force_shard:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        M = nodes * T
        out = {
            "metric": "training steps/sec (graph-snapshots/sec)", "value": (world if weak else 1) * args.steps / dt, "unit": "snapshots/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "fp32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: synthetic regional graph, {WORKLOADS[args.workload][0]} nodes / {edges} edges / {regions} regions "
                                   f"{'per GPU' if weak else 'in total, split by regions over the GPUs'}, F={F}, T={T}, O={O}, hidden=256; RegionalTemporalGCN forward+MSE+backward per "
                                   "snapshot, RMSprop step once per K steps (run.py semantics)",
                       "global_nodes": gnodes, "global_edges": gedges, "global_regions": gregions,
                       "parallelism": "single GPU" if world == 1 else f"region-sharded x{world}: halo-row all-to-all per step (one step ahead, side stream) + 1 grad all-reduce",
                       "final_loss": final_loss,
                       "device_allocs_in_timed_region": torch.cuda.memory_stats().get("num_device_alloc", 0) - alloc0},
        }
        if stages:
            per = {k: {"launches": c, "avg_ms": ms / c} for k, (c, ms) in stages.items()}
            mfma = [(ms, k) for k, (c, ms) in stages.items() if stage_flops(k, M, C, F)]
            tot_ms, dom = max(mfma)
            cnt = stages[dom][0]
            avg_s = tot_ms / cnt * 1e-3
            achieved = stage_flops(dom, M, C, F) / avg_s / 1e12
            traffic = PMC_TRAFFIC_CFG3.get(dom) if (args.workload == "cfg3" and world == 1) else None
            out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MATRIX_TFLOPS,
                               "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MATRIX_TFLOPS, "traffic": traffic,
                               "avg_ms": avg_s * 1e3, "flops_per_launch": stage_flops(dom, M, C, F)}
            if "spmm" in stages:
                c, ms = stages["spmm"]
                W = T * F
                dual = graph.m_rowptr is not None and W % 32 == 0
                nnz = int(graph.m_col.numel()) if dual else int(graph.col.numel())
                x_rows = nodes if shard is None else shard.topo.x_rows
                # read X once + CSR entries (col + 1 or 2 weights) + rowptr + write both outputs
                algo = x_rows * W * 4 + nnz * (12 if dual else 8) + (nodes + 1) * 4 * (1 if dual else 2) + 2 * nodes * W * 4
                gbs = algo / (ms / c * 1e-3) / 1e9
                out["roofline_spmm"] = {"kernel": "spmm_dual_panel (A_hat x and L~ x in one gather pass, width T*F)" if dual else "spmm_csr (stacked [A_hat; L~] x, width T*F)", "bound": "hbm", "achieved": gbs,
                                        "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                                        "traffic": PMC_TRAFFIC_CFG3["spmm"] if (args.workload == "cfg3" and world == 1 and dual) else None,
                                        "avg_ms": ms / c, "bytes_per_launch": algo}
            gemm_ms = sum(ms for k, (c, ms) in stages.items() if stage_flops(k, M, C, F))
            gemm_fl = sum(stage_flops(k, M, C, F) * c for k, (c, ms) in stages.items() if stage_flops(k, M, C, F))
            out["mfma_all_gemms"] = {"achieved": gemm_fl / (gemm_ms * 1e-3) / 1e12, "peak": PEAK_FP32_MATRIX_TFLOPS,
                                     "unit": "TFLOP/s", "share_of_step": gemm_ms / (dt * 1e3)}
            out["stages"] = per
        if split_dt is not None:
            out["opt_in_bf16x3_split"] = {
                "value": args.steps / split_dt, "unit": "snapshots/s", "ms_per_step": 1e3 * split_dt / args.steps,
                "note": "REGT_GEMM_MODE=bf16x3: gate/candidate/regional GEMMs, their data gradients and the wide weight gradients as "
                        "6 bf16 partial products of an exact 3-way bf16 split, fp32 accumulate; passes the same parity suite; "
                        "NOT the headline value"}
        if tpims is not None:
            out["tpims_configs1"] = tpims
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nodes, edges, regions, F, T, O)
            out["cpu_baseline"]["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1 or force_shard:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
