#!/usr/bin/env python3
"""bench.py — RAD HNSW neighbor-expansion throughput on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on; it fits one GPU):
100M x 1024-bit fingerprints resident in HBM; an HNSW graph built over them on the GPU by the
library's own insert kernels (connectivity 8, level-0 width 16, expansion_add 64: 35 s;
`--graph synthetic` swaps in a closed-form adjacency generator that is set up in 20 ms); `nq`
independent best-first RAD traversals (Tanimoto-scored; nq defaults to twice the number the device
holds resident at once: 2 x 16384 on MI355X), each run to n_to_score = 100k.  One "step" = one pass
of the hot path over one batch of nq queries: state re-arm (query upload, epoch bump) + traversal
kernel launch(es) to completion.  The fingerprints are synthetic (closed-form generator on the
device: no dataset can be downloaded here); corpus, graph and state are resident in HBM before the
timed region starts.

N > 1 (one process per GPU, launched by torch.distributed.run): weak scaling — every rank
holds its own 100M-row shard (N x 100M rows in total) with its shard-local graph and runs the
same query batch against it; the global budget N x n_to_score is split over the shards by the
per-round RCCL all-gather of frontier scores (rad_amd/sharded.py); value = expansions of all
ranks / max-over-ranks time.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

# the host driver of these boxes only supports dmabuf IPC: RCCL across processes needs it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", dest="n", type=int, default=100_000_000, help="rows per GPU")
    ap.add_argument("--ndim", type=int, default=1024)
    ap.add_argument("--connectivity", type=int, default=8)
    ap.add_argument("--nq", type=int, default=0,
                    help="concurrent traversals per GPU per step (0 = what the device holds resident at once)")
    ap.add_argument("--n-to-score", type=int, default=100_000)
    ap.add_argument("--corpus-mode", type=int, default=1, help="0 dense Bernoulli(0.5), 1 clustered sparse")
    ap.add_argument("--graph", choices=["built", "synthetic"], default="built",
                    help="adjacency: a real HNSW graph built on the GPU by Index.add over the synthetic rows (default; "
                         "35 s for 100M) or the closed-form generator on the device (set up in 20 ms)")
    ap.add_argument("--expansion-add", type=int, default=64, help="expansion_add of the built graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-traversals", type=int, default=0, help="0 = 32 per host core")
    ap.add_argument("--exchange", choices=["rccl", "gloo"], default="rccl",
                    help="N>1 exchange step: rccl (product path) or gloo (rehearsal of the N>1 logic on a box with one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    return ap.parse_args()


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_device:
        local_rank = 0
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    dist = None
    if world > 1:
        # torch is plumbing only: rendezvous (gloo), barrier, max-over-ranks, and handing the
        # RCCL unique id of the library's own communicator to the other ranks
        import torch
        import torch.distributed as dist_mod
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group(backend="gloo")
        dist = dist_mod

    from rad_amd import _lib
    from rad_amd.device import DeviceIndex, DeviceTraversal, RcclComm
    from rad_amd.sharded import ShardedTraversal

    _lib.lib()
    if _lib.device_count() <= local_rank:
        raise SystemExit("bench.py needs an MI355X per rank; there is no CPU fallback")

    comm = None
    allgather = None
    exchange_used = args.exchange
    if dist is not None and args.exchange == "rccl":
        import torch
        err = ""
        try:
            box = [RcclComm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            comm = RcclComm(rank, world, box[0], local_rank)
            probe = comm.allgather_u64(np.array([rank + 1], np.uint64))
            if probe.reshape(-1).tolist() != list(range(1, world + 1)):
                raise RuntimeError(f"RCCL all-gather self-test returned {probe.reshape(-1).tolist()}")
            allgather = comm.allgather_u64
        except Exception as e:   # noqa: BLE001 - reported in the JSON line, never silent
            err = f"{type(e).__name__}: {e}"
        # all ranks take the same exchange: if RCCL failed anywhere, the 16 B/traversal control
        # exchange moves to gloo and the JSON line says so (the data path stays on the GPUs)
        bad = torch.tensor([1 if err else 0])
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            errs = [None] * world
            dist.all_gather_object(errs, err)
            exchange_used = "gloo (RCCL unavailable: " + "; ".join(sorted({x for x in errs if x}))[:300] + ")"
            allgather = None
            comm = None
    if dist is not None and allgather is None:
        import torch

        def allgather(a):
            t = torch.from_numpy(np.ascontiguousarray(a, np.uint64).view(np.int64))
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            return np.stack([o.numpy().view(np.uint64) for o in outs])

    def barrier_sync():
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    n, ndim, M = args.n, args.ndim, args.connectivity
    # every shard is an independent n-row corpus (own seed) with its shard-local graph
    idx = DeviceIndex(ndim, M, 2 * M, args.expansion_add, device=local_rank)
    idx.synth_vectors(n, seed=20260101 + rank, mode=args.corpus_mode)
    if args.graph == "synthetic":
        idx.synth_graph(seed=777 + rank)
    else:
        # a real HNSW graph: the rows go through Index.add's insert kernels (the rows come back to the
        # host once, add() takes host rows as the reference's does)
        X = np.empty((n, idx.row_bytes), np.uint8)
        for f in range(0, n, 4_000_000):
            c = min(4_000_000, n - f)
            X[f:f + c] = idx.read_vectors(f, c)
        idx.close()
        idx = DeviceIndex(ndim, M, 2 * M, args.expansion_add, device=local_rank)
        t_build = time.perf_counter()
        for f in range(0, n, 5_000_000):
            idx.add_rows(X[f:f + 5_000_000], seed=777 + rank, max_batch=16384)
        t_build = time.perf_counter() - t_build
        del X
    info = idx.info()
    B = info.row_stride
    auto_nq = args.nq <= 0
    if auto_nq:
        # two resident rounds: traversals end at different times (12.1-12.7k expansions each), and the
        # second round's workgroups fill the slots the early finishers leave: +6.5 % over one exactly
        # resident round (16384 on MI355X); falls back to one round if the state does not fit in HBM
        args.nq = 2 * idx.traversal_capacity()

    # query batches: rows of shard 0's corpus — every rank regenerates them from the closed-form
    # definition, so all ranks run the SAME queries; a different batch per step
    n_batches = args.warmup + args.steps
    qrng = np.random.default_rng(4242)
    batches = []
    qsrc = idx if rank == 0 else DeviceIndex(ndim, M, 2 * M, 64, device=local_rank)
    for b in range(n_batches):
        first = int(qrng.integers(0, n - args.nq))
        if rank == 0:
            batches.append(idx.read_vectors(first, args.nq))
        else:
            qsrc.synth_vectors(args.nq, seed=20260101, mode=args.corpus_mode, first_row=first, n_total=n)
            batches.append(qsrc.read_vectors(0, args.nq))
    if rank != 0:
        qsrc.close()
    # sharded: the global budget is world x n_to_score, split over the shards round by round
    # (rad_amd/sharded.py); the local state is sized with 25 % headroom over the even split
    local_cap = args.n_to_score if world == 1 else args.n_to_score + args.n_to_score // 4
    trav = None
    try:
        trav = DeviceTraversal(idx, batches[0], local_cap)
    except _lib.RadHipError as e:
        if not (auto_nq and e.code == -4):      # RADHIP_E_NOMEM
            raise
    if auto_nq:
        fits = 1 if trav is not None else 0
        if dist is not None:                    # every rank runs the same batch size
            import torch
            f = torch.tensor([fits])
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            fits = int(f.item())
        if not fits:
            if trav is not None:
                trav.close()
            args.nq //= 2
            batches = [b[:args.nq] for b in batches]
            trav = DeviceTraversal(idx, batches[0], local_cap)
    exch = {"rounds": 0, "bytes": 0}

    def step(b):
        trav.reset(batches[b])
        if allgather is None:
            running = trav.run(0)
            assert running == 0
        else:
            st_ = ShardedTraversal(trav, allgather, rank, world, args.n_to_score * world, local_cap)
            st_.run()
            exch["rounds"] += st_.rounds
            exch["bytes"] += st_.exchanged_bytes
        ms, launches = trav.kernel_time()
        st = trav.stats()
        return ms, launches, int(st.n_pops.sum()), int(st.n_scored.sum()), int(st.n_nbr.sum())

    for w in range(args.warmup):
        step(w)

    barrier_sync()
    t0 = time.perf_counter()
    k_ms = 0.0
    k_launches = pops = evals = nbrs = 0
    for s in range(args.steps):
        ms, launches, p, e, nb = step(args.warmup + s)
        k_ms += ms
        k_launches += launches
        pops += p
        evals += e
        nbrs += nb
    barrier_sync()
    elapsed = time.perf_counter() - t0

    tot = np.array([elapsed, float(pops), float(evals), k_ms, float(k_launches)], dtype=np.float64)
    if dist is not None:
        import torch
        tt = torch.tensor(tot)
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max = float(tmax[0].item())
        pops_all, evals_all = float(tsum[1].item()), float(tsum[2].item())
    else:
        elapsed_max, pops_all, evals_all = elapsed, float(pops), float(evals)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # roofline of the dominant kernel (trav_kernel) on rank 0: algorithmic bytes per launch =
    # evals x (B + 4) [fingerprint row + its u32 slot in the adjacency row] + pops x 4 [degree word]
    alg_bytes_per_launch = (evals * (B + 4) + pops * 4) / max(k_launches, 1)
    avg_launch_ms = k_ms / max(k_launches, 1)
    achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    traffic = None
    req_bound = None
    prof = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(prof):
        try:
            with open(prof) as f:
                pj = json.load(f)
            if (pj.get("n") == n and pj.get("nq") == args.nq and pj.get("n_to_score") == args.n_to_score
                    and pj.get("graph", "synthetic") == args.graph):
                traffic = pj.get("hbm_bytes_per_launch")
                rb = pj.get("request_bound")
                if rb and world == 1:
                    kernel_rate = pops / (k_ms * 1e-3) if k_ms > 0 else 0.0
                    req_bound = {"ceiling_expansions_per_s": rb["ceiling_expansions_per_s"],
                                 "kernel_expansions_per_s": kernel_rate,
                                 "frac": kernel_rate / rb["ceiling_expansions_per_s"],
                                 "request_equivalents_per_expansion": rb["request_equivalents_per_expansion"],
                                 "device_random_requests_per_s": rb["device_random_requests_per_s"],
                                 "source": "profiles/traffic_latest.json (PMC + scripts/hbm_random.hip, measured offline)"}
        except Exception:
            traffic = None

    out = {
        "metric": "neighbor-expansions/sec (1024-bit Tanimoto) + HBM GB/s vs roofline",
        "value": pops_all / elapsed_max,
        "unit": "expansions/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed_max / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64 popcount (integer)",
        "data": "synthetic",
        "config": {
            "workload": f"{n * world // 1_000_000}M x {ndim}-bit fingerprints ({n // 1_000_000}M per GPU resident in HBM), "
                        f"connectivity={M} (level-0 width {2 * M}), {args.nq} concurrent best-first RAD traversals per GPU "
                        f"to n_to_score={args.n_to_score}, synthetic corpus, "
                        + ("closed-form synthetic graph" if args.graph == "synthetic" else
                           f"HNSW graph built on the GPU (expansion_add={args.expansion_add}, {t_build:.0f} s)"),
            "rows_per_gpu": n, "ndim": ndim, "connectivity": M, "nq_per_gpu": args.nq,
            "n_to_score": args.n_to_score, "corpus_mode": args.corpus_mode,
            "parallelism": ("1 process per GPU, corpus sharded by contiguous row range, global n_to_score = "
                            f"{world} x {args.n_to_score} split over shards by a per-round all-gather of frontier "
                            f"scores + scored counts, exchange = {exchange_used}") if world > 1 else "single GPU",
            "exchange_rounds_per_step": (exch["rounds"] / max(args.steps + args.warmup, 1)) if world > 1 else 0,
        },
        "evals_per_s": evals_all / elapsed_max,
        "evals_per_expansion": evals_all / max(pops_all, 1.0),
        "roofline": {
            "bound": "hbm", "kernel": trav.kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes_per_launch, "avg_launch_ms": avg_launch_ms,
            "launches": k_launches,
            # real HBM bytes (PMC, measured offline: profiles/traffic_latest.json) over this run's launch time
            "hbm_real_gbs": (traffic / (avg_launch_ms * 1e-3) / 1e9) if traffic else None,
            "hbm_real_frac": (traffic / (avg_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            # the bound that explains frac: HBM serves random requests at a fixed rate whatever their size
            "request_bound": req_bound,
        },
    }

    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(idx, batches[args.warmup], args)

    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def host_cores() -> int:
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(idx, queries, args):
    """The oracle (C restatement of the reference control flow, pthreads over independent
    traversals) timed on this box's host cores, on a bounded sample of the same workload:
    same corpus + graph (copied back from HBM), same n_to_score, fewer traversals."""
    from oracle import rad_oracle as O
    O.build()
    cores = host_cores()
    info = idx.info()
    n = info.n
    X = np.empty((n, idx.row_bytes), np.uint8)
    chunk = 4_000_000
    for f in range(0, n, chunk):
        c = min(chunk, n - f)
        X[f:f + c] = idx.read_vectors(f, c)
    levels, adj0, upper_row, adjU = idx.read_graph()
    g = O.Graph(int(n), int(info.connectivity_base), int(info.connectivity), int(info.max_level),
                int(info.entry), levels, adj0, upper_row, adjU)
    nt = args.cpu_traversals or 32 * cores
    nt = min(nt, queries.shape[0])
    t0 = time.perf_counter()
    n_scored, n_pops, n_nbr = O.rad_traverse_many(g, X, queries[:nt], args.n_to_score, cores)
    dt = time.perf_counter() - t0
    return {"value": float(n_pops.sum()) / dt, "unit": "expansions/s", "cores": cores, "kind": "port",
            "evals_per_s": float(n_scored.sum()) / dt,
            "sample": f"{nt} of the {queries.shape[0]} traversals of one step (same corpus, graph, n_to_score), "
                      f"{dt:.1f} s wall on {cores} threads; usearch-shaped C restatement (oracle/), not usearch"}


if __name__ == "__main__":
    main()
