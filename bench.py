#!/usr/bin/env python3
"""bench.py — RAD HNSW neighbor-expansion throughput on MI355X.  No torch anywhere.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on; it fits one GPU):
100M x 1024-bit fingerprints resident in HBM; an HNSW graph built over them on the GPU by the library's
own insert kernels (connectivity 8, level-0 width 16, expansion_add 64: ~30 s); `nq` independent
best-first RAD traversals (Tanimoto-scored; nq defaults to twice what the device holds resident at
once: 2 x 16384 on MI355X), each run to n_to_score = 100k.  One "step" = one pass of the hot path over
one batch of nq queries: state re-arm (query upload, epoch bump) + traversal kernel launch to completion.
The fingerprints are synthetic (closed-form generator on the device: no dataset can be downloaded here);
`--corpus-mode 2` (default) is the hierarchical corpus — neighbourhood structure at every scale, the
built graph is a usable HNSW graph (recall figures in profiles/r02) — `--corpus-mode 1` is round 1's
two-level clustered corpus, measured as well and reported under "reference_corpus_r01".
Corpus, graph and state are resident in HBM before the timed region starts.

N > 1: `python bench.py --gpus N` spawns its own N rank processes (before any HIP call; also runs under
`python -m torch.distributed.run`, reading RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).  Rendezvous is a
plain TCP star (rad_amd/rendezvous.py); the exchange of the data path is RCCL inside the library.
  --mode replicas  (default) the independent units of this path are the traversals: every GPU holds the
                   whole corpus and graph (20 GB of 288), the query batches are split, no collective
  --mode sharded   BASELINE's partitioning: ONE graph over the whole corpus, rows and traversals
                   partitioned over the ranks, per frontier step an RCCL all-gather of the candidate
                   slots and a reduce-scatter of their scores; results bit-identical to one GPU
BOTH legs run in every N > 1 bench (the sharded one parity-checked against the single-GPU kernel); `value`
comes from --mode, the other leg is reported beside it.  Strict best-first over remote rows costs two
collectives per frontier step and ~10^4 steps per batch: it is the mode for a corpus that does not fit one
GPU, not the throughput mode ("if RCCL latency dominates: replicas ... state which mode
each number comes from").

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

# the host driver of these boxes only supports dmabuf IPC: RCCL across processes needs it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
METRIC = "neighbor-expansions/sec (1024-bit Tanimoto) + HBM GB/s vs roofline"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", dest="n", type=int, default=100_000_000, help="rows of the corpus (sharded / replicated over the GPUs)")
    ap.add_argument("--ndim", type=int, default=1024)
    ap.add_argument("--connectivity", type=int, default=8)
    ap.add_argument("--nq", type=int, default=0,
                    help="concurrent traversals per GPU per step (0 = twice what the device holds resident at once)")
    ap.add_argument("--n-to-score", type=int, default=100_000)
    ap.add_argument("--corpus-mode", type=int, default=2,
                    help="0 dense Bernoulli(0.5), 1 two-level clustered sparse (round 1), 2 hierarchical sparse")
    ap.add_argument("--graph", choices=["built", "synthetic"], default="built",
                    help="adjacency: an HNSW graph built on the GPU by Index.add (default) or the closed-form "
                         "generator (corpus mode 1 only; set up in 20 ms)")
    ap.add_argument("--expansion-add", type=int, default=64, help="expansion_add of the built graph")
    ap.add_argument("--table", choices=["auto", "hash", "group"], default="auto",
                    help="visited/scored table of the traversal kernel (auto = the library's default, the per-slot hash table; "
                         "group = the grouped table over the graph-locality layout, measured slower: profiles/r02)")
    ap.add_argument("--no-reference-corpus", action="store_true", help="skip the round-1 corpus leg (N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work the cpu_baseline sample should take at least")
    ap.add_argument("--mode", choices=["sharded", "replicas"], default="replicas", help="which N > 1 leg `value` reports (both always run)")
    ap.add_argument("--sharded-timeout", type=float, default=420.0,
                    help="seconds the sharded leg may take before the line is printed without it (a hung collective must not cost the run)")
    ap.add_argument("--sharded-nq", type=int, default=16384,
                    help="traversals per rank and step of the sharded leg (a step costs 86 us at 8192, 89 us at 16384: the more the better; 5.5 MB of state each)")
    ap.add_argument("--exchange", choices=["rccl", "host", "gloo"], default="rccl",
                    help="sharded leg: rccl (product: device buffers, one stream) or host (rehearsal of N ranks on one GPU: "
                         "host-staged buffers over the TCP group; `gloo` is an alias)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    return ap.parse_args()


# ------------------------------------------------------------------ launching N ranks
def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: N fresh child processes, started before this process
    makes any HIP call (it never does), rank 0's JSON line goes straight to our stdout."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RAD_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        for p in procs:
            p.wait()
            rc = rc or p.returncode
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if rc:
        print(f"bench.py: a rank exited with status {rc}", file=sys.stderr)
    return rc


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


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def pctl(xs, q):
    return float(np.percentile(np.asarray(xs, np.float64), q)) if len(xs) else None


# ------------------------------------------------------------------ one corpus: build, measure
def build_index(args, mode, device, layout=True):
    from rad_amd.device import DeviceIndex
    n, ndim, M = args.n, args.ndim, args.connectivity
    idx = DeviceIndex(ndim, M, 2 * M, args.expansion_add, device=device)
    idx.synth_vectors(n, seed=20260101, mode=mode)
    t_build = 0.0
    if args.graph == "synthetic" and mode == 1:
        idx.synth_graph(seed=777)
    else:
        X = np.empty((n, idx.row_bytes), np.uint8)    # add() takes host rows, as the reference's does
        for f in range(0, n, 4_000_000):
            c = min(4_000_000, n - f)
            X[f:f + c] = idx.read_vectors(f, c)
        idx.close()
        idx = DeviceIndex(ndim, M, 2 * M, args.expansion_add, device=device)
        t_build = time.perf_counter()
        for f in range(0, n, 5_000_000):
            idx.add_rows(X[f:f + 5_000_000], seed=777, max_batch=16384)
        t_build = time.perf_counter() - t_build
        del X
    info = None
    if layout and args.table == "group":
        info = idx.optimize_layout()
    return idx, t_build, info


def query_batches(idx, n_batches, nq, n, seed):
    qrng = np.random.default_rng(seed)
    return [idx.read_vectors(int(qrng.integers(0, n - nq)), nq) for _ in range(n_batches)]


def run_traversal_leg(args, idx, batches, steps, warmup, barrier):
    """`steps` timed steps of the single-GPU hot path on this rank's index.  Returns a dict of sums and the
    per-step kernel times."""
    from rad_amd.device import DeviceTraversal
    trav = DeviceTraversal(idx, batches[0], args.n_to_score)

    def step(b):
        trav.reset(batches[b])
        running = trav.run(0)
        assert running == 0
        ms, launches = trav.kernel_time()
        st = trav.stats()
        return ms, launches, st

    for w in range(warmup):
        step(w)
    barrier()
    t0 = time.perf_counter()
    k_ms, k_launches, pops, evals, nbrs = [], 0, 0, 0, 0
    last = None
    for s in range(steps):
        ms, launches, st = step(warmup + s)
        k_ms.append(ms / max(launches, 1))
        k_launches += launches
        pops += int(st.n_pops.sum()); evals += int(st.n_scored.sum()); nbrs += int(st.n_nbr.sum())
        last = st
    barrier()
    elapsed = time.perf_counter() - t0
    out = {"elapsed": elapsed, "pops": pops, "evals": evals, "nbrs": nbrs, "k_ms": k_ms, "launches": k_launches,
           "kernel": trav.kernel, "table": trav.table, "state_bytes": trav.state_bytes(), "last_stats": last,
           "remids": float(last.n_remid.mean()), "repivots": float(last.n_repivot.mean()), "flushes": float(last.n_flush.mean())}
    trav.close()
    return out


def roofline_of(leg, B):
    alg = (leg["evals"] * (B + 4) + leg["pops"] * 4) / max(leg["launches"], 1)
    avg_ms = float(np.mean(leg["k_ms"]))
    ach = alg / (avg_ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": leg["kernel"], "table": leg["table"], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms,
            "launch_ms_median": pctl(leg["k_ms"], 50), "launch_ms_p10": pctl(leg["k_ms"], 10), "launch_ms_p90": pctl(leg["k_ms"], 90),
            "launches": leg["launches"]}


def recall_of(idx, Q, k=10, ef=128):
    """recall@k of the graph search against the exact top-k (both on the GPU): is the built graph a graph?"""
    import ctypes as C
    from rad_amd import _lib
    from rad_amd._lib import check, ptr
    nq = Q.shape[0]
    s = np.full((nq, k), 0xFFFFFFFF, np.uint32); a = np.zeros((nq, k), np.uint32); o = np.zeros((nq, k), np.uint32)
    cnt = np.zeros(nq, np.uint32)
    check(_lib.lib().radhip_search(idx._h, ptr(Q), nq, k, ef, ptr(s), ptr(a), ptr(o), ptr(cnt), None, None))
    es, _ea, _eo, _ec = idx.topk(Q, k)
    return float(np.mean([len(set(s[i]) & set(es[i])) / k for i in range(nq)]))


def cpu_baseline(idx, queries, gpu_stats, args):
    """The oracle (C restatement of the reference control flow with a usearch-shaped index; pthreads over
    independent traversals), rebuilt -O3 -march=native for this host and timed on its cores on a bounded
    sample of the same workload: same corpus + graph (copied back from HBM), same n_to_score, fewer
    traversals.  The traversals it runs are ALSO the parity sample: their (scored, expansions, neighbours)
    counters must equal the GPU's for the same queries."""
    from oracle import rad_oracle as O
    O.use_library(O.build_native())
    cores = host_cores()
    info = idx.info()
    n = info.n
    X = np.empty((n, idx.row_bytes), np.uint8)
    for f in range(0, n, 4_000_000):
        c = min(4_000_000, n - f)
        X[f:f + c] = idx.read_vectors(f, c)
    levels, adj0, upper_row, adjU = idx.read_graph()
    g = O.Graph(int(n), int(info.connectivity_base), int(info.connectivity), int(info.max_level),
                int(info.entry), levels, adj0, upper_row, adjU)
    nt = min(32 * cores, queries.shape[0])
    done, wall, pops, evals = 0, 0.0, 0, 0
    ok = 0
    while done < queries.shape[0]:
        q = queries[done:done + nt]
        t0 = time.perf_counter()
        n_scored, n_pops, n_nbr = O.rad_traverse_many(g, X, q, args.n_to_score, cores)
        wall += time.perf_counter() - t0
        sl = slice(done, done + q.shape[0])
        ok += int(((n_scored == gpu_stats.n_scored[sl]) & (n_pops == gpu_stats.n_pops[sl]) & (n_nbr == gpu_stats.n_nbr[sl])).sum())
        pops += int(n_pops.sum()); evals += int(n_scored.sum())
        done += q.shape[0]
        if wall >= args.cpu_seconds:
            break
    n1 = min(8, queries.shape[0])
    t0 = time.perf_counter()
    s1, p1, _ = O.rad_traverse_many(g, X, queries[:n1], args.n_to_score, 1)
    w1 = time.perf_counter() - t0
    return {"value": pops / wall, "unit": "expansions/s", "cores": cores, "kind": "port",
            "evals_per_s": evals / wall, "one_thread_value": float(p1.sum()) / w1, "cpu_model": cpu_model(),
            "sample": f"{done} of the {queries.shape[0]} traversals of the last timed step (same corpus, graph, n_to_score), "
                      f"{wall:.1f} s wall on {cores} threads + {n1} traversals on 1 thread ({w1:.1f} s); usearch-shaped C "
                      f"restatement (oracle/, -O3 -march=native, software prefetch), not usearch"}, f"{ok}/{done}", ok == done


# ------------------------------------------------------------------ the row-sharded leg
def run_sharded_leg(args, idx, grp, rank, world, local_rank, barrier):
    """BASELINE's partitioning.  Before the rows of the other ranks are dropped, the single-GPU kernel runs
    this rank's sharded queries on the whole corpus: the sharded run must reproduce its counters exactly."""
    from rad_amd.device import DeviceShard, DeviceTraversal, RcclComm
    from rad_amd.sharded import RowShardedTraversal
    n = args.n
    nq = args.sharded_nq
    n_batches = args.warmup + args.steps
    qrng = np.random.default_rng(99)
    firsts = [int(qrng.integers(0, n - world * nq)) for _ in range(n_batches)]
    Qall = [idx.read_vectors(f, world * nq) for f in firsts]          # rank-major, identical on every rank
    ref = DeviceTraversal(idx, Qall[-1][rank * nq:(rank + 1) * nq], args.n_to_score)
    ref.run(0)
    want = ref.stats()
    ref.close()
    rows = n // world
    first = rank * rows
    count = rows if rank < world - 1 else n - first
    idx.keep_rows(first, count)
    comm, note = None, ""
    use_host = args.exchange != "rccl"
    if not use_host:
        # RCCL: rank 0 ALWAYS broadcasts an (ok, id-or-error) pair, every rank reports its init, and the
        # whole group takes the same path; a hung init is killed by a watchdog instead of waiting forever
        try:
            box = (True, RcclComm.unique_id()) if rank == 0 else None
        except Exception as e:   # noqa: BLE001
            box = (False, f"{type(e).__name__}: {e}")
        ok, payload = grp.broadcast_obj(box)
        err = "" if ok else payload
        if ok:
            dog = threading.Timer(180.0, lambda: (print(f"bench.py rank {rank}: RCCL init hung", file=sys.stderr), os._exit(3)))
            dog.daemon = True
            dog.start()
            try:
                comm = RcclComm(rank, world, payload, local_rank)
            except Exception as e:   # noqa: BLE001
                err = f"{type(e).__name__}: {e}"
            dog.cancel()
        errs = [e for e in grp.allgather_obj(err) if e]
        if errs:
            if comm is not None:
                comm.close()
            comm, use_host = None, True
            note = " (RCCL unavailable: " + "; ".join(sorted(set(errs)))[:240] + ")"
    res = {"steps": 0, "bytes": 0, "pops": 0, "evals": 0}
    last = None

    sh = DeviceShard(idx, rank, world, first, count, Qall[0], args.n_to_score)

    def one(b):
        nonlocal last
        sh.reset(Qall[b])
        if use_host:
            drv = RowShardedTraversal(sh, grp.allgather_u32, grp.reduce_scatter_sum_u32, rank, world)
            steps, xb = drv.run(), 0
            xb = drv.exchanged_bytes
        else:
            steps = sh.run(comm)
            xb = sh.timing()[3]
        st = sh.stats()
        last = st
        return steps, xb, int(st.n_pops.sum()), int(st.n_scored.sum())

    for w in range(args.warmup):
        one(w)
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        steps, xb, p, e = one(args.warmup + s)
        res["steps"] += steps; res["bytes"] += xb; res["pops"] += p; res["evals"] += e
    barrier()
    res["elapsed"] = time.perf_counter() - t0
    good = int(((last.n_scored == want.n_scored) & (last.n_pops == want.n_pops) & (last.n_nbr == want.n_nbr)).sum())
    res["parity_ok"], res["parity_n"] = good, nq
    sh.close()
    res["exchange"] = ("host-staged buffers over the TCP group" if use_host else "RCCL ncclAllGather + ncclReduceScatter on device buffers") + note
    if comm is not None:
        comm.close()
    return res


# ------------------------------------------------------------------ main (one rank)
def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from rad_amd import _lib
    from rad_amd.rendezvous import TcpGroup
    _lib.lib()
    if _lib.device_count() <= local_rank:
        raise SystemExit("bench.py needs an MI355X per rank (there is no CPU fallback); --single-device rehearses N ranks on GPU 0")
    port = int(os.environ.get("MASTER_PORT", "29500")) + (0 if os.environ.get("RAD_BENCH_SPAWNED") else 1)
    grp = TcpGroup(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), port)

    def barrier():
        grp.barrier()

    if args.table != "auto":
        os.environ["RADHIP_TABLE"] = args.table
    n = args.n
    idx, t_build, lay = build_index(args, args.corpus_mode, local_rank)
    info = idx.info()
    B = info.row_stride
    if args.nq <= 0:
        # two resident rounds: traversals end at different times and the second round's workgroups take over
        # the slots the early finishers leave
        args.nq = 2 * idx.traversal_capacity()
    n_batches = args.warmup + args.steps
    # replicas / single GPU: every rank runs its OWN query batches (the queries are what is split)
    batches = query_batches(idx, n_batches, args.nq, n, 4242 + rank)
    leg = run_traversal_leg(args, idx, batches, args.steps, args.warmup, barrier)
    recall = recall_of(idx, batches[-1][:128]) if (rank == 0 and args.graph == "built") else None

    sums = grp.allreduce([leg["pops"], leg["evals"]], "sum")
    elapsed_max = float(grp.allreduce([leg["elapsed"]], "max")[0])
    value_replicas = float(sums[0]) / elapsed_max

    out = {
        "metric": METRIC, "value": value_replicas, "unit": "expansions/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64 popcount (integer)", "data": "synthetic",
    }
    graph_desc = ("closed-form synthetic graph" if (args.graph == "synthetic" and args.corpus_mode == 1) else
                  f"HNSW graph built on the GPU (expansion_add={args.expansion_add}, {t_build:.0f} s)")
    corpus_desc = {0: "dense random corpus", 1: "two-level clustered sparse corpus (round 1)",
                   2: "hierarchical sparse corpus (neighbourhood structure at every scale)"}[args.corpus_mode]
    config = {
        "workload": f"{n // 1_000_000}M x {args.ndim}-bit fingerprints resident in HBM, connectivity={args.connectivity} "
                    f"(level-0 width {2 * args.connectivity}), {args.nq} concurrent best-first RAD traversals per GPU to "
                    f"n_to_score={args.n_to_score}, synthetic {corpus_desc}, {graph_desc}",
        "rows": n, "ndim": args.ndim, "connectivity": args.connectivity, "nq_per_gpu": args.nq, "n_to_score": args.n_to_score,
        "corpus_mode": args.corpus_mode, "graph_recall_at_10_ef128": recall,
        "layout": None if lay is None else {"seconds": lay.seconds, "groups_per_row": lay.groups_per_row, "degree": lay.degree},
        "parallelism": "single GPU",
    }
    out["config"] = config
    out["evals_per_s"] = float(sums[1]) / elapsed_max
    out["evals_per_expansion"] = float(sums[1]) / max(float(sums[0]), 1.0)
    out["queue_per_traversal"] = {"repivots": leg["repivots"], "remids": leg["remids"], "flushes": leg["flushes"]}
    out["roofline"] = roofline_of(leg, B)
    prof = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(prof):
        try:
            with open(prof) as f:
                pj = json.load(f)
            if (pj.get("n") == n and pj.get("nq") == args.nq and pj.get("n_to_score") == args.n_to_score
                    and pj.get("corpus_mode", 1) == args.corpus_mode and pj.get("table") == leg["table"]):
                tr = pj.get("hbm_bytes_per_launch")
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_source"] = "profiles/traffic_latest.json (rocprofv3 --pmc, separate passes, measured offline)"
                if tr:
                    out["roofline"]["hbm_real_gbs"] = tr / (out["roofline"]["avg_launch_ms"] * 1e-3) / 1e9
        except Exception:
            pass

    if world > 1:
        box = {}

        def _leg():
            try:
                box["res"] = run_sharded_leg(args, idx, grp, rank, world, local_rank, barrier)
            except BaseException as e:   # noqa: BLE001 - reported in the line, never silent
                box["err"] = f"{type(e).__name__}: {e}"
        th = threading.Thread(target=_leg, daemon=True)
        th.start()
        th.join(args.sharded_timeout)
        if th.is_alive() or "err" in box:
            # the replicas leg is measured: print the line with what went wrong in the sharded leg and leave
            # (no further collective: the other ranks are in the same state or gone)
            why = box.get("err", f"no result after {args.sharded_timeout:.0f} s (hung collective?)")
            if rank == 0:
                if args.mode == "sharded":
                    print(f"bench.py: the sharded leg failed ({why}) and --mode sharded asked for its value", file=sys.stderr)
                    os._exit(4)
                config["parallelism"] = "replicas (--mode replicas): every GPU holds the whole corpus and graph, the query batch is split, no collective"
                out["sharded"] = {"error": why}
                out["replicas"] = {"value": value_replicas, "unit": "expansions/s"}
                print(json.dumps(out), flush=True)
            os._exit(0 if args.mode == "replicas" else 4)
        sh = box["res"]
        tot = grp.allreduce([sh["pops"], sh["evals"], sh["parity_ok"], sh["parity_n"]], "sum")
        el = float(grp.allreduce([sh["elapsed"]], "max")[0])
        sharded = {
            "value": float(tot[0]) / el, "unit": "expansions/s", "ms_per_step": el / args.steps * 1e3,
            "evals_per_s": float(tot[1]) / el, "traversals_per_gpu_per_step": args.sharded_nq,
            "frontier_steps_per_step": sh["steps"] / max(args.steps, 1),
            "exchanged_bytes_per_rank_per_step": sh["bytes"] / max(args.steps, 1),
            "parity_vs_single_gpu": f"{int(tot[2])}/{int(tot[3])}", "exchange": sh["exchange"],
            "partitioning": f"one HNSW graph over all {n} rows (adjacency replicated), rows sharded by contiguous slot range "
                            f"({n // world} per GPU), traversals partitioned over the ranks; per frontier step an all-gather of "
                            f"the candidate slots and a reduce-scatter of their (and, or) scores; strict best-first, results "
                            f"bit-identical to one GPU",
        }
        replicas = {"value": value_replicas, "unit": "expansions/s", "ms_per_step": out["ms_per_step"],
                    "partitioning": "every GPU holds the whole corpus and graph, the query batch is split, no collective"}
        if int(tot[2]) != int(tot[3]):
            raise SystemExit(f"bench.py: the sharded traversals differ from the single-GPU ones ({sharded['parity_vs_single_gpu']}): no value printed")
        if args.mode == "sharded":
            out["value"], out["ms_per_step"] = sharded["value"], sharded["ms_per_step"]
            out["evals_per_s"] = sharded["evals_per_s"]
            config["parallelism"] = "row-sharded (--mode sharded): " + sharded["partitioning"] + "; exchange = " + sharded["exchange"]
            config["nq_per_gpu"] = args.sharded_nq
            out["roofline"]["note"] = ("roofline of the single-GPU traversal kernel on this rank (replicas leg); the sharded step is "
                                       "bound by its two collectives per frontier step, not by HBM")
        else:
            config["parallelism"] = "replicas (--mode replicas): " + replicas["partitioning"]
        out["sharded"], out["replicas"] = sharded, replicas

    if rank != 0:
        grp.barrier()
        grp.close()
        return

    if world == 1:
        if not args.no_cpu_baseline:
            cb, sample, ok = cpu_baseline(idx, batches[-1], leg["last_stats"], args)
            out["cpu_baseline"] = cb
            out["parity_sample"] = sample
            if not ok:
                raise SystemExit(f"bench.py: GPU and oracle disagree on the parity sample ({sample}): no value printed")
        if args.corpus_mode != 1 and not args.no_reference_corpus and args.graph == "built":
            # round 1's corpus on the same kernel, for continuity (its built graph is not a usable HNSW graph at this
            # size: recall is reported beside the numbers)
            idx.close()
            idx1, tb1, _ = build_index(args, 1, local_rank, layout=False)
            b1 = query_batches(idx1, 1 + min(args.steps, 3), args.nq, n, 4242)
            leg1 = run_traversal_leg(args, idx1, b1, min(args.steps, 3), 1, lambda: None)
            rf1 = roofline_of(leg1, B)
            out["reference_corpus_r01"] = {
                "value": leg1["pops"] / leg1["elapsed"], "unit": "expansions/s", "evals_per_s": leg1["evals"] / leg1["elapsed"],
                "evals_per_expansion": leg1["evals"] / max(leg1["pops"], 1), "roofline_frac": rf1["frac"],
                "roofline_achieved_gbs": rf1["achieved"], "avg_launch_ms": rf1["avg_launch_ms"], "table": leg1["table"],
                "graph_recall_at_10_ef128": recall_of(idx1, b1[-1][:128]), "build_s": tb1, "steps": min(args.steps, 3),
                "workload": "round 1's bench workload: two-level clustered sparse corpus, HNSW graph built on the GPU"}
            idx1.close()
    print(json.dumps(out), flush=True)
    grp.barrier()
    grp.close()


if __name__ == "__main__":
    main()
