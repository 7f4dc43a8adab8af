#!/usr/bin/env python3
"""bench.py — RAD HNSW neighbor-expansion throughput on MI355X.  No torch anywhere.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on; it fits one GPU):
100M x 1024-bit fingerprints resident in HBM; an HNSW graph built over them on the GPU by the library's
own insert kernels (connectivity 8, level-0 width 16, expansion_add 64: ~30 s); `nq` independent
best-first RAD traversals (Tanimoto-scored; nq defaults to four times what the device holds resident at
once: 4 x 16384 on MI355X, 217 GB of traversal state beside the 20 GB index), each run to n_to_score = 100k.  One "step" = one pass of the hot path over
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
RANDOM_LINE_PEAK_G = 38.2   # G random 128-B line reads per second at a 64 GiB footprint, measured (profiles/r03/latency_footprint.log)
# request rates of the memory side by shape, 32 GiB footprint (scripts/probe_request_size.hip, profiles/r04/probe_request_size.md)
READ_LINES_G = 44.0         # random 128-B line reads: 38.4 (16 B per lane, 64 lines per instruction) .. 48.5 (<= 32 lines per instruction)
PARTIAL_WRITES_G = 21.9     # random writes of 4 .. 32 B (one 32-B request each) into lines that are NOT in the L2
FULL_WRITES_G = 49.5        # random full 64-B writes; a 4-B store into a line a read has just fetched costs the same (mix_store4_probed_line)
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
                    help="concurrent traversals per GPU per step (0 = four times what the device holds resident at once, memory permitting)")
    ap.add_argument("--n-to-score", type=int, default=100_000)
    ap.add_argument("--corpus-mode", type=int, default=2,
                    help="0 dense Bernoulli(0.5), 1 two-level clustered sparse (round 1), 2 hierarchical sparse")
    ap.add_argument("--graph", choices=["built", "synthetic"], default="built",
                    help="adjacency: an HNSW graph built on the GPU by Index.add (default) or the closed-form "
                         "generator (corpus mode 1 only; set up in 20 ms)")
    ap.add_argument("--expansion-add", type=int, default=400,
                    help="expansion_add of the built graph: 400 = the reference README's value (README.md:52), the headline graph since round 4 "
                         "(162 s for 100M rows); rounds 1-3 benched a graph built with 64 (20 s), kept as the labelled secondary leg")
    ap.add_argument("--secondary-expansion-add", type=int, default=64,
                    help="N = 1: a second, shorter leg on a graph built with this expansion_add (0 = skip)")
    ap.add_argument("--graph-cache", default="", help="profiling sessions: .npz the built graph is saved to / loaded from, so that every "
                                                       "rocprofv3 pass does not build it again (the graph is the same; only setup time changes)")
    ap.add_argument("--build-batch", type=int, default=16384,
                    help="inserts per batch of the graph build (a batch searches the pre-batch graph; at most 1/16 of the graph so far, at most 65536)")
    ap.add_argument("--chain", type=int, default=20,
                    help="steps (batches) chained into ONE launch of the traversal kernel: a launch ends with its longest traversals running "
                         "alone (~150 ms whatever its size), so the tail is paid once per chain (0 = the two-object pipeline of overlapped launches)")
    ap.add_argument("--no-overlap", action="store_true", help="one traversal object, one launch after the other (A/B against the two-stream pipeline)")
    ap.add_argument("--no-kernel-legs", action="store_true", help="skip the K1 scan / K2 gather / top-k measurements of the N = 1 line")
    ap.add_argument("--no-config-legs", action="store_true", help="skip the BASELINE configs[1] / configs[4] legs of the N = 1 line")
    ap.add_argument("--config-budget-s", type=float, default=150.0, help="wall-clock budget of the configs legs together")
    ap.add_argument("--table", choices=["auto", "hash", "group", "local"], default="local",
                    help="visited/scored table of the traversal kernel: local (default since round 4) = the bucket table with a node's home "
                         "bucket taken from its graph-locality layout id (+7 % at 100M rows: profiles/r04), auto = the library's own choice "
                         "(the slot-hashed bucket table), group = the grouped table (2 bits per node; fewer requests, more instructions: a draw), "
                         "hash = one 8-byte entry per probe")
    ap.add_argument("--no-reference-corpus", action="store_true", help="skip the round-1 corpus leg (N = 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU work the cpu_baseline sample should take at least")
    ap.add_argument("--mode", choices=["sharded", "replicas", "peer"], default="replicas",
                    help="which N > 1 leg `value` reports: replicas (every GPU holds the corpus), sharded (rows sharded, RCCL all-gather per frontier "
                         "step), peer (rows sharded, every rank maps the peers' shards into one address range and runs the single-GPU kernel "
                         "over xGMI reads: no collective); in replicas mode the other two run as side legs")
    ap.add_argument("--no-peer-leg", action="store_true", help="skip the peer-mapped side leg of an N > 1 replicas run")
    ap.add_argument("--sharded-timeout", type=float, default=300.0,
                    help="seconds the sharded leg may take before the line is printed without it (a hung collective must not cost the run)")
    ap.add_argument("--sharded-nq", type=int, default=65536,
                    help="traversals per rank and batch of the sharded leg (0.8 MB each for the scored list; the slots hold the rest)")
    ap.add_argument("--sharded-slots", type=int, default=32768,
                    help="slots per rank of the sharded leg (queue + sets of a traversal: 3.6 MB); a slot whose traversal is done takes "
                         "the next one of the batch (0 = one slot per traversal)")
    ap.add_argument("--sharded-groups", type=int, default=1, choices=[1, 2],
                    help="groups the sharded traversals of a rank are split into: 2 = two streams and two communicators, one group's "
                         "step kernel overlaps the other's collectives (RCCL exchange only)")
    ap.add_argument("--sharded-reference", choices=["auto", "rank0", "none"], default="auto",
                    help="--mode sharded: parity reference = the single-GPU kernel on rank 0, which holds the whole corpus for one "
                         "sequential phase before it creates its shard (auto: when corpus + graph fit one GPU)")
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


_T0 = time.perf_counter()


def note(msg):
    """progress on stderr (the JSON line on stdout stays alone): which stage a long or failed run was in"""
    print(f"[bench {time.perf_counter() - _T0:7.1f} s] {msg}", file=sys.stderr, flush=True)


def pctl(xs, q):
    return float(np.percentile(np.asarray(xs, np.float64), q)) if len(xs) else None


# ------------------------------------------------------------------ one corpus: build, measure
def build_index(args, mode, device, layout=True):
    from rad_amd.device import DeviceIndex
    n, ndim, M = args.n, args.ndim, args.connectivity
    idx = DeviceIndex(ndim, M, 2 * M, args.expansion_add, device=device)
    idx.synth_vectors(n, seed=20260101, mode=mode)
    note(f"corpus mode {mode}: {n} rows generated on the device")
    t_build = 0.0
    if args.graph == "synthetic" and mode == 1:
        idx.synth_graph(seed=777)
    else:
        cache = getattr(args, "graph_cache", "")
        cache = f"{cache}.n{n}.m{M}.ef{args.expansion_add}.mode{mode}.b{args.build_batch}.npz" if cache else ""
        if cache and os.path.exists(cache):
            z = np.load(cache)
            idx.load_graph(z["levels"], z["adj0"], z["upper_row"], z["adjU"], int(z["max_level"]), int(z["entry"]))
            t_build = float(z["t_build"])
            note(f"graph loaded from {cache} (built in {t_build:.1f} s when it was made)")
        else:
            # the rows are resident already (generated on the device): they are linked where they are, no host copy
            # of the corpus (radhip_index_link_resident == radhip_index_add of the same rows, tests/test_gpu_sharded.py)
            t_build = time.perf_counter()
            idx.link_resident(seed=777, max_batch=args.build_batch)
            t_build = time.perf_counter() - t_build
            note(f"graph built in {t_build:.1f} s")
            if cache:
                levels, adj0, upper_row, adjU = idx.read_graph()
                inf = idx.info()
                np.savez(cache, levels=levels, adj0=adj0, upper_row=upper_row, adjU=adjU, max_level=int(inf.max_level), entry=int(inf.entry), t_build=t_build)
                note(f"graph saved to {cache}")
    info = None
    if layout and args.table in ("group", "local"):
        info = idx.optimize_layout()
    return idx, t_build, info


def query_batches(idx, n_batches, nq, n, seed):
    qrng = np.random.default_rng(seed)
    return [idx.read_vectors(int(qrng.integers(0, n - nq)), nq) for _ in range(n_batches)]


def union_ms(intervals):
    """length of the union of [start, end] intervals (ms)"""
    tot, cur_s, cur_e = 0.0, None, None
    for s_, e_ in sorted(intervals):
        if cur_e is None or s_ > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s_, e_
        else:
            cur_e = max(cur_e, e_)
    return tot + ((cur_e - cur_s) if cur_e is not None else 0.0)


def run_chained_leg(args, idx, batches, steps, warmup, barrier):
    """`steps` timed steps with up to --chain of them per LAUNCH: one traversal object whose batch is chain x nq traversals, state per
    resident row of the kernel, a ring of scored lists (radhip_traversal_create_ring).  Rows take the traversals of the whole
    chain from one counter, so a step's longest traversals finish beside the next step's instead of alone on the device.  Every
    traversal of every step is computed and counted; the scored lists that are still readable at the end are those of the last
    `ring` traversals (>= the last step: the parity sample reads them).  The warm-up steps are a launch of their own."""
    from rad_amd.device import DeviceTraversal
    nq = batches[0].shape[0]
    chain = max(1, min(args.chain, max(steps, warmup)))
    first = np.concatenate([batches[i % len(batches)] for i in range(chain)])
    obj, ring = None, 0
    from rad_amd._lib import RadHipError, E_NOMEM
    for ring in (2 * nq, nq):            # lists of the last two steps when there is room, else of the last one
        try:
            obj = DeviceTraversal(idx, first, args.n_to_score, list_ring=ring)
            break
        except RadHipError as e:
            if e.code != E_NOMEM or ring == nq:
                raise
    note(f"traversal state for chains of {chain} x {nq} traversals allocated ({obj.state_bytes() / 1e9:.1f} GB, kernel {obj.kernel}, table {obj.table}, "
         f"{obj.slots or obj.nq} rows' worth of tables, ring of {obj.list_ring or obj.nq} scored lists)")
    acc = {"pops": 0, "evals": 0, "nbrs": 0, "k_ms": [], "iv": []}

    def launch(first_b, n_b, timed):
        q = inputs[(first_b, n_b)]
        obj.reset(q)
        assert obj.run(0) == 0
        if timed:
            st = obj.stats()
            ms, launches = obj.kernel_time()
            acc["k_ms"].append(ms / max(launches, 1)); acc["iv"].append(obj.launch_interval())
            acc["pops"] += int(st.n_pops.sum()); acc["evals"] += int(st.n_scored.sum()); acc["nbrs"] += int(st.n_nbr.sum())
            return st
        return None

    # the query matrix of every launch, laid out before the clock starts (the step = upload + arm + traverse + read the counters back)
    inputs, b = {}, 0
    for lo_b, n_all in ((0, warmup), (warmup, steps)):
        b = lo_b
        while b < lo_b + n_all:
            c = min(chain, lo_b + n_all - b)
            inputs[(b, c)] = np.ascontiguousarray(np.concatenate(batches[b:b + c]) if c > 1 else batches[b])
            b += c
    b = 0
    while b < warmup:
        c = min(chain, warmup - b); launch(b, c, False); b += c
    note("warm-up done")
    barrier()
    t0 = time.perf_counter()
    last, last_n = None, 0
    while b < warmup + steps:
        c = min(chain, warmup + steps - b)
        last, last_n = launch(b, c, True), c
        b += c
    barrier()
    elapsed = time.perf_counter() - t0
    note(f"{steps} timed steps done in {len(acc['k_ms'])} launch(es) ({elapsed / max(steps, 1) * 1e3:.0f} ms per step)")
    lo = (last_n - 1) * nq                              # the last step's traversals within the last launch
    sl = slice(lo, lo + nq)
    last_step = type(last)(*[None if v is None else v[sl] for v in (last.n_scored, last.n_pops, last.n_nbr, last.status, last.n_repivot, last.n_flush, last.n_remid, last.n_upper)])
    out = {"elapsed": elapsed, "pops": acc["pops"], "evals": acc["evals"], "nbrs": acc["nbrs"], "k_ms": acc["k_ms"], "launches": len(acc["k_ms"]),
           "busy_ms": union_ms(acc["iv"]), "objects": 1, "slots": obj.slots, "chain": chain, "list_ring": obj.list_ring,
           "kernel": obj.kernel, "table": obj.table, "state_bytes": obj.state_bytes(), "last_stats": last_step,
           "last_hashes": obj.result_hashes(lo, nq),
           "remids": float(last_step.n_remid.mean()), "repivots": float(last_step.n_repivot.mean()), "flushes": float(last_step.n_flush.mean())}
    obj.close()
    return out


def run_traversal_leg(args, idx, batches, steps, warmup, barrier, overlap=True):
    """(--chain 0) `steps` timed steps of the single-GPU hot path on this rank's index.  One step = one batch: re-arm (query upload; the
    rows' epochs make table reuse free) + one launch of the traversal kernel to completion of the batch.  The traversal
    state lives per resident ROW of the kernel (RADHIP_TRAV_SLOTS) and two objects on two streams take the batches in turn,
    so that the next batch's wavefronts start while the last traversals of this one still run (a launch ends with its
    longest traversals running alone: ~130 ms whatever its size).  The pipeline is empty before the clock starts and is
    drained before it stops: exactly `steps` batches are started and finished inside the timed region."""
    from rad_amd.device import DeviceTraversal
    if getattr(args, "chain", 0) > 1 and overlap and steps > 1:
        return run_chained_leg(args, idx, batches, steps, warmup, barrier)
    n_obj = min(2, getattr(args, "objects", 2)) if (overlap and steps + warmup > 1) else 1
    objs = [DeviceTraversal(idx, batches[0], args.n_to_score, slots=True, own_stream=n_obj > 1) for _ in range(n_obj)]
    note(f"traversal state for {n_obj} x {objs[0].nq} traversals allocated ({sum(o.state_bytes() for o in objs) / 1e9:.1f} GB, kernel {objs[0].kernel}, "
         f"table {objs[0].table}, {objs[0].slots or objs[0].nq} rows' worth of tables)")
    acc = {"pops": 0, "evals": 0, "nbrs": 0, "k_ms": [], "iv": [], "last": None, "last_obj": None}

    def collect(o, timed):
        assert o.finish() == 0
        if not timed:
            return
        st = o.stats()
        ms, launches = o.kernel_time()
        acc["k_ms"].append(ms / max(launches, 1))
        acc["iv"].append(o.launch_interval())
        acc["pops"] += int(st.n_pops.sum()); acc["evals"] += int(st.n_scored.sum()); acc["nbrs"] += int(st.n_nbr.sum())
        acc["last"], acc["last_obj"] = st, o

    def pipeline(first, count, timed):
        flying = []
        for s_ in range(count):
            o = objs[s_ % n_obj]
            if o in flying:
                collect(o, timed); flying.remove(o)
            o.reset(batches[first + s_])
            o.start(); flying.append(o)
        for o in flying:
            collect(o, timed)

    pipeline(0, warmup, False)
    note("warm-up done")
    barrier()
    t0 = time.perf_counter()
    pipeline(warmup, steps, True)
    barrier()
    elapsed = time.perf_counter() - t0
    note(f"{steps} timed steps done ({elapsed / max(steps, 1) * 1e3:.0f} ms each)")
    last, lo = acc["last"], acc["last_obj"]
    out = {"elapsed": elapsed, "pops": acc["pops"], "evals": acc["evals"], "nbrs": acc["nbrs"], "k_ms": acc["k_ms"], "launches": len(acc["k_ms"]),
           "busy_ms": union_ms(acc["iv"]), "objects": n_obj, "slots": objs[0].slots,
           "kernel": lo.kernel, "table": lo.table, "state_bytes": sum(o.state_bytes() for o in objs), "last_stats": last,
           "last_hashes": lo.result_hashes(0, lo.nq),
           "remids": float(last.n_remid.mean()), "repivots": float(last.n_repivot.mean()), "flushes": float(last.n_flush.mean())}
    for o in objs:
        o.close()
    return out


def roofline_of(leg, B):
    """Algorithmic bytes (B + 4 per evaluation, 4 per expansion) over the time the device ran the traversal kernel.
    With two objects the launches of consecutive batches overlap (the next batch's wavefronts fill the device while the last
    traversals of this one finish), so `achieved` = ALL timed launches' bytes / the UNION of their HIP-event intervals —
    summed work over the time the kernel was on the device; `avg_launch_ms` is still the mean start-to-end duration of one launch
    (what rocprofv3 --kernel-trace --stats reports per kernel: it contains the time a launch shares the device with its neighbour)."""
    alg_total = leg["evals"] * (B + 4) + leg["pops"] * 4
    alg = alg_total / max(leg["launches"], 1)
    avg_ms = float(np.mean(leg["k_ms"]))
    busy = leg["busy_ms"] if leg.get("busy_ms") else avg_ms * leg["launches"]
    ach = alg_total / (busy * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": leg["kernel"], "table": leg["table"], "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_ms,
            "kernel_busy_ms": busy, "kernel_busy_ms_per_launch": busy / max(leg["launches"], 1),
            "accounting": "achieved = algorithmic bytes of all timed launches / union of their HIP-event intervals (= their sum when launches do not overlap: "
                          "--chain; launches of consecutive batches overlap on two streams with --chain 0)",
            "achieved_by_avg_launch_duration": alg / (avg_ms * 1e-3) / 1e9,
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


def kernel_legs(idx, n, B):
    """The other kernels of the path on the SAME resident corpus, each by HIP events on the library's stream (radhip_last_kernel_ms):
    K1 scan (query x all rows, N x B bytes per pass), K2 gather (random 128-B rows, B + 4 per pair) and the exact top-k scan
    (wall time: it is several kernels).  SURVEY.md §8d; VERDICT r03 #3(i)."""
    from rad_amd import _lib
    L = _lib.lib()
    out = {}
    q = idx.read_vectors(7, 8)
    chunk = 25_000_000          # (and, or) of every row come back to the host: bound the buffers
    for nq in (1, 8):
        ms = 0.0
        for f in range(0, n, chunk):
            c = min(chunk, n - f)
            idx.scan(q[:nq], f, c)
            ms += L.radhip_last_kernel_ms()
        # N x B per pass, the (and, or) outputs included when they are written (they are: 8 B per query and row)
        gbs = n * (B + 8 * nq) / (ms * 1e-3) / 1e9
        out[f"scan_{nq}q"] = {"ms": ms, "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS, "G_evals_per_s": n * nq / (ms * 1e-3) / 1e9,
                              "rows_only_GB/s": n * B / (ms * 1e-3) / 1e9,
                              "bytes": "rows read once per pass (N x B) + 8 B per (query, row) written (SURVEY.md §8d: outputs count when they are written)"}
    rng = np.random.default_rng(0)
    m = 20_000_000
    slots = rng.integers(0, n, m).astype(np.uint32)
    off = (np.arange(5, dtype=np.uint64) * (m // 4)).astype(np.uint64); off[-1] = m
    best = None
    for _ in range(2):
        idx.gather(q[:4], slots, off)
        ms = L.radhip_last_kernel_ms()
        best = ms if best is None else min(best, ms)
    gbs = m * (B + 4) / (best * 1e-3) / 1e9
    out["gather"] = {"ms": best, "pairs": m, "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS, "G_pairs_per_s": m / (best * 1e-3) / 1e9,
                     "bytes": "B + 4 per (query, slot) pair: a random 128-B row and its slot id"}
    Q = idx.read_vectors(1234, 8)
    idx.topk(Q, 10)
    t0 = time.perf_counter(); s_, _a, _o, _c = idx.topk(Q, 10); dt = time.perf_counter() - t0
    gbs = n * B / dt / 1e9
    out["topk_8q_k10"] = {"ms": dt * 1e3, "GB/s": gbs, "frac": gbs / HBM_PEAK_GBS, "G_evals_per_s": 8 * n / dt / 1e9,
                          "self_is_nearest": bool((s_[:, 0] == np.arange(1234, 1242)).all()),
                          "bytes": "rows read once for the 8 queries of a pass (N x B); wall time incl. query upload / result download"}
    return out


def config_legs(args, device):
    """BASELINE.json configs[1] and configs[4] on the driver's clock, behind a time budget: small enough to build
    their graphs here (GPU insert kernels, expansion_add as the config says), traversed by the kernel the library picks."""
    from rad_amd import _lib
    from rad_amd.device import DeviceIndex, DeviceTraversal
    t_start = time.perf_counter()
    out = {}

    def one(tag, n, ndim, M, ef, nq, nts, steps=2):
        if time.perf_counter() - t_start > args.config_budget_s:
            out[tag] = {"skipped": "time budget"}
            return
        idx = DeviceIndex(ndim, M, 2 * M, ef, device=device)
        idx.synth_vectors(n, seed=20260101, mode=2)
        t0 = time.perf_counter(); idx.link_resident(seed=777, max_batch=16384); tb = time.perf_counter() - t0
        Bc = idx.info().row_stride
        rng = np.random.default_rng(7)
        bs = [idx.read_vectors(int(rng.integers(0, n - nq)), nq) for _ in range(steps + 1)]
        a = argparse.Namespace(**vars(args)); a.n_to_score = nts
        leg = run_traversal_leg(a, idx, bs, steps, 1, lambda: None, overlap=not args.no_overlap)
        rf = roofline_of(leg, Bc)
        q8 = idx.read_vectors(3, 8)
        idx.scan(q8, 0, min(n, 8_000_000)); idx.scan(q8, 0, min(n, 8_000_000))
        sms = _lib.lib().radhip_last_kernel_ms()
        out[tag] = {"rows": n, "ndim": ndim, "connectivity": M, "expansion_add": ef, "graph_build_s": tb, "traversals_per_step": nq, "n_to_score": nts,
                    "steps": steps, "value": leg["pops"] / leg["elapsed"], "unit": "expansions/s", "evals_per_expansion": leg["evals"] / max(leg["pops"], 1),
                    "kernel": leg["kernel"], "table": leg["table"], "roofline_frac": rf["frac"], "roofline_achieved_gbs": rf["achieved"],
                    "avg_launch_ms": rf["avg_launch_ms"], "graph_recall_at_10_ef128": recall_of(idx, bs[-1][:64]),
                    "scan_8q_GB/s": min(n, 8_000_000) * Bc / (sms * 1e-3) / 1e9}
        idx.close()
        note(f"config leg {tag}: {out[tag]['value'] / 1e9:.3f} G expansions/s, frac {rf['frac']:.3f}")

    # configs[1]: 1M 1024-bit fingerprints, connectivity 8, single MI355X (Tanimoto kernels; traversals to 1 % of the corpus like the
    # two legs below — RAD's regime, index.html:628; to 100k = 10 % of this corpus most neighbours of a pop are scored already:
    # 2.2 evaluations per expansion, frac 0.064 in the round's earlier lines)
    one("c1_1M_1024bit_m8", 1_000_000, 1024, 8, 400, 65536, 10_000)     # (expansion_add as the reference README builds: README.md:52)
    # configs[4]: 2048-bit fingerprints, connectivity 32, expansion_add 400 (2M rows: what builds inside the budget)
    one("c4_2M_2048bit_m32_ef400", 2_000_000, 2048, 32, 400, 16384, 20_000)
    # the reference notebook's shape (examples/DUDEZ_example.ipynb:165-166): connectivity 16, expansion_add 400
    one("notebook_shape_2M_1024bit_m16_ef400", 2_000_000, 1024, 16, 400, 32768, 20_000)
    return out


def cpu_baseline(idx, queries, gpu_stats, args, gpu_hashes=None):
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
        n_scored, n_pops, n_nbr, hs = O.rad_traverse_many(g, X, q, args.n_to_score, cores, hashes=True)
        wall += time.perf_counter() - t0
        sl = slice(done, done + q.shape[0])
        same = (n_scored == gpu_stats.n_scored[sl]) & (n_pops == gpu_stats.n_pops[sl]) & (n_nbr == gpu_stats.n_nbr[sl])
        if gpu_hashes is not None:     # the whole scored list (slots, counts, order), as one 64-bit hash per traversal
            same &= hs == gpu_hashes[sl]
        ok += int(same.sum())
        pops += int(n_pops.sum()); evals += int(n_scored.sum())
        done += q.shape[0]
        if wall >= args.cpu_seconds:
            break
    n1 = min(8, queries.shape[0])
    t0 = time.perf_counter()
    s1, p1, _ = O.rad_traverse_many(g, X, queries[:n1], args.n_to_score, 1)[:3]
    w1 = time.perf_counter() - t0
    return {"value": pops / wall, "unit": "expansions/s", "cores": cores, "kind": "port",
            "evals_per_s": evals / wall, "one_thread_value": float(p1.sum()) / w1, "cpu_model": cpu_model(),
            "sample": f"{done} of the {queries.shape[0]} traversals of the last timed step (same corpus, graph, n_to_score), "
                      f"{wall:.1f} s wall on {cores} threads + {n1} traversals on 1 thread ({w1:.1f} s); usearch-shaped C "
                      f"restatement (oracle/, -O3 -march=native, software prefetch), not usearch"}, f"{ok}/{done}", ok == done


# ------------------------------------------------------------------ the row-sharded leg
def make_comms(args, grp, rank, world, local_rank, n_comms):
    """`n_comms` RCCL communicators over all ranks (one per group of traversals).  Rank 0 ALWAYS broadcasts an
    (ok, ids-or-error) pair, every rank reports its init, and the whole group takes the same path; a hung init is
    killed by a watchdog instead of waiting forever.  Returns (comms or None, note, rccl-info dict or None)."""
    from rad_amd.device import RcclComm
    if args.exchange != "rccl":
        return None, "", None
    try:
        box = (True, [RcclComm.unique_id() for _ in range(n_comms)]) if rank == 0 else None
    except Exception as e:   # noqa: BLE001
        box = (False, f"{type(e).__name__}: {e}")
    ok, payload = grp.broadcast_obj(box)
    err, comms = ("" if ok else payload), []
    if ok:
        dog = threading.Timer(180.0, lambda: (print(f"bench.py rank {rank}: RCCL init hung", file=sys.stderr), os._exit(3)))
        dog.daemon = True
        dog.start()
        try:
            for uid in payload:
                comms.append(RcclComm(rank, world, uid, local_rank))
        except Exception as e:   # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        dog.cancel()
    errs = [e for e in grp.allgather_obj(err) if e]
    if errs:
        for c in comms:
            c.close()
        return None, " (RCCL unavailable: " + "; ".join(sorted(set(errs)))[:240] + ")", None
    # what RCCL itself says the communicator spans: the N > 1 line carries it
    infos = grp.allgather_obj(comms[0].info())
    rccl = {"version": infos[0]["rccl_version"], "comm_count": infos[0]["comm_count"],
            "comm_counts_agree": len({i["comm_count"] for i in infos}) == 1,
            "comm_ranks": [i["comm_rank"] for i in infos], "rank_devices": [i["pci_bus_id"] for i in infos],
            "distinct_devices": len({i["pci_bus_id"] for i in infos}), "communicators_per_rank": n_comms}
    return comms, "", rccl


def bcast_array(grp, a, piece=1 << 27):
    """rank 0's array on every rank over the TCP group, in pieces (host-staged rehearsals only: the product path
    broadcasts the graph device to device, radhip_index_broadcast_graph)."""
    meta = grp.broadcast_obj((a.shape, a.dtype.str) if grp.rank == 0 else None)
    flat = np.ascontiguousarray(a).reshape(-1).view(np.uint8) if grp.rank == 0 else np.empty(int(np.prod(meta[0])) * np.dtype(meta[1]).itemsize, np.uint8)
    for f in range(0, flat.shape[0], piece):
        blob = grp.broadcast_obj(flat[f:f + piece].tobytes() if grp.rank == 0 else None)
        if grp.rank != 0:
            flat[f:f + piece] = np.frombuffer(blob, np.uint8)
    return flat.view(np.dtype(meta[1])).reshape(meta[0])


def synth_queries(args, device, firsts, count, mode):
    """query rows [f, f + count) of the closed-form corpus, from a scratch index that holds nothing else"""
    from rad_amd.device import DeviceIndex
    mini = DeviceIndex(args.ndim, args.connectivity, 2 * args.connectivity, args.expansion_add, device=device)
    out = []
    for f in firsts:
        mini.synth_vectors(count, seed=20260101, mode=mode, first_row=f, n_total=args.n)
        out.append(mini.read_vectors(0, count))
    mini.close()
    return out


def reference_sample(args, full, Qall_last, world, nq, ns, nfull):
    """The single-GPU kernel on an index that holds the WHOLE corpus, for the LAST `ns` traversals of every rank's
    last batch (with fewer slots than traversals those are the ones that slots took over from finished ones): counters of
    all of them, full scored lists of the first `nfull` of them per rank."""
    from rad_amd.device import DeviceTraversal
    Q = np.concatenate([Qall_last[(r + 1) * nq - ns:(r + 1) * nq] for r in range(world)])
    ref = DeviceTraversal(full, Q, args.n_to_score)
    ref.run(0)
    st = ref.stats()
    lists = {(r, i): ref.results(r * ns + i) for r in range(world) for i in range(nfull)}
    ref.close()
    return {"scored": st.n_scored.reshape(world, ns), "pops": st.n_pops.reshape(world, ns), "nbr": st.n_nbr.reshape(world, ns), "lists": lists}


def drive_shards(args, grp, rank, world, idx, first, count, Qall, comms, barrier, ref, ns, nfull):
    """`steps` timed batches of the row-sharded traversal on an index that holds rows [first, first + count) and the
    whole graph.  One group of traversals, or two on two streams (--sharded-groups 2, RCCL only)."""
    from rad_amd.device import DeviceShard
    from rad_amd.sharded import RowShardedTraversal
    nq = args.sharded_nq
    use_host = comms is None
    G = 1 if use_host else len(comms)
    nqg = nq // G

    def group_queries(Q, g):
        return np.concatenate([Q[r * nq + g * nqg:r * nq + (g + 1) * nqg] for r in range(world)])

    # fewer slots than traversals (the product loop only: the host-staged exchange does not carry the slots' traversal numbers)
    if not use_host and 0 < args.sharded_slots // G < nqg:
        os.environ["RADHIP_SHARD_SLOTS"] = str(args.sharded_slots // G)
    else:
        os.environ.pop("RADHIP_SHARD_SLOTS", None)
    shards = [DeviceShard(idx, rank, world, first, count, group_queries(Qall[0], g), args.n_to_score, own_stream=g > 0) for g in range(G)]
    res = {"steps": 0, "bytes": 0, "pops": 0, "evals": 0, "spec_asked": 0, "spec_used": 0, "spec_hits": 0}
    last = None

    def one(b):
        nonlocal last
        for g, sh in enumerate(shards):
            sh.reset(group_queries(Qall[b], g))
        if use_host:
            drv = RowShardedTraversal(shards[0], grp.allgather_u32, grp.reduce_scatter_sum_u32, rank, world)
            steps = drv.run()
            xb = drv.exchanged_bytes
        else:
            steps = shards[0].run(comms[0]) if G == 1 else shards[0].run_pair(comms[0], shards[1], comms[1])
            xb = sum(sh.timing()[3] for sh in shards)
        sts = [sh.stats() for sh in shards]
        last = sts[-1]
        sp = [sh.speculation() for sh in shards]
        return (steps, xb, sum(int(st.n_pops.sum()) for st in sts), sum(int(st.n_scored.sum()) for st in sts),
                sum(x[1] for x in sp), sum(x[2] for x in sp), sum(x[3] for x in sp))

    n_warm, n_steps = args.sh_warmup, args.sh_steps
    for w in range(n_warm):
        one(w)
    barrier()
    t0 = time.perf_counter()
    for s in range(n_steps):
        steps, xb, p, e, sa, su, shh = one(n_warm + s)
        res["steps"] += steps; res["bytes"] += xb; res["pops"] += p; res["evals"] += e
        res["spec_asked"] += sa; res["spec_used"] += su; res["spec_hits"] += shh
    barrier()
    res["elapsed"] = time.perf_counter() - t0
    res["spec_depth"] = shards[0].speculation()[0]
    res["width"] = shards[0].width
    res["engine"] = shards[0].engine
    res["slots"] = sum(sh.slots for sh in shards)
    res["groups"] = G
    res["state_bytes"] = sum(sh.state_bytes() for sh in shards)
    # parity gate: the last ns traversals of this rank (they ride in the last group) against the single-GPU kernel on the
    # whole corpus — all three counters, and the complete scored lists (slots and both counts) of the first nfull of them
    good = tot = 0
    if ref is not None:
        assert ns <= nqg
        k, o = ns, nqg - ns
        good = int(((last.n_scored[o:] == ref["scored"][rank]) & (last.n_pops[o:] == ref["pops"][rank]) & (last.n_nbr[o:] == ref["nbr"][rank])).sum())
        tot = k
        for i in range(min(nfull, k)):
            got, want = shards[-1].results(o + i), ref["lists"][(rank, i)]
            tot += 1
            good += int(all(np.array_equal(x, y) for x, y in zip(got, want)))
    res["parity_ok"], res["parity_n"] = good, tot
    for sh in shards:
        sh.close()
    return res


def peer_index(args, grp, rank, world, local_rank, graph_from=None):
    """This rank's view of a corpus whose ROWS are sharded over the ranks: its own shard generated on the device, the peers' shards
    mapped behind it (dmabuf descriptors over a Unix socket, xGMI reads), one contiguous range — the single-GPU kernels run on it
    unchanged.  The graph: copied device to device from `graph_from` (an index of this rank that has it), else the closed-form one."""
    from rad_amd.device import DeviceIndex
    from rad_amd.rendezvous import exchange_fds
    n, M = args.n, args.connectivity
    pidx = DeviceIndex(args.ndim, M, 2 * M, args.expansion_add, device=local_rank)
    rps = pidx.peer_create(rank, world, n)
    pidx.peer_fill_synth(seed=20260101, mode=args.corpus_mode)
    fd = pidx.peer_export()
    fds = exchange_fds(rank, world, fd, f"bench-{os.environ.get('MASTER_PORT', '0')}-{args.n}")
    for p_ in range(world):
        if p_ != rank:
            pidx.peer_import(p_, fds[p_])
            os.close(fds[p_])
    os.close(fd)
    grp.barrier()                      # every rank has imported what it needs before anybody can tear a shard down
    pidx.peer_seal()
    if graph_from is not None:
        pidx.copy_graph_from(graph_from)
    else:
        pidx.synth_graph(seed=777)
    return pidx, rps


def run_peer_leg(args, full, grp, rank, world, local_rank, barrier):
    """BASELINE's row partitioning at the single-GPU kernel's rate: rows sharded over the ranks, every rank maps all shards and
    traverses with the unchanged kernel.  Parity: the same queries on this rank's whole-corpus index (`full`, when it has one)
    must give the same scored lists (64-bit hashes of all of them)."""
    from rad_amd.device import DeviceTraversal
    pidx, rps = peer_index(args, grp, rank, world, local_rank, graph_from=full if args.graph == "built" else None)
    steps = min(args.steps, 4)
    batches = query_batches(pidx, 1 + steps, args.nq, args.n, 777 + rank)
    good = tot = 0
    if full is not None:
        ns = min(512, args.nq)
        a_ = DeviceTraversal(pidx, batches[0][:ns], args.n_to_score); a_.run(0)
        b_ = DeviceTraversal(full, batches[0][:ns], args.n_to_score); b_.run(0)
        good, tot = int((a_.result_hashes() == b_.result_hashes()).sum()), ns
        a_.close(); b_.close()
    leg = run_traversal_leg(args, pidx, batches, steps, 1, barrier, overlap=not args.no_overlap)
    res = {"pops": leg["pops"], "evals": leg["evals"], "elapsed": leg["elapsed"], "steps": steps, "rows_per_shard": rps,
           "index_bytes": int(pidx.info().device_bytes), "parity_ok": good, "parity_n": tot, "state_bytes": leg["state_bytes"],
           "busy_ms": leg["busy_ms"], "launches": leg["launches"]}
    pidx.close()
    return res


def peer_report(args, grp, pr, world, B):
    tot = grp.allreduce([pr["pops"], pr["evals"], pr["parity_ok"], pr["parity_n"]], "sum")
    el = float(grp.allreduce([pr["elapsed"]], "max")[0])
    idxb = grp.allgather_obj(pr["index_bytes"])
    alg = (float(tot[1]) * (B + 4) + float(tot[0]) * 4) / world
    return {"value": float(tot[0]) / el, "unit": "expansions/s", "ms_per_step": el / pr["steps"] * 1e3, "steps": pr["steps"], "warmup": 1,
            "evals_per_s": float(tot[1]) / el, "rows_per_shard": pr["rows_per_shard"], "index_bytes_per_rank": [int(b) for b in idxb],
            "parity_vs_whole_corpus_index": f"{int(tot[2])}/{int(tot[3])}" if tot[3] else None,
            "roofline_frac_per_gpu": alg / el / 1e9 / HBM_PEAK_GBS,
            "remote_share_of_row_reads": (world - 1) / world,
            "partitioning": f"rows sharded by contiguous slot range ({pr['rows_per_shard']} per GPU: ceil(N / G) rounded up to the 2-MiB granule), adjacency replicated; "
                            "every rank maps all shards into one virtual range (HIP VMM, dmabuf descriptors) and runs the single-GPU traversal kernel unchanged: "
                            "remote rows are 128-B reads over xGMI, no collective, no lock step; results bit-identical to one GPU by construction"}, int(tot[2]), int(tot[3])


def run_sharded_leg(args, idx, grp, rank, world, local_rank, barrier):
    """BASELINE's partitioning after the replicas leg: every rank still holds the whole corpus (that IS the replicas
    mode), so each rank computes the parity reference for its own traversals, then drops the other ranks' rows."""
    n, nq = args.n, args.sharded_nq
    # the side leg of a --mode replicas run: a batch of the sharded loop is ~36000 frontier steps (4 s on one GPU, more
    # with real collectives), so it times at most two batches behind one warm-up; `value` does not come from here
    args.sh_warmup, args.sh_steps = min(args.warmup, 1), min(args.steps, 2)
    n_batches = args.sh_warmup + args.sh_steps
    qrng = np.random.default_rng(99)
    firsts = [int(qrng.integers(0, n - world * nq)) for _ in range(n_batches)]
    Qall = [idx.read_vectors(f, world * nq) for f in firsts]          # rank-major, identical on every rank
    ns, nfull = min(256, nq // max(args.sharded_groups, 1)), 2
    mine = reference_sample(args, idx, Qall[-1][(rank + 1) * nq - ns:(rank + 1) * nq], 1, ns, ns, nfull)
    ref = {"scored": {rank: mine["scored"][0]}, "pops": {rank: mine["pops"][0]}, "nbr": {rank: mine["nbr"][0]},
           "lists": {(rank, i): mine["lists"][(0, i)] for i in range(nfull)}}
    rows = n // world
    first = rank * rows
    count = rows if rank < world - 1 else n - first
    idx.keep_rows(first, count)
    comms, note, rccl = make_comms(args, grp, rank, world, local_rank, max(args.sharded_groups, 1))
    res = drive_shards(args, grp, rank, world, idx, first, count, Qall, comms, barrier, ref, ns, nfull)
    res["exchange"] = ("host-staged buffers over the TCP group" if comms is None else "RCCL ncclAllGather + ncclReduceScatter on device buffers") + note
    res["rccl"] = rccl
    res["index_bytes"] = int(idx.info().device_bytes)
    res["setup"] = "after the replicas leg: every rank held the whole corpus, computed its own parity reference, then kept its rows (radhip_index_keep_rows)"
    for c in comms or []:
        c.close()
    return res


def run_sharded_native(args, grp, rank, world, local_rank, barrier):
    """--mode sharded: the shard-native setup (config[3]).  Every rank generates ONLY its rows; the graph is the
    closed-form one (every rank generates it, no row is read) or is built once on rank 0 and broadcast device to
    device; the parity reference is one sequential phase on rank 0 before it creates its shard, when the corpus fits."""
    from rad_amd.device import DeviceIndex
    n, nq, mode, M = args.n, args.sharded_nq, args.corpus_mode, args.connectivity
    rows = n // world
    first = rank * rows
    count = rows if rank < world - 1 else n - first
    args.sh_warmup, args.sh_steps = args.warmup, args.steps          # --mode sharded: `value` comes from here, exactly K steps
    n_batches = args.warmup + args.steps
    qrng = np.random.default_rng(99)
    firsts = [int(qrng.integers(0, n - world * nq)) for _ in range(n_batches)]
    Qall = synth_queries(args, local_rank, firsts, world * nq, mode)
    comms, note, rccl = make_comms(args, grp, rank, world, local_rank, max(args.sharded_groups, 1))
    synthetic = args.graph == "synthetic" and mode == 1
    row_stride = 16 * (1 << max(0, int(np.ceil(np.log2(max(1, (args.ndim + 127) // 128))))))
    graph_bytes = n * (2 * M * 4 + 5) + (n // max(M - 1, 1)) * M * 4
    fits = n * row_stride + graph_bytes < 215e9
    want_ref = args.sharded_reference == "rank0" or (args.sharded_reference == "auto" and fits)
    ns, nfull = min(64, nq // max(args.sharded_groups, 1)), 2
    ref, idx, t_build, peak_full = None, None, 0.0, 0
    if rank == 0 and (want_ref or not synthetic):
        full = DeviceIndex(args.ndim, M, 2 * M, args.expansion_add, device=local_rank)
        full.synth_vectors(n, seed=20260101, mode=mode)
        if synthetic:
            full.synth_graph(seed=777)
        else:
            t_build = time.perf_counter()
            full.link_resident(seed=777, max_batch=args.build_batch)      # rows already resident: no host copy of the corpus
            t_build = time.perf_counter() - t_build
        peak_full = int(full.info().device_bytes)
        if want_ref:
            ref = reference_sample(args, full, Qall[-1], world, nq, ns, nfull)
        if synthetic:
            full.close()
        else:
            full.keep_rows(first, count)
            idx = full
    ref = grp.broadcast_obj(ref)
    if idx is None:
        idx = DeviceIndex(args.ndim, M, 2 * M, args.expansion_add, device=local_rank)
        idx.synth_vectors_shard(count, first, n, seed=20260101, mode=mode)
        if synthetic:
            idx.synth_graph(seed=777)
    if not synthetic and world > 1:
        if comms is not None:
            idx.broadcast_graph(comms[0], 0)                    # RCCL broadcast of the device arrays over xGMI
        else:
            g = idx.read_graph() if rank == 0 else (None,) * 4
            inf0 = grp.broadcast_obj((int(idx.info().max_level), int(idx.info().entry)) if rank == 0 else None)
            arrs = [bcast_array(grp, a) for a in g]
            if rank != 0:
                idx.load_graph(arrs[0], arrs[1], arrs[2], arrs[3], inf0[0], inf0[1])
    res = drive_shards(args, grp, rank, world, idx, first, count, Qall, comms, barrier, ref, ns, nfull)
    res["exchange"] = ("host-staged buffers over the TCP group" if comms is None else "RCCL ncclAllGather + ncclReduceScatter on device buffers") + note
    res["rccl"] = rccl
    res["index_bytes"] = int(idx.info().device_bytes)
    res["t_build"] = t_build
    res["recall"] = None
    res["peak_full_bytes_rank0"] = peak_full
    res["setup"] = ("shard-native: every rank generated only its rows (radhip_index_synth_vectors_shard); graph = " +
                    ("closed form on every rank (no row is read)" if synthetic else
                     f"built once on rank 0 from rows resident there ({t_build:.0f} s), broadcast " + ("device to device (RCCL)" if comms is not None else "over the TCP group")) +
                    ("; parity reference: rank 0 held the whole corpus for one sequential phase before creating its shard" if want_ref else "; no parity reference (the corpus does not fit one GPU, or --sharded-reference none)"))
    idx.close()
    for c in comms or []:
        c.close()
    return res


def sharded_report(args, grp, sh, n, world):
    """sums over the ranks of one sharded leg -> the dict the JSON line carries"""
    tot = grp.allreduce([sh["pops"], sh["evals"], sh["parity_ok"], sh["parity_n"], sh["spec_asked"], sh["spec_used"], sh["spec_hits"]], "sum")
    el = float(grp.allreduce([sh["elapsed"]], "max")[0])
    idxb = grp.allgather_obj(sh["index_bytes"])
    rep = {
        "value": float(tot[0]) / el, "unit": "expansions/s", "ms_per_step": el / args.sh_steps * 1e3, "steps": args.sh_steps, "warmup": args.sh_warmup,
        "evals_per_s": float(tot[1]) / el, "traversals_per_gpu_per_step": args.sharded_nq, "groups_per_gpu": sh["groups"],
        "frontier_steps_per_step": sh["steps"] / max(args.sh_steps, 1),
        "exchanged_bytes_per_rank_per_step": sh["bytes"] / max(args.sh_steps, 1),
        "request_slots_per_traversal_per_step": sh["width"], "engine": sh["engine"], "slots_per_gpu": sh["slots"],
        "speculation": {"depth": sh["spec_depth"], "scores_requested": int(tot[4]), "scores_used": int(tot[5]),
                        "expansions_finished_from_them": int(tot[6]),
                        "wasted_evaluations": int(tot[4] - tot[5]),
                        "wasted_fraction_of_all_evaluations": float(tot[4] - tot[5]) / max(float(tot[1] + tot[4] - tot[5]), 1.0)},
        "parity_vs_single_gpu": f"{int(tot[2])}/{int(tot[3])}" if tot[3] else None, "exchange": sh["exchange"], "rccl": sh["rccl"],
        "setup": sh["setup"], "index_bytes_per_rank": [int(b) for b in idxb], "state_bytes_per_rank": sh["state_bytes"],
        "partitioning": f"one HNSW graph over all {n} rows (adjacency replicated), rows sharded by contiguous slot range "
                        f"({n // world} per GPU), traversals partitioned over the ranks; per frontier step an all-gather of "
                        f"the candidate slots and a reduce-scatter of their (and, or) scores; strict best-first, results "
                        f"bit-identical to one GPU",
    }
    return rep, int(tot[2]), int(tot[3])


# ------------------------------------------------------------------ --mode sharded: no rank holds the corpus
def main_sharded(args, grp, rank, world, local_rank, barrier):
    n = args.n
    box = {}

    def _leg():
        try:
            box["res"] = run_sharded_native(args, grp, rank, world, local_rank, barrier)
        except BaseException as e:   # noqa: BLE001 - reported, never silent
            box["err"] = f"{type(e).__name__}: {e}"
    th = threading.Thread(target=_leg, daemon=True)
    th.start()
    th.join(args.sharded_timeout)
    if th.is_alive() or "err" in box:
        why = box.get("err", f"no result after {args.sharded_timeout:.0f} s (hung collective?)")
        print(f"bench.py rank {rank}: the sharded leg failed ({why}); --mode sharded has no other value to print", file=sys.stderr)
        os._exit(4)      # (never a re-exec: this process has touched the GPU)
    sh = box["res"]
    sharded, p_ok, p_n = sharded_report(args, grp, sh, n, world)
    if p_ok != p_n:
        raise SystemExit(f"bench.py: the sharded traversals differ from the single-GPU ones ({sharded['parity_vs_single_gpu']}): no value printed")
    if rank != 0:
        grp.barrier()
        grp.close()
        return
    B = 16 * (1 << max(0, int(np.ceil(np.log2(max(1, (args.ndim + 127) // 128))))))
    tot_pops = sharded["value"] * sharded["ms_per_step"] * 1e-3 * args.steps
    tot_evals = sharded["evals_per_s"] * sharded["ms_per_step"] * 1e-3 * args.steps
    alg = (tot_evals * (B + 4) + tot_pops * 4) / max(args.steps, 1) / world
    ach = alg / (sharded["ms_per_step"] * 1e-3) / 1e9
    graph_desc = "closed-form synthetic graph" if (args.graph == "synthetic" and args.corpus_mode == 1) else \
        f"HNSW graph built on rank 0's GPU (expansion_add={args.expansion_add}, {sh['t_build']:.0f} s) and broadcast"
    corpus_desc = {0: "dense random corpus", 1: "two-level clustered sparse corpus (round 1)",
                   2: "hierarchical sparse corpus (neighbourhood structure at every scale)"}[args.corpus_mode]
    out = {
        "metric": METRIC, "value": sharded["value"], "unit": "expansions/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": sharded["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u64 popcount (integer)", "data": "synthetic",
        "config": {
            "workload": f"{n // 1_000_000}M x {args.ndim}-bit fingerprints SHARDED over {world} GPU(s) ({n // world} rows each), connectivity="
                        f"{args.connectivity} (level-0 width {2 * args.connectivity}), {args.sharded_nq} best-first RAD traversals per GPU and step to "
                        f"n_to_score={args.n_to_score}, synthetic {corpus_desc}, {graph_desc}",
            "rows": n, "ndim": args.ndim, "connectivity": args.connectivity, "nq_per_gpu": args.sharded_nq, "n_to_score": args.n_to_score,
            "corpus_mode": args.corpus_mode,
            "parallelism": "row-sharded (--mode sharded): " + sharded["partitioning"] + "; exchange = " + sharded["exchange"]},
        "evals_per_s": sharded["evals_per_s"], "evals_per_expansion": sharded["evals_per_s"] / max(sharded["value"], 1.0),
        "roofline": {"bound": "hbm", "kernel": "shard_step_kernel + shard_eval_kernel (one frontier step)", "achieved": ach, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": alg,
                     "avg_launch_ms": sharded["ms_per_step"],
                     "note": "a batch of the row-sharded mode is ~10^4 frontier steps, each a step kernel that ends with its slowest pop plus two small "
                             "collectives: bound by latency, not by HBM; avg_launch_ms is the whole batch"},
        "sharded": sharded,
    }
    print(json.dumps(out), flush=True)
    grp.barrier()
    grp.close()


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

    # (--mode sharded: the locality layout — pair rows, twice the adjacency — would only serve rank 0's parity reference)
    if args.table != "auto" and not (args.mode == "sharded" and args.table == "local"):
        os.environ["RADHIP_TABLE"] = args.table
    n = args.n
    if args.mode == "sharded":
        return main_sharded(args, grp, rank, world, local_rank, barrier)
    idx, t_build, lay = build_index(args, args.corpus_mode, local_rank)
    info = idx.info()
    B = info.row_stride
    n_batches = args.warmup + args.steps
    if args.nq <= 0:
        # four times what the device holds resident at once, or as many as it has room for beside the index (3.3 MB of
        # state per traversal to n_to_score = 100k): the traversals of a batch differ in length (+-18 %, the longest 2 x
        # the mean) and a launch ends with its longest one running alone — ~130 ms whatever the batch.  Measured on 20M
        # rows: 1.32 / 1.44 / 1.62 G expansions/s at 1 / 2 / 4 resident rounds per launch; the marginal rate between
        # them is 1.86 G (profiles/r03)
        # Round 4: the tables live per resident row, so what a batch costs beyond the rows' 40 GB is its scored lists
        # (0.8 MB per traversal), and two objects take the batches in turn (94 GB each at 4 x 16384).
        from rad_amd._lib import RadHipError, E_NOMEM
        from rad_amd.device import DeviceTraversal
        cap = idx.traversal_capacity()
        # two objects (overlapped launches) with four resident rounds each when there is room; beside a corpus that fills most of the
        # device (1B rows: 202 GB) one object with as many rounds as fit
        chained = args.chain > 1 and not args.no_overlap and args.steps > 1
        plans = ([] if (args.no_overlap or chained) else [(4.0, 2), (3.0, 2), (2.0, 2)]) + [(4.0, 1), (3.0, 1), (2.0, 1), (1.5, 1), (1.0, 1)]
        for mult, n_obj in plans:
            args.nq = int(cap * mult)
            probes = []
            try:
                if chained:        # one object: the rows' tables + a ring of one step's scored lists + a chain's headers and queries
                    c_ = max(1, min(args.chain, max(args.steps, args.warmup)))
                    probes.append(DeviceTraversal(idx, idx.read_vectors(0, c_ * args.nq), args.n_to_score, list_ring=args.nq))
                else:
                    q0 = idx.read_vectors(0, args.nq)
                    for _ in range(n_obj):
                        probes.append(DeviceTraversal(idx, q0, args.n_to_score, slots=True))
                args.objects = n_obj
                break
            except RadHipError as e:
                if e.code != E_NOMEM or (mult, n_obj) == plans[-1]:
                    raise
            finally:
                for p_ in probes:
                    p_.close()
        args.nq = int(grp.allreduce([args.nq], "min")[0])       # the same batch size on every rank
        args.objects = int(grp.allreduce([getattr(args, "objects", 2)], "min")[0])
    # replicas / single GPU: every rank runs its OWN query batches (the queries are what is split)
    batches = query_batches(idx, n_batches, args.nq, n, 4242 + rank)
    leg = run_traversal_leg(args, idx, batches, args.steps, args.warmup, barrier, overlap=not args.no_overlap)
    recall = recall_of(idx, batches[-1][:128]) if (rank == 0 and args.graph == "built") else None
    recall400 = recall_of(idx, batches[-1][:128], ef=400) if (rank == 0 and args.graph == "built") else None
    note(f"recall@10 of the built graph: {recall} at ef 128, {recall400} at ef 400")

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
        "corpus_mode": args.corpus_mode, "expansion_add": args.expansion_add, "graph_build_s": t_build, "graph_build_batch": args.build_batch,
        "graph_recall_at_10_ef128": recall, "graph_recall_at_10_ef400": recall400,
        "traversal_state": {"objects": leg["objects"], "rows_with_tables_per_object": leg["slots"] or args.nq, "bytes": leg["state_bytes"],
                            "steps_per_launch": leg.get("chain", 1), "scored_list_ring": leg.get("list_ring", 0),
                            "what": "tables / key pool / run table per resident row of the kernel (RADHIP_TRAV_SLOTS), query + header per traversal; "
                                    + ("up to --chain steps go into ONE launch (rows take the traversals of the whole chain from one counter: a step's longest "
                                       "traversals finish beside the next step's), scored lists in a ring that keeps the last step's (radhip_traversal_create_ring)"
                                       if leg.get("chain", 1) > 1 else
                                       "a scored list per traversal; two objects on two streams take the batches in turn (radhip_traversal_start / _finish)")},
        "layout": None if lay is None else {"seconds": lay.seconds, "groups_per_row": lay.groups_per_row, "degree": lay.degree},
        "parallelism": "single GPU",
    }
    out["config"] = config
    out["evals_per_s"] = float(sums[1]) / elapsed_max
    out["evals_per_expansion"] = float(sums[1]) / max(float(sums[0]), 1.0)
    out["queue_per_traversal"] = {"repivots": leg["repivots"], "remids": leg["remids"], "flushes": leg["flushes"]}
    out["roofline"] = roofline_of(leg, B)
    out["build_id"] = _lib.build_id()
    out["traverse_build_id"] = _lib.traverse_build_id()
    prof = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(prof):
        try:
            with open(prof) as f:
                pj = json.load(f)
            # measured offline (rocprofv3 --pmc passes): only valid for the kernels it was measured on — the hash of the
            # traversal kernels' sources must match, or the figure is dropped, not printed
            if (pj.get("traverse_build_id") == _lib.traverse_build_id() and pj.get("n") == n and pj.get("nq") == args.nq and pj.get("n_to_score") == args.n_to_score
                    and pj.get("corpus_mode", 1) == args.corpus_mode and pj.get("table") == leg["table"] and pj.get("expansion_add", 64) == args.expansion_add):
                rf = out["roofline"]
                # the counters were taken on ONE launch of one batch; a launch of this run is a chain of steps: per-expansion counts,
                # scaled by the expansions of a launch here
                scale = (leg["pops"] / max(leg["launches"], 1)) / max(pj.get("expansions_per_launch") or 1.0, 1.0)
                tr = pj.get("hbm_bytes_per_launch")
                tr = tr * scale if tr else tr
                rf["traffic"] = tr
                rf["traffic_per_expansion_bytes"] = (pj.get("hbm_bytes_per_launch") or 0) / max(pj.get("expansions_per_launch") or 1.0, 1.0)
                rf["traffic_source"] = "profiles/traffic_latest.json (rocprofv3 --pmc, separate passes, measured offline; every memory-side read request is a 128-B line, partial writes are 32-B requests: profiles/r04/probe_request_size.md)"
                busy_per_launch = rf["kernel_busy_ms_per_launch"] * 1e-3
                if tr:
                    rf["hbm_real_gbs"] = tr / busy_per_launch / 1e9
                # The ceiling this kernel really sits under: memory-side REQUESTS.  scripts/probe_request_size.hip (profiles/r04):
                # an MI355X serves 38-49 G random 128-B line reads per second over a 32 GiB footprint (38 when a 16-B-per-lane
                # load touches 64 lines, 49 when it touches 32 or fewer), 49 G full 64-B writes per second, and 21.9 G writes of
                # 32 B or less per second into lines that are not in the L2 — but a 4-B store into the line a probe of the same
                # round has fetched (the kernel's table entry store) goes at the 64-B rate: the expansion-shaped mix runs at
                # 2.12 G rounds/s with its stores into probed lines or as whole sectors, 1.82 G/s with them elsewhere.
                rq, wq = pj.get("read_requests_128B"), (pj.get("write_requests") or {}).get("total")
                w64 = (pj.get("write_requests") or {}).get("64B") or 0
                if rq and wq:
                    rq, wq, w64 = rq * scale, wq * scale, w64 * scale
                    rate = (rq + wq) / busy_per_launch / 1e9
                    rf["memory_requests"] = {
                        "reads_per_launch": rq, "writes_per_launch": wq, "writes_64B_per_launch": w64, "achieved_G_per_s": rate,
                        "device_random_line_reads_G_per_s": RANDOM_LINE_PEAK_G, "frac": rate / RANDOM_LINE_PEAK_G}
                    # what the SAME request mix could reach if nothing but the memory system's request rates bound it
                    t_model = rq / (READ_LINES_G * 1e9) + wq / (FULL_WRITES_G * 1e9)
                    t_cold = rq / (READ_LINES_G * 1e9) + (wq - w64) / (PARTIAL_WRITES_G * 1e9) + w64 / (FULL_WRITES_G * 1e9)
                    alg = rf["algorithmic_bytes_per_launch"]
                    rf["attainable"] = {
                        "by_request_count_GBs": alg / ((rq + wq) / (RANDOM_LINE_PEAK_G * 1e9)) / 1e9,
                        "by_request_cost_model_GBs": alg / t_model / 1e9,
                        "model": f"time >= reads / {READ_LINES_G} G/s + writes / {FULL_WRITES_G} G/s: the kernel's writes of <= 32 B are table entries "
                                 "stored into the line their probe has just fetched, which cost what a 64-B write costs "
                                 "(rates measured by scripts/probe_request_size.hip on this device class, 32 GiB footprint)",
                        "if_small_writes_missed_the_L2_GBs": alg / t_cold / 1e9,
                        "achieved_over_model": rf["achieved"] / (alg / t_model / 1e9)}
        except Exception:
            pass

    armed = False
    if world > 1 and rank == 0 and args.mode == "replicas":
        # The measured leg is done.  The side legs below run paths no single-GPU box can rehearse (reads over xGMI, multi-rank
        # RCCL): should a GPU fault end this process inside the HIP runtime, or the launcher end it because another rank died,
        # the library writes this line instead — the replicas result and what happened (include/rad_hip.h radhip_arm_last_words)
        try:
            from rad_amd import _lib as _lw
            last = dict(out)
            last["config"] = dict(config, parallelism="replicas (--mode replicas): every GPU holds the whole corpus and graph, the query batch is split, no collective")
            why = {"error": "the process was ended by a signal inside the side legs (a GPU fault, or the launcher's SIGTERM after another rank died): not measured"}
            last["peer_mapped"], last["sharded"] = why, why
            last["replicas"] = {"value": value_replicas, "unit": "expansions/s"}
            armed = _lw.lib().radhip_arm_last_words(json.dumps(last).encode()) == 0
        except Exception:   # noqa: BLE001 - a guard, not a requirement
            armed = False

    if world > 1 and not args.no_peer_leg:
        # side leg 1: the peer-mapped corpus (rows sharded, single-GPU kernel over xGMI reads); this rank's whole-corpus index is
        # still here and serves as the parity reference
        pbox = {}

        def _pleg():
            try:
                pbox["res"] = run_peer_leg(args, idx, grp, rank, world, local_rank, barrier)
            except BaseException as e:   # noqa: BLE001 - reported in the line, never silent
                pbox["err"] = f"{type(e).__name__}: {e}"
        th = threading.Thread(target=_pleg, daemon=True)
        th.start()
        th.join(args.sharded_timeout)
        if th.is_alive():
            # a rank is stuck inside the leg (a peer that died in the descriptor exchange): the replicas leg is measured, print and leave
            if rank == 0:
                config["parallelism"] = "replicas (--mode replicas): every GPU holds the whole corpus and graph, the query batch is split, no collective"
                out["peer_mapped"] = {"error": f"no result after {args.sharded_timeout:.0f} s"}
                out["replicas"] = {"value": value_replicas, "unit": "expansions/s"}
                print(json.dumps(out), flush=True)
            os._exit(0 if args.mode == "replicas" else 4)
        errs = [e for e in grp.allgather_obj(pbox.get("err", "")) if e]
        if errs:
            out["peer_mapped"] = {"error": "; ".join(sorted(set(errs)))[:400]}
            if args.mode == "peer":
                raise SystemExit(f"bench.py: the peer-mapped leg failed ({out['peer_mapped']['error']}) and --mode peer asked for its value")
        else:
            peer, pp_ok, pp_n = peer_report(args, grp, pbox["res"], world, B)
            if pp_ok != pp_n:
                raise SystemExit(f"bench.py: traversals over the peer-mapped corpus differ from the whole-corpus index ({peer['parity_vs_whole_corpus_index']}): no value printed")
            out["peer_mapped"] = peer
            note(f"peer-mapped leg: {peer['value'] / 1e9:.3f} G expansions/s over {world} GPUs")

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
        sharded, p_ok, p_n = sharded_report(args, grp, sh, n, world)
        replicas = {"value": value_replicas, "unit": "expansions/s", "ms_per_step": out["ms_per_step"],
                    "partitioning": "every GPU holds the whole corpus and graph, the query batch is split, no collective"}
        if p_ok != p_n:
            raise SystemExit(f"bench.py: the sharded traversals differ from the single-GPU ones ({sharded['parity_vs_single_gpu']}): no value printed")
        if args.mode == "sharded":
            out["value"], out["ms_per_step"] = sharded["value"], sharded["ms_per_step"]
            out["evals_per_s"] = sharded["evals_per_s"]
            config["parallelism"] = "row-sharded (--mode sharded): " + sharded["partitioning"] + "; exchange = " + sharded["exchange"]
            config["nq_per_gpu"] = args.sharded_nq
            out["roofline"]["note"] = ("roofline of the single-GPU traversal kernel on this rank (replicas leg); the sharded step is "
                                       "bound by its two collectives per frontier step, not by HBM")
        elif args.mode == "peer" and "value" in out.get("peer_mapped", {}):
            pm = out["peer_mapped"]
            out["value"], out["ms_per_step"], out["evals_per_s"] = pm["value"], pm["ms_per_step"], pm["evals_per_s"]
            out["steps"] = pm["steps"]
            config["parallelism"] = "peer-mapped (--mode peer): " + pm["partitioning"]
        else:
            config["parallelism"] = "replicas (--mode replicas): " + replicas["partitioning"]
        out["sharded"], out["replicas"] = sharded, replicas

    if armed:
        from rad_amd import _lib as _lw
        _lw.lib().radhip_arm_last_words(None)

    if rank != 0:
        grp.barrier()
        grp.close()
        return

    if world == 1:
        if not args.no_cpu_baseline:
            note("cpu_baseline: copying corpus and graph to the host")
            # the parity sample alternates between the head and the tail of the batch: the first traversals of a launch are the
            # ones its rows start with, the last ones are taken over by rows that have finished others (traverse4.inc `take`)
            nqb, blk = batches[-1].shape[0], 32 * host_cores()
            head = [np.arange(i, min(i + blk, nqb)) for i in range(0, min(4096, nqb), blk)]
            tail = [np.arange(max(nqb - i - blk, 0), nqb - i) for i in range(0, min(4096, nqb), blk)]
            sel = np.concatenate([x for pair in zip(head, tail) for x in pair])
            sel = sel[np.sort(np.unique(sel, return_index=True)[1])]          # (small batches: head and tail overlap)
            st_l = leg["last_stats"]
            picked = type("S", (), {"n_scored": st_l.n_scored[sel], "n_pops": st_l.n_pops[sel], "n_nbr": st_l.n_nbr[sel]})
            cb, sample, ok = cpu_baseline(idx, batches[-1][sel], picked, args, leg["last_hashes"][sel])
            note(f"cpu_baseline done, parity sample {sample}")
            out["cpu_baseline"] = cb
            out["parity_sample"] = sample
            out["parity_sample_what"] = ("oracle vs GPU per traversal: scored / expansions / neighbours counters AND a 64-bit order-sensitive hash of the whole scored "
                                         "list (slots, and, or); traversals taken alternately from the head and the tail of the last batch")
            if not ok:
                raise SystemExit(f"bench.py: GPU and oracle disagree on the parity sample ({sample}): no value printed")
        if not args.no_kernel_legs:
            try:
                out["kernels"] = kernel_legs(idx, n, B)
                note(f"kernel legs: {out['kernels']}")
            except Exception as e:   # noqa: BLE001 - a side leg must not cost the line
                out["kernels"] = {"error": f"{type(e).__name__}: {e}"}
        idx.close()
        if args.secondary_expansion_add and args.secondary_expansion_add != args.expansion_add and args.graph == "built":
            # the graph rounds 1-3 benched (expansion_add 64: a lighter build, more new nodes per expansion), on the same kernel
            try:
                a2 = argparse.Namespace(**vars(args)); a2.expansion_add = args.secondary_expansion_add
                idx2, tb2, _ = build_index(a2, args.corpus_mode, local_rank, layout=(args.table in ("group", "local")))
                st2 = min(args.steps, 4)
                b2 = query_batches(idx2, 1 + st2, args.nq, n, 4242)
                leg2 = run_traversal_leg(a2, idx2, b2, st2, 1, lambda: None, overlap=not args.no_overlap)
                rf2 = roofline_of(leg2, B)
                out["secondary_graph"] = {
                    "value": leg2["pops"] / leg2["elapsed"], "unit": "expansions/s", "ms_per_step": leg2["elapsed"] / st2 * 1e3, "steps": st2, "warmup": 1,
                    "evals_per_s": leg2["evals"] / leg2["elapsed"], "evals_per_expansion": leg2["evals"] / max(leg2["pops"], 1),
                    "roofline_frac": rf2["frac"], "roofline_achieved_gbs": rf2["achieved"], "avg_launch_ms": rf2["avg_launch_ms"],
                    "kernel_busy_ms_per_launch": rf2["kernel_busy_ms_per_launch"], "table": leg2["table"],
                    "expansion_add": a2.expansion_add, "graph_build_s": tb2,
                    "graph_recall_at_10_ef128": recall_of(idx2, b2[-1][:128]), "graph_recall_at_10_ef400": recall_of(idx2, b2[-1][:128], ef=400),
                    "workload": f"the bench workload of rounds 1-3: the same corpus and kernel, the graph built with expansion_add={a2.expansion_add}"}
                idx2.close()
                note(f"secondary graph (expansion_add {a2.expansion_add}): {out['secondary_graph']['value'] / 1e9:.3f} G expansions/s")
            except Exception as e:   # noqa: BLE001
                out["secondary_graph"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_config_legs:
            try:
                out["configs"] = config_legs(args, local_rank)
            except Exception as e:   # noqa: BLE001
                out["configs"] = {"error": f"{type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)
    grp.barrier()
    grp.close()


if __name__ == "__main__":
    main()
