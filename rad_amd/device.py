"""Thin object wrappers over the librad_hip C ABI: DeviceIndex and DeviceTraversal.

These hold opaque C handles; all arithmetic of the hot path happens in the HIP
kernels behind them (rad_amd/csrc).  numpy is used only to own host buffers.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._lib import NO_SLOT, RadHipError, check, ptr


def distance_f32(and_cnt, or_cnt) -> np.ndarray:
    """Float edge value of the integer counts: 1.0f - (float)and/(float)or in
    float32 (one division, one subtraction), 0.0 when or == 0."""
    a = np.asarray(and_cnt, dtype=np.float32)
    o = np.asarray(or_cnt, dtype=np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.divide(a, o, dtype=np.float32)
        d = np.subtract(np.float32(1.0), q, dtype=np.float32)
    return np.where(np.asarray(or_cnt) == 0, np.float32(0.0), d).astype(np.float32)


class DeviceIndex:
    """Corpus + layered adjacency resident in HBM (radhip_index_t)."""

    def __init__(self, ndim: int, connectivity: int, connectivity_base: int = 0,
                 expansion_add: int = 128, device: int = 0):
        self._h = C.c_void_p()
        self._L = _lib.lib()
        check(self._L.radhip_index_create(ndim, connectivity, connectivity_base, expansion_add,
                                          device, C.byref(self._h)))
        self.row_bytes = (ndim + 7) // 8

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.radhip_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- metadata ---------------------------------------------------------
    def info(self) -> _lib.IndexInfo:
        out = _lib.IndexInfo()
        check(self._L.radhip_index_info(self._h, C.byref(out)))
        return out

    # -- corpus -----------------------------------------------------------
    def load_vectors(self, rows: np.ndarray) -> None:
        rows = _lib.as_rows(rows, self.row_bytes)
        check(self._L.radhip_index_load_vectors(self._h, ptr(rows), rows.shape[0]))

    def synth_vectors(self, n: int, seed: int, mode: int = 1, first_row: int = 0,
                      n_total: Optional[int] = None) -> None:
        check(self._L.radhip_index_synth_vectors(self._h, n, first_row,
                                                 n if n_total is None else n_total, seed, mode))

    def load_vectors_shard(self, rows: np.ndarray, first: int, n_total: int) -> None:
        """One rank's rows only: `rows` are slots [first, first + len(rows)) of a corpus of n_total rows.  The
        index never holds the rest; only the sharded traversal (DeviceShard) reads its fingerprints."""
        rows = _lib.as_rows(rows, self.row_bytes)
        check(self._L.radhip_index_load_vectors_shard(self._h, ptr(rows), first, rows.shape[0], n_total))

    def synth_vectors_shard(self, count: int, first_row: int, n_total: int, seed: int, mode: int = 1) -> None:
        """Rows [first_row, first_row + count) of the closed-form corpus of n_total rows as a shard: they keep
        their global slots, the rest of the corpus never exists on this device."""
        check(self._L.radhip_index_synth_vectors_shard(self._h, count, first_row, n_total, seed, mode))

    def link_resident(self, seed: int = 0, max_batch: int = 4096) -> None:
        """Link the rows that are already resident (load_vectors / synth_vectors) into the graph, exactly as
        add_rows would have — without a host copy of the corpus going through the call."""
        check(self._L.radhip_index_link_resident(self._h, seed, max_batch))

    def broadcast_graph(self, comm: "RcclComm", root: int = 0) -> None:
        """The adjacency of `root` on every rank (ncclBroadcast of the device arrays over xGMI)."""
        check(self._L.radhip_index_broadcast_graph(self._h, comm._h, root))

    def read_vectors(self, first: int, count: int) -> np.ndarray:
        out = np.empty((count, self.row_bytes), np.uint8)
        check(self._L.radhip_index_read_vectors(self._h, first, count, ptr(out)))
        return out

    # -- graph ------------------------------------------------------------
    def load_graph(self, levels, adj0, upper_row, adjU, max_level: int, entry: int) -> None:
        inf = self.info()
        levels = np.ascontiguousarray(levels, np.int8)
        n = levels.shape[0]
        adj0 = np.ascontiguousarray(adj0, np.uint32).reshape(n, inf.connectivity_base)
        upper_row = np.ascontiguousarray(upper_row, np.uint32).reshape(n)
        adjU = np.ascontiguousarray(adjU, np.uint32).reshape(-1, inf.connectivity)
        check(self._L.radhip_index_load_graph(self._h, n, max_level, entry, ptr(levels), ptr(adj0),
                                              ptr(upper_row), ptr(adjU), adjU.shape[0]))

    def synth_graph(self, seed: int) -> None:
        check(self._L.radhip_index_synth_graph(self._h, seed))

    def read_graph(self):
        inf = self.info()
        n = inf.n
        levels = np.empty(n, np.int8)
        adj0 = np.empty((n, inf.connectivity_base), np.uint32)
        upper_row = np.empty(n, np.uint32)
        adjU = np.empty((inf.n_upper_rows, inf.connectivity), np.uint32)
        check(self._L.radhip_index_read_graph(self._h, ptr(levels), ptr(adj0), ptr(upper_row), ptr(adjU)))
        return levels, adj0, upper_row, adjU

    def get_neighbors(self, slot: int, level: int) -> np.ndarray:
        out = np.empty(64, np.uint32)
        n = C.c_uint32(0)
        check(self._L.radhip_get_neighbors(self._h, slot, level, ptr(out), 64, C.byref(n)))
        return out[: n.value].copy()

    def get_top_level_nodes(self) -> np.ndarray:
        n = C.c_uint64(0)
        check(self._L.radhip_get_top_level_nodes(self._h, None, 0, C.byref(n)))
        out = np.empty(n.value, np.uint32)
        check(self._L.radhip_get_top_level_nodes(self._h, ptr(out), n.value, C.byref(n)))
        return out

    def optimize_layout(self, n_threads: int = 0) -> "_lib.LayoutInfo":
        """Compute the graph-locality layout the grouped visited table is keyed by (performance only:
        no result depends on it).  Returns its statistics."""
        check(self._L.radhip_index_optimize_layout(self._h, n_threads))
        return self.layout_info()

    def layout_info(self) -> "_lib.LayoutInfo":
        out = _lib.LayoutInfo()
        check(self._L.radhip_index_layout_info(self._h, C.byref(out)))
        return out

    def set_layout(self, lid: np.ndarray) -> None:
        lid = np.ascontiguousarray(lid, np.uint32)
        check(self._L.radhip_index_set_layout(self._h, ptr(lid)))

    def read_layout(self) -> np.ndarray:
        out = np.empty(self.info().n, np.uint32)
        check(self._L.radhip_index_read_layout(self._h, ptr(out)))
        return out

    def keep_rows(self, first: int, count: int) -> None:
        """Row-sharded multi-GPU mode: keep rows [first, first+count) of the corpus, free the rest."""
        check(self._L.radhip_index_keep_rows(self._h, first, count))

    # -- peer-mapped corpus: the row shards of all ranks of a node in ONE virtual range (xGMI reads, no lock step) --------
    def peer_create(self, rank: int, world: int, n_total: int) -> int:
        """Reserve the range for n_total rows and create this rank's shard; returns the rows per shard (ceil(n_total / world)
        rounded up to the 2-MiB allocation granule).  Rank r holds rows [r * rows_per_shard, min((r + 1) * rows_per_shard, n_total))."""
        rps = C.c_uint64(0)
        check(self._L.radhip_index_peer_create(self._h, rank, world, n_total, C.byref(rps)))
        return int(rps.value)

    def peer_fill_synth(self, seed: int, mode: int = 1) -> None:
        check(self._L.radhip_index_peer_fill_synth(self._h, seed, mode))

    def peer_fill_rows(self, rows: np.ndarray) -> None:
        rows = _lib.as_rows(rows, self.row_bytes) if len(rows) else np.zeros((0, self.row_bytes), np.uint8)
        check(self._L.radhip_index_peer_fill_rows(self._h, ptr(rows) if rows.shape[0] else None, rows.shape[0]))

    def peer_export(self) -> int:
        """dmabuf file descriptor of this rank's shard (the caller hands it to the peers and closes it)"""
        fd = C.c_int(-1)
        check(self._L.radhip_index_peer_export(self._h, C.byref(fd)))
        return int(fd.value)

    def peer_import(self, peer_rank: int, fd: int) -> None:
        check(self._L.radhip_index_peer_import(self._h, peer_rank, fd))

    def peer_seal(self) -> None:
        """all shards mapped: the index holds the whole corpus from here on (read-only)"""
        check(self._L.radhip_index_peer_seal(self._h))

    def copy_graph_from(self, src: "DeviceIndex") -> None:
        """the graph of `src` (same device, same process), device to device"""
        check(self._L.radhip_index_copy_graph_from(self._h, src._h))

    def traversal_capacity(self) -> int:
        """Traversals resident on the device at once (one wavefront each)."""
        n = C.c_uint32(0)
        check(self._L.radhip_traversal_resident_capacity(self._h, C.byref(n)))
        return n.value

    # -- Tanimoto kernels ---------------------------------------------------
    def scan(self, queries: np.ndarray, first: int = 0, count: Optional[int] = None):
        """K1: (and, or) of every query against rows [first, first+count)."""
        q = _lib.as_rows(queries, self.row_bytes, "queries")
        if count is None:
            count = self.info().n - first
        a = np.empty((q.shape[0], count), np.uint32)
        o = np.empty((q.shape[0], count), np.uint32)
        check(self._L.radhip_tanimoto_scan(self._h, ptr(q), q.shape[0], first, count, ptr(a), ptr(o)))
        return a, o

    def add_rows(self, rows: np.ndarray, seed: int = 0, max_batch: int = 4096) -> None:
        """Append rows and link them into the graph on the GPU (radhip_index_add; rad_amd.index.Index.add
        is the keyed front end of this)."""
        r = _lib.as_rows(rows, self.row_bytes, "rows")
        check(self._L.radhip_index_add(self._h, ptr(r), r.shape[0], seed, max_batch))

    def topk(self, queries: np.ndarray, k: int, first: int = 0, count: Optional[int] = None):
        """K1 reduced on the chip: the k nearest rows of [first, first+count) per query in (distance,
        slot) order.  Returns (slots [nq, k], and, or, counts); rows shorter than k are padded."""
        q = _lib.as_rows(queries, self.row_bytes, "queries")
        if count is None:
            count = self.info().n - first
        nq = q.shape[0]
        s = np.empty((nq, k), np.uint32)
        a = np.empty((nq, k), np.uint32)
        o = np.empty((nq, k), np.uint32)
        c = np.zeros(nq, np.uint32)
        check(self._L.radhip_tanimoto_topk(self._h, ptr(q), nq, k, first, count, ptr(s), ptr(a), ptr(o), ptr(c)))
        return s, a, o, c

    def gather(self, queries: np.ndarray, cand_slots: np.ndarray, cand_offsets: np.ndarray):
        """K2: (and, or) of query i against cand_slots[cand_offsets[i]:cand_offsets[i+1]]."""
        q = _lib.as_rows(queries, self.row_bytes, "queries")
        s = np.ascontiguousarray(cand_slots, np.uint32)
        off = np.ascontiguousarray(cand_offsets, np.uint64)
        if off.shape[0] != q.shape[0] + 1 or int(off[-1]) != s.shape[0]:
            raise ValueError("cand_offsets must have nq+1 entries ending at len(cand_slots)")
        a = np.empty(s.shape[0], np.uint32)
        o = np.empty(s.shape[0], np.uint32)
        check(self._L.radhip_tanimoto_gather(self._h, ptr(q), q.shape[0], ptr(s), ptr(off), ptr(a), ptr(o)))
        return a, o


@dataclass
class TraversalStats:
    n_scored: np.ndarray
    n_pops: np.ndarray
    n_nbr: np.ndarray
    status: np.ndarray
    n_repivot: np.ndarray = None
    n_flush: np.ndarray = None
    n_remid: np.ndarray = None
    n_upper: np.ndarray = None


# numpy view of _lib.TravStats (include/rad_hip.h radhip_trav_stats_t)
_TRAV_STATS_DTYPE = np.dtype([("n_scored", "<u8"), ("n_pops", "<u8"), ("n_nbr", "<u8"), ("n_repivot", "<u8"),
                              ("n_flush", "<u8"), ("status", "<i4"), ("n_remid", "<i4"), ("n_upper", "<u8")])
assert _TRAV_STATS_DTYPE.itemsize == C.sizeof(_lib.TravStats)


class DeviceTraversal:
    """nq independent RAD traversals, Tanimoto-scored, state in HBM (radhip_traversal_t)."""

    def __init__(self, index: DeviceIndex, queries: np.ndarray, n_to_score: int, log_pops: bool = False,
                 slots: bool = False, own_stream: bool = False, list_ring: int = 0):
        """slots: the heavy state (tables, queue pools) once per resident row of the kernel instead of once per traversal —
        such a batch runs to completion (no max_pops, no set_targets); own_stream: a HIP stream of the object's own, so
        that start() / finish() of two objects overlap."""
        self._L = _lib.lib()
        self.index = index
        q = _lib.as_rows(queries, index.row_bytes, "queries")
        self.nq = q.shape[0]
        self.n_to_score = int(n_to_score)
        self._h = C.c_void_p()
        flags = (_lib.TRAV_LOG_POPS if log_pops else 0) | (_lib.TRAV_SLOTS if slots else 0) | (_lib.TRAV_OWN_STREAM if own_stream else 0)
        if list_ring:
            # chained batches: per-row state and a ring of scored lists — results() / result_hashes() reach the last `list_ring`
            # traversals of a batch, stats() all of them
            check(self._L.radhip_traversal_create_ring(index._h, ptr(q), self.nq, self.n_to_score, flags, int(list_ring), C.byref(self._h)))
        else:
            check(self._L.radhip_traversal_create(index._h, ptr(q), self.nq, self.n_to_score, flags, C.byref(self._h)))
        self.slots = int(self._L.radhip_traversal_slots(self._h))   # 0: state per traversal (flag not given, or not applicable)
        self.list_ring = int(self._L.radhip_traversal_list_ring(self._h))   # 0: one scored list per traversal
        self.n_active = self.nq

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.radhip_traversal_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, queries: np.ndarray) -> None:
        """re-arm the object with new queries: all nq of them, or the first len(queries) <= nq (the others rest)"""
        q = _lib.as_rows(queries, self.index.row_bytes, "queries")
        if not 0 < q.shape[0] <= self.nq:
            raise ValueError(f"reset takes 1..{self.nq} queries")
        check(self._L.radhip_traversal_reset_count(self._h, ptr(q), q.shape[0]))
        self.n_active = q.shape[0]

    def run(self, max_pops: int = 0) -> int:
        """Advance every unfinished traversal by at most max_pops expansions
        (0 = to completion); returns how many are still running."""
        running = C.c_uint32(0)
        check(self._L.radhip_traversal_run(self._h, max_pops, C.byref(running)))
        return running.value

    def start(self) -> None:
        """enqueue the launch of the (re-armed) batch and return; finish() waits for it"""
        check(self._L.radhip_traversal_start(self._h))

    def finish(self) -> int:
        running = C.c_uint32(0)
        check(self._L.radhip_traversal_finish(self._h, C.byref(running)))
        return running.value

    def elapsed_to(self, other: "DeviceTraversal") -> float:
        """ms on the device's clock from the start of this object's last launch to the end of `other`'s last launch"""
        ms = C.c_double(0)
        check(self._L.radhip_traversal_elapsed_between(self._h, other._h, C.byref(ms)))
        return ms.value

    def launch_interval(self):
        """(start, end) of the last finished launch in ms on the device's clock since a fixed point of the process"""
        a, b = C.c_double(0), C.c_double(0)
        check(self._L.radhip_traversal_launch_interval(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def stats(self) -> TraversalStats:
        arr = (_lib.TravStats * self.nq)()
        check(self._L.radhip_traversal_stats(self._h, arr))
        rec = np.frombuffer(arr, dtype=_TRAV_STATS_DTYPE, count=self.nq)[:self.n_active]   # one view, no per-record Python
        return TraversalStats(rec["n_scored"].astype(np.int64), rec["n_pops"].astype(np.int64),
                              rec["n_nbr"].astype(np.int64), rec["status"].astype(np.int32),
                              rec["n_repivot"].astype(np.int64), rec["n_flush"].astype(np.int64), rec["n_remid"].astype(np.int64),
                              rec["n_upper"].astype(np.int64))

    def results(self, q: int):
        """(slots, and, or) of traversal q in traversal order."""
        n = C.c_uint64(0)
        check(self._L.radhip_traversal_results(self._h, q, None, None, None, 0, C.byref(n)))
        k = n.value
        s = np.empty(k, np.uint32)
        a = np.empty(k, np.uint32)
        o = np.empty(k, np.uint32)
        check(self._L.radhip_traversal_results(self._h, q, ptr(s), ptr(a), ptr(o), k, C.byref(n)))
        return s, a, o

    def result_hashes(self, first: int = 0, count: Optional[int] = None) -> np.ndarray:
        """order-sensitive 64-bit hash of the scored lists of traversals [first, first+count), formed on the device"""
        count = self.n_active - first if count is None else count
        out = np.zeros(count, np.uint64)
        check(self._L.radhip_traversal_result_hashes(self._h, first, count, ptr(out)))
        return out

    def pop_log(self, q: int):
        n = C.c_uint64(0)
        check(self._L.radhip_traversal_pop_log(self._h, q, None, None, 0, C.byref(n)))
        k = n.value
        nodes = np.empty(k, np.uint32)
        levels = np.empty(k, np.uint8)
        check(self._L.radhip_traversal_pop_log(self._h, q, ptr(nodes), ptr(levels), k, C.byref(n)))
        return nodes, levels

    def set_targets(self, targets) -> None:
        """Per-traversal stop targets (clamped to n_to_score); raising one resumes a parked traversal."""
        t = np.ascontiguousarray(targets, np.uint64)
        if t.shape[0] != self.nq:
            raise ValueError(f"need {self.nq} targets")
        check(self._L.radhip_traversal_set_targets(self._h, ptr(t)))

    def frontier(self):
        """(best queue key, scored count) per traversal after the last run(); key == 2**64-1
        means the queue is empty; key >> 38 is the 24-bit distance of the best candidate."""
        k = np.empty(self.nq, np.uint64)
        n = np.empty(self.nq, np.uint64)
        check(self._L.radhip_traversal_frontier(self._h, ptr(k), ptr(n)))
        return k, n

    def kernel_time(self):
        ms = C.c_double(0)
        n = C.c_uint64(0)
        check(self._L.radhip_traversal_kernel_time(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def state_bytes(self) -> int:
        return int(self._L.radhip_traversal_state_bytes(self._h))

    @property
    def table(self) -> str:
        """'bucket' (16-B buckets of four entries, one request per probe: the four-per-wavefront kernel's default),
        'local' (the bucket table with a node's home bucket taken from its graph-locality layout id: the ids of a layout block share
        a 128-B line), 'grouped' (2 bits per node, keyed by the layout) or 'hash' (one 8-byte entry per probe)."""
        return {1: "grouped", 2: "bucket", 3: "local"}.get(int(self._L.radhip_traversal_table(self._h)), "hash")

    @property
    def kernel(self) -> str:
        """The kernel this batch was bound to: four traversals per wavefront for batches larger than
        5/4 of a resident round of the one-per-wavefront kernel (rows <= 16 wide), else one per wavefront."""
        return "trav4_kernel" if int(self._L.radhip_traversal_kernel(self._h)) == 4 else "trav_kernel"


class DeviceShard:
    """One rank of the row-sharded traversal (radhip_shard_t): its rows of the corpus, the whole graph, its
    nq traversals of the batch.  `queries_all` is [world * nq, row_bytes], rank-major, the same on every rank."""

    def __init__(self, index: DeviceIndex, rank: int, world: int, row_first: int, row_count: int,
                 queries_all: np.ndarray, n_to_score: int, log_pops: bool = False, own_stream: bool = False):
        self._L = _lib.lib()
        self.index = index
        q = _lib.as_rows(queries_all, index.row_bytes, "queries_all")
        if q.shape[0] % world:
            raise ValueError("queries_all must hold world * nq rows")
        self.rank, self.world, self.nq = int(rank), int(world), q.shape[0] // world
        self._h = C.c_void_p()
        flags = (_lib.TRAV_LOG_POPS if log_pops else 0) | (_lib.SHARD_OWN_STREAM if own_stream else 0)
        check(self._L.radhip_shard_create(index._h, rank, world, row_first, row_count, ptr(q), self.nq, int(n_to_score),
                                          flags, C.byref(self._h)))
        self.width = int(self._L.radhip_shard_width(self._h))
        self.slots = int(self._L.radhip_shard_slots(self._h))     # < nq: slots take the traversals of the batch one after the other
        self.engine = {0: "thread", 1: "wave", 2: "row"}[int(self._L.radhip_shard_engine(self._h))]

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.radhip_shard_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self, queries_all: np.ndarray) -> None:
        q = _lib.as_rows(queries_all, self.index.row_bytes, "queries_all")
        if q.shape[0] != self.world * self.nq:
            raise ValueError(f"reset needs exactly {self.world * self.nq} queries")
        check(self._L.radhip_shard_reset(self._h, ptr(q)))

    # -- the product loop (RCCL on device buffers)
    def run(self, comm: "RcclComm", max_steps: int = 0) -> int:
        steps = C.c_uint64(0)
        check(self._L.radhip_shard_run(self._h, comm._h, max_steps, C.byref(steps)))
        return steps.value

    def run_pair(self, comm: "RcclComm", other: "DeviceShard", other_comm: "RcclComm", max_steps: int = 0) -> int:
        """The product loop for two groups of traversals at once (self on the index's stream, `other` created
        with own_stream=True, a communicator each): one group's step kernel overlaps the other's collectives."""
        steps = C.c_uint64(0)
        check(self._L.radhip_shard_run_pair(self._h, comm._h, other._h, other_comm._h, max_steps, C.byref(steps)))
        return steps.value

    def speculation(self):
        """(depth, speculative scores requested, of which used, expansions finished from them)"""
        d = C.c_uint32(0)
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        check(self._L.radhip_shard_speculation(self._h, C.byref(d), C.byref(a), C.byref(b), C.byref(c)))
        return d.value, a.value, b.value, c.value

    # -- host-staged pieces (the `local` engine of rad_amd.sharded.RowShardedTraversal)
    def step(self, scores_in: np.ndarray):
        s = np.ascontiguousarray(scores_in, np.uint32).reshape(self.nq, self.width)
        check(self._L.radhip_shard_set_scores_in(self._h, ptr(s)))
        live = C.c_uint32(0)
        check(self._L.radhip_shard_step(self._h, C.byref(live)))
        req = np.empty((self.nq, self.width), np.uint32)
        check(self._L.radhip_shard_get_requests(self._h, ptr(req)))
        return req, live.value

    def evaluate(self, requests_all: np.ndarray) -> np.ndarray:
        r = np.ascontiguousarray(requests_all, np.uint32).reshape(self.world, self.nq, self.width)
        check(self._L.radhip_shard_set_requests_all(self._h, ptr(r)))
        check(self._L.radhip_shard_evaluate(self._h))
        out = np.empty((self.world, self.nq, self.width), np.uint32)
        check(self._L.radhip_shard_get_scores_out(self._h, ptr(out)))
        return out

    def stats(self) -> TraversalStats:
        arr = (_lib.TravStats * self.nq)()
        check(self._L.radhip_shard_stats(self._h, arr))
        rec = np.frombuffer(arr, dtype=_TRAV_STATS_DTYPE, count=self.nq)
        return TraversalStats(rec["n_scored"].astype(np.int64), rec["n_pops"].astype(np.int64),
                              rec["n_nbr"].astype(np.int64), rec["status"].astype(np.int32))

    def results(self, q: int):
        n = C.c_uint64(0)
        check(self._L.radhip_shard_results(self._h, q, None, None, None, 0, C.byref(n)))
        k = n.value
        s = np.empty(k, np.uint32); a = np.empty(k, np.uint32); o = np.empty(k, np.uint32)
        check(self._L.radhip_shard_results(self._h, q, ptr(s), ptr(a), ptr(o), k, C.byref(n)))
        return s, a, o

    def pop_log(self, q: int):
        n = C.c_uint64(0)
        check(self._L.radhip_shard_pop_log(self._h, q, None, None, 0, C.byref(n)))
        k = n.value
        nodes = np.empty(k, np.uint32); levels = np.empty(k, np.uint8)
        check(self._L.radhip_shard_pop_log(self._h, q, ptr(nodes), ptr(levels), k, C.byref(n)))
        return nodes, levels

    def timing(self):
        """(step-kernel ms [whole loop ms after run()], evaluation-kernel ms, steps, exchanged bytes)"""
        a, b = C.c_double(0), C.c_double(0)
        n, x = C.c_uint64(0), C.c_uint64(0)
        check(self._L.radhip_shard_timing(self._h, C.byref(a), C.byref(b), C.byref(n), C.byref(x)))
        return a.value, b.value, n.value, x.value

    def state_bytes(self) -> int:
        return int(self._L.radhip_shard_state_bytes(self._h))


class RcclComm:
    """RCCL communicator of the C ABI (one process per GPU).  `unique_id()` is called on rank 0;
    the 128 bytes reach the other ranks through whatever channel the host program has."""

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        check(_lib.lib().radhip_comm_unique_id(buf))
        return bytes(buf)

    def __init__(self, rank: int, world: int, unique_id: bytes, device: int):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        idb = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        check(self._L.radhip_comm_create(rank, world, idb, device, C.byref(self._h)))

    def info(self) -> dict:
        """What the communicator really spans, from RCCL itself: version, ncclCommCount, ncclCommUserRank, and
        the PCI bus id of the device this rank drives."""
        out = _lib.CommInfo()
        check(self._L.radhip_comm_info(self._h, C.byref(out)))
        v = int(out.rccl_version)
        return {"rccl_version": f"{v // 10000}.{(v // 100) % 100}.{v % 100}", "rccl_version_code": v, "comm_count": int(out.comm_count),
                "comm_rank": int(out.comm_rank), "device": int(out.device), "pci_bus_id": out.pci_bus_id.decode("ascii", "replace")}

    def allgather_u64(self, local: np.ndarray) -> np.ndarray:
        """[count] u64 per rank -> [world, count]"""
        a = np.ascontiguousarray(local, np.uint64).reshape(-1)
        out = np.empty((self.world, a.shape[0]), np.uint64)
        check(self._L.radhip_comm_allgather_u64(self._h, ptr(a), a.shape[0], ptr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.radhip_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


__all__ = ["RcclComm", "DeviceShard", "DeviceIndex", "DeviceTraversal", "TraversalStats", "distance_f32", "NO_SLOT", "RadHipError"]
