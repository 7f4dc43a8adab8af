"""ctypes wrapper over oracle/_build/librad_oracle.so — CPU ORACLE.

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py; never by anything under rad_amd/.
See rad_oracle.h for the reference file:line each function restates.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "librad_oracle.so")
NO_SLOT = 0xFFFFFFFF


def build_native() -> str:
    """-O3 -march=native build for THIS host (the cpu_baseline leg of bench.py times it); the portable
    build stays what the parity tests load."""
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])
    return os.path.join(_HERE, "_build", "librad_oracle_native.so")


def use_library(path: str) -> None:
    """Point the wrapper at another build of the same source (before the first call)."""
    global _SO, _lib
    _SO, _lib = path, None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "rad_oracle.c")
    hdr = os.path.join(_HERE, "rad_oracle.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class _Graph(C.Structure):
    _fields_ = [("n", C.c_uint64), ("cap0", C.c_uint32), ("capU", C.c_uint32),
                ("max_level", C.c_int32), ("entry", C.c_uint32),
                ("levels", C.c_void_p), ("adj0", C.c_void_p),
                ("upper_row", C.c_void_p), ("adjU", C.c_void_p),
                ("n_upper_rows", C.c_uint64)]


class _Stats(C.Structure):
    _fields_ = [("n_scored", C.c_uint64), ("n_pops", C.c_uint64),
                ("n_evals", C.c_uint64), ("n_nbr", C.c_uint64),
                ("f_valid", C.c_uint32), ("f_and", C.c_uint32), ("f_or", C.c_uint32),
                ("f_slot", C.c_uint32), ("f_level", C.c_uint32), ("f_pad", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if _SO.endswith("librad_oracle.so"):
            build()
        L = C.CDLL(_SO)
        L.orc_distance_f32.restype = C.c_float
        L.orc_distance_f32.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_graph_top_level.restype = C.c_uint64
        L.orc_hnsw_create.restype = C.c_void_p
        L.orc_hnsw_create.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64]
        L.orc_hnsw_destroy.argtypes = [C.c_void_p]
        L.orc_hnsw_add.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32]
        L.orc_hnsw_size.restype = C.c_uint64
        L.orc_hnsw_size.argtypes = [C.c_void_p]
        L.orc_hnsw_graph.argtypes = [C.c_void_p, C.POINTER(_Graph)]
        L.orc_hnsw_rows.restype = C.c_void_p
        L.orc_hnsw_rows.argtypes = [C.c_void_p]
        L.orc_hnsw_level_of.restype = C.c_int
        L.orc_hnsw_level_of.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32]
        L.orc_hnsw_search.restype = C.c_uint32
        L.orc_graph_search.restype = C.c_uint32
        L.orc_synth_max_level.restype = C.c_int32
        L.orc_synth_max_level.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_synth_upper_rows.restype = C.c_uint64
        L.orc_synth_upper_rows.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_stepper_create.restype = C.c_void_p
        L.orc_stepper_create.argtypes = [C.POINTER(_Graph), C.c_uint64, C.c_uint64]
        L.orc_stepper_destroy.argtypes = [C.c_void_p]
        L.orc_stepper_step.restype = C.c_int
        L.orc_stepper_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_stepper_status.restype = C.c_int
        L.orc_stepper_status.argtypes = [C.c_void_p]
        L.orc_stepper_stats.argtypes = [C.c_void_p, C.POINTER(_Stats)]
        L.orc_stepper_results.restype = C.c_uint64
        L.orc_stepper_results.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        L.orc_stepper_pop_log.restype = C.c_uint64
        L.orc_stepper_pop_log.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class Graph:
    """Host graph store in the layout shared with the product's load_graph."""
    n: int
    cap0: int
    capU: int
    max_level: int
    entry: int
    levels: np.ndarray      # int8 [n]
    adj0: np.ndarray        # uint32 [n, cap0]
    upper_row: np.ndarray   # uint32 [n]
    adjU: np.ndarray        # uint32 [n_upper_rows, capU]

    def c_struct(self) -> _Graph:
        g = _Graph()
        g.n, g.cap0, g.capU = self.n, self.cap0, self.capU
        g.max_level, g.entry = self.max_level, self.entry
        g.levels, g.adj0 = _p(self.levels), _p(self.adj0)
        g.upper_row, g.adjU = _p(self.upper_row), _p(self.adjU)
        g.n_upper_rows = self.adjU.shape[0]
        return g

    def neighbors(self, slot: int, level: int):
        out = np.empty(max(self.cap0, self.capU), np.uint32)
        g = self.c_struct()
        k = lib().orc_graph_neighbors(C.byref(g), C.c_uint32(slot), C.c_int(level), _p(out),
                                      C.c_uint32(out.size))
        if k < 0:
            raise KeyError(f"node {slot} does not exist on level {level}")
        return out[:k].copy()

    def top_level(self):
        out = np.empty(self.n, np.uint32)
        g = self.c_struct()
        k = lib().orc_graph_top_level(C.byref(g), _p(out), C.c_uint64(out.size))
        return out[:k].copy()


def tanimoto_counts(a: np.ndarray, b: np.ndarray):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    ca, co = C.c_uint32(), C.c_uint32()
    lib().orc_tanimoto_counts(_p(a), _p(b), C.c_size_t(a.size), C.byref(ca), C.byref(co))
    return ca.value, co.value


def distance_f32(and_cnt: int, or_cnt: int) -> float:
    return float(lib().orc_distance_f32(and_cnt, or_cnt))


def scan(corpus: np.ndarray, query: np.ndarray):
    corpus = np.ascontiguousarray(corpus, np.uint8)
    query = np.ascontiguousarray(query, np.uint8)
    n, rb = corpus.shape
    a = np.empty(n, np.uint32)
    o = np.empty(n, np.uint32)
    lib().orc_scan(_p(corpus), C.c_uint64(n), C.c_size_t(rb), _p(query), _p(a), _p(o))
    return a, o


def gather(corpus: np.ndarray, query: np.ndarray, slots: np.ndarray):
    corpus = np.ascontiguousarray(corpus, np.uint8)
    query = np.ascontiguousarray(query, np.uint8)
    slots = np.ascontiguousarray(slots, np.uint32)
    a = np.empty(slots.size, np.uint32)
    o = np.empty(slots.size, np.uint32)
    lib().orc_gather(_p(corpus), C.c_size_t(corpus.shape[1]), _p(query), _p(slots),
                     C.c_uint64(slots.size), _p(a), _p(o))
    return a, o


@dataclass
class TraverseResult:
    slots: np.ndarray
    and_cnt: np.ndarray
    or_cnt: np.ndarray
    pop_nodes: np.ndarray
    pop_levels: np.ndarray
    n_pops: int
    n_nbr: int
    frontier: tuple = None   # (and, or, slot, level) of the best item left in the queue, or None


def rad_traverse(graph: Graph, corpus: np.ndarray, query: np.ndarray, n_to_score: int,
                 max_pops: int = 0, log_pops: bool = True) -> TraverseResult:
    corpus = np.ascontiguousarray(corpus, np.uint8)
    query = np.ascontiguousarray(query, np.uint8)
    cap = int(n_to_score) + graph.cap0 + graph.capU + int(graph.n if graph.n < 4096 else 4096) + 64
    s = np.empty(cap, np.uint32)
    a = np.empty(cap, np.uint32)
    o = np.empty(cap, np.uint32)
    pcap = (cap * 2 + 1024) if log_pops else 0
    pn = np.empty(pcap, np.uint32) if log_pops else None
    pl = np.empty(pcap, np.uint8) if log_pops else None
    st = _Stats()
    g = graph.c_struct()
    rc = lib().orc_rad_traverse(C.byref(g), _p(corpus), C.c_size_t(corpus.shape[1]), _p(query),
                                C.c_uint64(n_to_score), C.c_uint64(max_pops), _p(s), _p(a), _p(o),
                                C.c_uint64(cap), _p(pn), _p(pl), C.c_uint64(pcap), C.byref(st))
    if rc:
        raise RuntimeError(f"orc_rad_traverse failed rc={rc}")
    k = st.n_scored
    np_ = min(st.n_pops, pcap)
    return TraverseResult(s[:k].copy(), a[:k].copy(), o[:k].copy(),
                          pn[:np_].copy() if log_pops else np.empty(0, np.uint32),
                          pl[:np_].copy() if log_pops else np.empty(0, np.uint8),
                          int(st.n_pops), int(st.n_nbr),
                          (int(st.f_and), int(st.f_or), int(st.f_slot), int(st.f_level)) if st.f_valid else None)


def rad_traverse_many(graph: Graph, corpus: np.ndarray, queries: np.ndarray, n_to_score: int,
                      n_threads: int, hashes: bool = False):
    """Stats only (cpu_baseline leg): returns (n_scored, n_pops, n_nbr) arrays — and, with hashes=True, the
    order-sensitive 64-bit hash of every traversal's scored list (orc_result_hash)."""
    corpus = np.ascontiguousarray(corpus, np.uint8)
    queries = np.ascontiguousarray(queries, np.uint8)
    nq = queries.shape[0]
    st = (_Stats * nq)()
    g = graph.c_struct()
    hs = np.zeros(nq, np.uint64)
    L = lib()
    L.orc_rad_traverse_many_h.restype = C.c_int
    rc = L.orc_rad_traverse_many_h(C.byref(g), _p(corpus), C.c_size_t(corpus.shape[1]),
                                   _p(queries), C.c_uint32(nq), C.c_uint64(n_to_score),
                                   C.c_int(n_threads), st, _p(hs) if hashes else None)
    if rc:
        raise RuntimeError(f"orc_rad_traverse_many failed rc={rc}")
    out = (np.array([x.n_scored for x in st]), np.array([x.n_pops for x in st]),
           np.array([x.n_nbr for x in st]))
    return out + (hs,) if hashes else out


def result_hash(slots, and_cnt, or_cnt) -> int:
    s = np.ascontiguousarray(slots, np.uint32); a = np.ascontiguousarray(and_cnt, np.uint32); o = np.ascontiguousarray(or_cnt, np.uint32)
    L = lib()
    L.orc_result_hash.restype = C.c_uint64
    return int(L.orc_result_hash(_p(s), _p(a), _p(o), C.c_uint64(s.shape[0])))


class Stepper:
    """orc_stepper: the traversal of rad_traverse cut at the fingerprint read.  step(and, or) applies the
    counts of the slots returned by the previous call and returns the next slots that need a score (an
    empty array once the traversal has ended).  It never touches the corpus."""

    def __init__(self, graph: Graph, n_to_score: int, log_pops: bool = True):
        self._g = graph
        self._gs = graph.c_struct()          # keeps the arrays alive
        self._cap = int(n_to_score) * 2 + 4096 + int(graph.n if graph.n < 65536 else 65536)
        self._h = lib().orc_stepper_create(C.byref(self._gs), C.c_uint64(n_to_score), C.c_uint64(self._cap if log_pops else 0))
        self._req = np.empty(64, np.uint32)
        self.width = max(graph.cap0, graph.capU)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_stepper_destroy(self._h)
            self._h = None

    def step(self, and_in=None, or_in=None) -> np.ndarray:
        a = np.ascontiguousarray(and_in if and_in is not None else np.empty(0), np.uint32)
        o = np.ascontiguousarray(or_in if or_in is not None else np.empty(0), np.uint32)
        n = lib().orc_stepper_step(self._h, _p(a), _p(o), _p(self._req), C.c_uint32(self.width))
        st = self.status
        if st < 0:
            raise RuntimeError(f"orc_stepper_step failed, status {st}")
        return self._req[:n].copy()

    @property
    def status(self) -> int:
        return int(lib().orc_stepper_status(self._h))

    def result(self) -> TraverseResult:
        st = _Stats()
        lib().orc_stepper_stats(self._h, C.byref(st))
        k = int(st.n_scored)
        s = np.empty(k, np.uint32); a = np.empty(k, np.uint32); o = np.empty(k, np.uint32)
        lib().orc_stepper_results(self._h, _p(s), _p(a), _p(o), C.c_uint64(k))
        pn = np.empty(self._cap, np.uint32); pl = np.empty(self._cap, np.uint8)
        m = int(lib().orc_stepper_pop_log(self._h, _p(pn), _p(pl), C.c_uint64(self._cap)))
        return TraverseResult(s, a, o, pn[:m].copy(), pl[:m].copy(), int(st.n_pops), int(st.n_nbr))


class Hnsw:
    """usearch-shaped HNSW builder/searcher (parity unpinned vs usearch)."""

    def __init__(self, ndim_bits, connectivity, connectivity_base=0, expansion_add=128, seed=0):
        self.ndim_bits = ndim_bits
        self.h = lib().orc_hnsw_create(ndim_bits, connectivity, connectivity_base, expansion_add, seed)
        self.row_bytes = (ndim_bits + 7) // 8

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_hnsw_destroy(self.h)
            self.h = None

    def add(self, rows: np.ndarray, max_batch: int = 1):
        rows = np.ascontiguousarray(rows, np.uint8)
        assert rows.shape[1] == self.row_bytes
        lib().orc_hnsw_add(self.h, _p(rows), C.c_uint64(rows.shape[0]), C.c_uint32(max_batch))

    def __len__(self):
        return int(lib().orc_hnsw_size(self.h))

    def graph(self) -> Graph:
        g = _Graph()
        lib().orc_hnsw_graph(self.h, C.byref(g))
        n = g.n

        def arr(ptr, dtype, count):
            if count == 0:
                return np.empty(0, dtype)
            buf = (C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype).copy()
        levels = arr(g.levels, np.int8, n)
        adj0 = arr(g.adj0, np.uint32, n * g.cap0).reshape(n, g.cap0)
        upper_row = arr(g.upper_row, np.uint32, n)
        adjU = arr(g.adjU, np.uint32, g.n_upper_rows * g.capU).reshape(g.n_upper_rows, g.capU)
        return Graph(int(n), int(g.cap0), int(g.capU), int(g.max_level), int(g.entry), levels,
                     adj0, upper_row, adjU)

    def search(self, query, k, ef):
        query = np.ascontiguousarray(query, np.uint8)
        s = np.empty(k, np.uint32)
        a = np.empty(k, np.uint32)
        o = np.empty(k, np.uint32)
        ne, npop = C.c_uint64(), C.c_uint64()
        n = lib().orc_hnsw_search(C.c_void_p(self.h), _p(query), C.c_uint32(k), C.c_uint32(ef),
                                  _p(s), _p(a), _p(o), C.byref(ne), C.byref(npop))
        return s[:n].copy(), a[:n].copy(), o[:n].copy(), ne.value, npop.value


def graph_search(graph: Graph, corpus, query, k, ef):
    corpus = np.ascontiguousarray(corpus, np.uint8)
    query = np.ascontiguousarray(query, np.uint8)
    s = np.empty(k, np.uint32)
    a = np.empty(k, np.uint32)
    o = np.empty(k, np.uint32)
    ne, npop = C.c_uint64(), C.c_uint64()
    g = graph.c_struct()
    n = lib().orc_graph_search(C.byref(g), _p(corpus), C.c_size_t(corpus.shape[1]), _p(query),
                               C.c_uint32(k), C.c_uint32(ef), _p(s), _p(a), _p(o), C.byref(ne),
                               C.byref(npop))
    return s[:n].copy(), a[:n].copy(), o[:n].copy(), ne.value, npop.value


def hnsw_level_of(seed, slot, connectivity):
    return int(lib().orc_hnsw_level_of(seed, slot, connectivity))


def synth_rows(first_row, n_rows, n_total, ndim_bits, seed, mode):
    nbytes = (ndim_bits + 7) // 8
    out = np.empty((n_rows, nbytes), np.uint8)
    lib().orc_synth_rows(_p(out), C.c_uint64(first_row), C.c_uint64(n_rows), C.c_uint64(n_total),
                         C.c_uint32(ndim_bits), C.c_uint64(seed), C.c_int(mode))
    return out


def synth_graph(n, connectivity, cap0, seed) -> Graph:
    L = lib().orc_synth_max_level(n, connectivity)
    nu = lib().orc_synth_upper_rows(n, connectivity)
    levels = np.empty(n, np.int8)
    adj0 = np.empty((n, cap0), np.uint32)
    upper_row = np.empty(n, np.uint32)
    adjU = np.empty((nu, connectivity), np.uint32)
    lib().orc_synth_graph(C.c_uint64(n), C.c_uint32(connectivity), C.c_uint32(cap0),
                          C.c_uint64(seed), _p(levels), _p(adj0), _p(upper_row), _p(adjU))
    return Graph(n, cap0, connectivity, int(L), 0, levels, adj0, upper_row, adjU)
