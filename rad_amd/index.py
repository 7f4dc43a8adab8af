"""`Index` — the duck-typed index object RAD hands to its HNSW service.

Stands in for `usearch.index.Index` of the reference's un-vendored usearch fork
(.gitmodules:1-3) for every attribute RAD touches:

    Index(ndim=1024, dtype='b1', metric='tanimoto', connectivity=8, expansion_add=400)
                                              README.md:47-53, scripts/start_hnsw_server.py:44-50
    add(keys, vectors, log=...)               README.md:58, examples/DUDEZ_example.ipynb:192
    get_neighbors(node_id, level)             rad/hnsw_service.py:222, rad/hnsw_server.py:483
    get_top_level_nodes()                     rad/hnsw_service.py:229, rad/hnsw_server.py:196
    get_node_ids_from_keys(keys)              examples/DUDEZ_example.ipynb:408
    len(), max_level, connectivity, dtype, ndim, capacity, memory_usage, multi, levels_stats
                                              rad/hnsw_service.py:400-412, rad/hnsw_server.py:148-161
    Index(path=..., view=True, exclude_vectors=True)    scripts/start_hnsw_server.py:69
    save(path) / load(path) / Index.restore(path)

Graph construction, search and Tanimoto evaluation run in librad_hip (HIP, gfx950); adjacency
reads are served from the library's host mirror of the graph, so an Index loaded with
`exclude_vectors=True` (or used in a forked process that never computes) needs no GPU.
The on-disk format is this build's own (.npz); usearch's binary format is not in the
reference tree and is not read.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional, Sequence

import numpy as np

from . import _lib
from ._lib import NO_SLOT, check, ptr
from .device import DeviceIndex, distance_f32


@dataclass
class LevelStats:
    nodes: int
    edges: int
    max_edges: int
    allocated_bytes: int


@dataclass
class Matches:
    keys: np.ndarray
    distances: np.ndarray
    counts: np.ndarray
    slots: np.ndarray
    visited_members: int = 0
    computed_distances: int = 0

    def __len__(self):
        return self.keys.shape[0]


class Index:
    def __init__(self, ndim: Optional[int] = None, dtype: str = "b1", metric: str = "tanimoto",
                 connectivity: int = 16, expansion_add: int = 128, expansion_search: int = 64,
                 connectivity_base: Optional[int] = None, multi: bool = False,
                 path: Optional[str] = None, view: bool = False, exclude_vectors: bool = False,
                 device: int = 0, seed: int = 0, max_batch: int = 4096, **kwargs):
        if path is not None and ndim is None:
            meta = self._peek(path)
            ndim, connectivity = int(meta["ndim"]), int(meta["connectivity"])
            connectivity_base = int(meta["connectivity_base"])
            expansion_add, expansion_search = int(meta["expansion_add"]), int(meta["expansion_search"])
            seed = int(meta["seed"])
        if ndim is None:
            raise ValueError("Index needs ndim (or a path to load)")
        if str(dtype) not in ("b1", "ScalarKind.B1", "b1x8"):
            raise ValueError("rad_amd.Index supports dtype='b1' (packed binary fingerprints) only")
        if str(metric).lower() not in ("tanimoto", "metrickind.tanimoto", "jaccard"):
            raise ValueError("rad_amd.Index supports metric='tanimoto' only")
        if multi:
            raise ValueError("multi=True is not supported")
        self._ndim = int(ndim)
        self._M = int(connectivity)
        self._cap0 = int(connectivity_base) if connectivity_base else 2 * self._M
        self.expansion_add = int(expansion_add)
        self.expansion_search = int(expansion_search)
        self._seed = int(seed)
        self._max_batch = int(max_batch)
        self._device = int(device)
        self._dev = DeviceIndex(self._ndim, self._M, self._cap0, self.expansion_add, device)
        # the key <-> slot map lives in the library (radhip_index_set_keys / radhip_slots_from_keys)
        self._exclude_vectors = False
        if path is not None:
            self.load(path, exclude_vectors=exclude_vectors)

    # ------------------------------------------------------------ properties
    @property
    def ndim(self):
        return self._ndim

    @property
    def dtype(self):
        return "b1"

    @property
    def metric(self):
        return "tanimoto"

    @property
    def connectivity(self):
        return self._M

    @property
    def connectivity_base(self):
        return self._cap0

    @property
    def multi(self):
        return False

    @property
    def max_level(self):
        return max(0, int(self._dev.info().max_level))

    @property
    def size(self):
        return len(self)

    @property
    def capacity(self):
        return len(self)

    @property
    def memory_usage(self):
        inf = self._dev.info()
        n = inf.n
        host = n * (1 + 4 + 4 * inf.connectivity_base) + inf.n_upper_rows * inf.connectivity * 4 + 8 * n
        return int(inf.device_bytes + host)

    @property
    def keys(self):
        n = len(self)
        out = np.empty(n, np.uint64)
        check(_lib.lib().radhip_index_read_keys(self._dev._h, 0, n, ptr(out)))
        return out

    def __len__(self):
        return int(self._dev.info().n)     # the library's row count: never diverges from the device

    @property
    def levels_stats(self):
        levels, adj0, upper_row, adjU = self._dev.read_graph()
        out = []
        for l in range(self.max_level + 1):
            members = levels >= l
            if l == 0:
                edges = int((adj0 != NO_SLOT).sum())
                cap = self._cap0
            else:
                rows = upper_row[members].astype(np.int64) + (l - 1)
                edges = int((adjU[rows] != NO_SLOT).sum())
                cap = self._M
            nodes = int(members.sum())
            out.append(LevelStats(nodes, edges, nodes * cap, nodes * cap * 4))
        return out

    def device_index(self) -> DeviceIndex:
        return self._dev

    def keys_of(self, slots) -> np.ndarray:
        sl = np.ascontiguousarray(slots, dtype=np.uint32)
        out = np.empty(sl.shape, np.uint64)
        check(_lib.lib().radhip_keys_from_slots(self._dev._h, ptr(sl), sl.size, ptr(out)))
        return out

    def _set_keys(self, first: int, keys: np.ndarray) -> None:
        keys = np.ascontiguousarray(keys, np.uint64)
        check(_lib.lib().radhip_index_set_keys(self._dev._h, first, ptr(keys), keys.shape[0]))

    # ------------------------------------------------------------ build
    def add(self, keys, vectors, log: bool = False, threads: int = 0, copy: bool = True, **kwargs):
        """Append vectors (np.packbits rows) under integer keys and link them into the graph
        on the GPU (radhip_index_add)."""
        vectors = _lib.as_rows(vectors, self._dev.row_bytes)
        n = vectors.shape[0]
        if keys is None:
            keys = np.arange(len(self), len(self) + n, dtype=np.uint64)
        keys = np.atleast_1d(np.asarray(keys)).astype(np.uint64)
        if keys.shape[0] != n:
            raise ValueError(f"{keys.shape[0]} keys for {n} vectors")
        if self._exclude_vectors:
            raise RuntimeError("index was loaded with exclude_vectors=True; it cannot be extended")
        first = len(self)
        check(_lib.lib().radhip_index_add(self._dev._h, ptr(vectors), n, self._seed, self._max_batch))
        self._set_keys(first, keys)
        return keys

    def search(self, vectors, count: int = 10, expansion: Optional[int] = None, exact: bool = False, **kwargs) -> Matches:
        """k nearest by best-first graph search (radhip_search), or exactly by the scan kernel."""
        q = _lib.as_rows(vectors, self._dev.row_bytes, "queries")
        nq = q.shape[0]
        k = int(min(count, len(self))) if len(self) else 0
        if k == 0:
            z = np.zeros((nq, 0))
            return Matches(z.astype(np.uint64), z.astype(np.float32), np.zeros(nq, np.uint32), z.astype(np.uint32))
        if exact:
            # brute force on the device, reduced to the k best on the chip (radhip_tanimoto_topk); larger k
            # than the kernel keeps in LDS falls back to the full scan + a host sort
            if k <= 1984:
                slots, aa, oo, counts = self._dev.topk(q, k, 0, len(self))
            else:
                a, o = self._dev.scan(q, 0, len(self))
                qk = ((o.astype(np.int64) - a.astype(np.int64)) << 23) // np.maximum(o.astype(np.int64), 1)
                order = np.lexsort((np.broadcast_to(np.arange(a.shape[1]), a.shape), qk), axis=1)[:, :k]
                slots = order.astype(np.uint32)
                aa = np.take_along_axis(a, order, 1)
                oo = np.take_along_axis(o, order, 1)
                counts = np.full(nq, k, np.uint32)
            keys = np.zeros(slots.shape, np.uint64)
            ok = slots != NO_SLOT
            keys[ok] = self.keys_of(slots[ok])
            return Matches(keys, distance_f32(aa, oo), counts, slots)
        ef = max(int(expansion or self.expansion_search), k)
        slots = np.full((nq, k), NO_SLOT, np.uint32)
        a = np.zeros((nq, k), np.uint32)
        o = np.zeros((nq, k), np.uint32)
        counts = np.zeros(nq, np.uint32)
        ev = np.zeros(nq, np.uint64)
        pp = np.zeros(nq, np.uint64)
        check(_lib.lib().radhip_search(self._dev._h, ptr(q), nq, k, ef, ptr(slots), ptr(a), ptr(o),
                                       ptr(counts), ptr(ev), ptr(pp)))
        keys = np.zeros((nq, k), np.uint64)
        valid = np.arange(k)[None, :] < counts[:, None]      # rows are padded with NO_SLOT past counts
        keys[valid] = self.keys_of(slots[valid])
        return Matches(keys, distance_f32(a, o), counts, slots, int(pp.sum()), int(ev.sum()))

    # ------------------------------------------------------------ adjacency reads (RAD's hot calls)
    def get_neighbors(self, node_id: int, level: int) -> np.ndarray:
        """Flat [neighbor_slot, neighbor_key, ...] of one node on one level; raises if the node
        does not exist on that level."""
        out = np.empty(128, np.uint64)
        n = C.c_uint32(0)
        check(_lib.lib().radhip_get_neighbors_keyed(self._dev._h, int(node_id), int(level), ptr(out), 64, C.byref(n)))
        return out[: 2 * n.value].copy()

    def get_top_level_nodes(self) -> np.ndarray:
        slots = self._dev.get_top_level_nodes()
        out = np.empty(2 * slots.shape[0], np.uint64)
        out[0::2] = slots
        out[1::2] = self.keys_of(slots)
        return out

    def get_node_ids_from_keys(self, keys) -> np.ndarray:
        """key -> slot (radhip_slots_from_keys: binary search in the library, no Python dict);
        an unknown key raises KeyError, as a dict lookup would."""
        k = np.ascontiguousarray(np.atleast_1d(keys), dtype=np.uint64)
        out = np.empty(k.shape[0], np.uint32)
        missing = C.c_uint64(0)
        check(_lib.lib().radhip_slots_from_keys(self._dev._h, ptr(k), k.shape[0], ptr(out), C.byref(missing)))
        if missing.value:
            raise KeyError(int(k[out == NO_SLOT][0]))
        return out.astype(np.uint64)

    # ------------------------------------------------------------ external graphs
    def load_graph(self, keys, vectors, levels, adj0, upper_row, adjU, max_level: int, entry: int):
        """Install an externally built layered graph (and, optionally, its vectors)."""
        levels = np.asarray(levels)
        keys = np.arange(levels.shape[0], dtype=np.uint64) if keys is None else np.asarray(keys).astype(np.uint64)
        if vectors is not None:
            self._dev.load_vectors(vectors)
        self._dev.load_graph(levels, adj0, upper_row, adjU, max_level, entry)
        self._set_keys(0, keys)
        self._exclude_vectors = vectors is None

    # ------------------------------------------------------------ persistence
    @staticmethod
    def _peek(path):
        with np.load(path, allow_pickle=False) as z:
            return {k: z[k] for k in ("ndim", "connectivity", "connectivity_base", "expansion_add",
                                      "expansion_search", "seed")}

    def save(self, path: str) -> None:
        levels, adj0, upper_row, adjU = self._dev.read_graph()
        inf = self._dev.info()
        vec = self._dev.read_vectors(0, len(self)) if (inf.has_vectors and not self._exclude_vectors) else np.empty((0, self._dev.row_bytes), np.uint8)
        with open(path, "wb") as f:
            np.savez(f, format=np.bytes_(b"rad_amd.index.v1"), ndim=self._ndim, connectivity=self._M,
                     connectivity_base=self._cap0, expansion_add=self.expansion_add,
                     expansion_search=self.expansion_search, seed=self._seed, max_level=inf.max_level,
                     entry=inf.entry, keys=self.keys, levels=levels, adj0=adj0, upper_row=upper_row,
                     adjU=adjU, vectors=vec)

    def load(self, path: str, exclude_vectors: bool = False) -> None:
        with np.load(path, allow_pickle=False) as z:
            if bytes(z["format"]) != b"rad_amd.index.v1":
                raise ValueError(f"{path} is not a rad_amd index file (usearch's binary format is not supported)")
            if int(z["ndim"]) != self._ndim or int(z["connectivity"]) != self._M or int(z["connectivity_base"]) != self._cap0:
                raise ValueError("index file parameters do not match this Index")
            vec = None if (exclude_vectors or z["vectors"].shape[0] == 0) else z["vectors"]
            self.load_graph(z["keys"], vec, z["levels"], z["adj0"], z["upper_row"], z["adjU"],
                            int(z["max_level"]), int(z["entry"]))

    view = load

    @classmethod
    def restore(cls, path: str, view: bool = False, exclude_vectors: bool = False, **kwargs) -> "Index":
        return cls(path=path, view=view, exclude_vectors=exclude_vectors, **kwargs)
