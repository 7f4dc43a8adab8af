#!/usr/bin/env python3
"""Experiment (GPU box): the statistics of scripts/locality_sim.py on the real bench graph
(100M rows built on the GPU), traversals by the oracle on the host copy.

    python tests/tools/locality_gpu.py [n_rows] [n_to_score] [n_queries]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rad_oracle as O   # noqa: E402
from rad_amd.device import DeviceIndex   # noqa: E402
from scripts.locality_sim import load_helper   # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    n_to_score = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    M, cap0 = 8, 16
    O.build()
    H = load_helper()
    t0 = time.time()
    idx = DeviceIndex(1024, M, cap0, 64, device=0)
    idx.synth_vectors(n, seed=20260101, mode=1)
    X = np.empty((n, 128), np.uint8)
    for f in range(0, n, 4_000_000):
        c = min(4_000_000, n - f)
        X[f:f + c] = idx.read_vectors(f, c)
    idx.close()
    idx = DeviceIndex(1024, M, cap0, 64, device=0)
    for f in range(0, n, 5_000_000):
        idx.add_rows(X[f:f + 5_000_000], seed=777, max_batch=16384)
    print(f"built {n} rows in {time.time() - t0:.0f} s", flush=True)
    levels, adj0, upper_row, adjU = idx.read_graph()
    info = idx.info()
    idx.close()
    g = O.Graph(n, cap0, M, int(info.max_level), int(info.entry), levels, adj0, upper_row, adjU)
    adj0 = np.ascontiguousarray(adj0, np.uint32).reshape(n, cap0)
    rng = np.random.default_rng(5)
    qs = rng.integers(0, n, nq)
    pops0, scored = [], []
    t1 = time.time()
    for qi in qs:
        r = O.rad_traverse(g, X, X[qi], n_to_score)
        pops0.append(r.pop_nodes[r.pop_levels == 0].astype(np.uint32))
        scored.append(r.slots.astype(np.uint32))
    print(f"{nq} oracle traversals in {time.time() - t1:.1f} s: {np.mean([len(p) for p in pops0]):.0f} level-0 pops, "
          f"{np.mean([len(s) for s in scored]):.0f} scored each", flush=True)
    del X

    nc = n // 32
    orders = {"identity": np.arange(n, dtype=np.uint32)}
    ext = np.arange(n, dtype=np.uint64)
    orders["closed-form cluster (c*32+m)"] = ((ext % nc) * 32 + ext // nc).astype(np.uint32)
    del ext
    for G in (480,):
        lid = np.empty(n, np.uint32)
        t1 = time.time()
        H.block_grow_order(adj0.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint32(cap0), C.c_uint32(G),
                           lid.ctypes.data_as(C.c_void_p))
        orders[f"block-grow G={G} ({time.time() - t1:.1f}s)"] = lid
    lid = np.empty(n, np.uint32)
    t1 = time.time()
    H.bfs_block_order(adj0.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint32(cap0), C.c_uint32(0),
                      lid.ctypes.data_as(C.c_void_p))
    orders[f"plain BFS ({time.time() - t1:.1f}s)"] = lid

    H.rows_distinct_groups.restype = C.c_uint64
    for name, lid in orders.items():
        for G in (128, 256, 480, 1024):
            per_exp, groups = [], []
            for p, s in zip(pops0, scored):
                tot = H.rows_distinct_groups(adj0.ctypes.data_as(C.c_void_p), C.c_uint32(cap0),
                                             lid.ctypes.data_as(C.c_void_p), C.c_uint32(G),
                                             p.ctypes.data_as(C.c_void_p), C.c_uint64(len(p)))
                per_exp.append(tot / max(len(p), 1))
                groups.append(len(np.unique(lid[s] // G)))
            print(f"{name:34s} G={G:5d}: lines/expansion {np.mean(per_exp):5.2f}   groups/traversal mean {np.mean(groups):8.0f} "
                  f"max {np.max(groups):8.0f} (fill {n_to_score / np.mean(groups):5.1f})", flush=True)


if __name__ == "__main__":
    main()
