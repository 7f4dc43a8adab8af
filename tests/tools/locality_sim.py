#!/usr/bin/env python3
"""Experiment (CPU only): how many 128-B lines would a group-bitmap visited table touch per
expansion, and how many groups does a traversal touch in all, under graph-locality renumberings
of the slots?  Uses the oracle builder/traversal on the bench's synthetic corpus at small n.

    python tests/tools/locality_sim.py [n_rows] [n_to_score] [n_queries]
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rad_oracle as O   # noqa: E402


def load_helper():
    src = os.path.join(ROOT, "scripts", "locsim.c")
    out = os.path.join(ROOT, "scripts", "_locsim.so")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", out, src])
    return C.CDLL(out)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_to_score = int(sys.argv[2]) if len(sys.argv) > 2 else 20_000
    nq = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    graph_file = sys.argv[4] if len(sys.argv) > 4 else None
    mode = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    ef_add = int(sys.argv[6]) if len(sys.argv) > 6 else 64
    M, cap0 = 8, 16
    O.build()
    H = load_helper()
    t0 = time.time()
    X = O.synth_rows(0, n, n, 1024, 20260101, mode)
    if graph_file and os.path.exists(graph_file):
        z = np.load(graph_file)
        g = O.Graph(n, cap0, M, int(z["max_level"]), int(z["entry"]), z["levels"], z["adj0"], z["upper_row"], z["adjU"])
    else:
        h = O.Hnsw(1024, M, cap0, ef_add, seed=777)
        for f in range(0, n, 200_000):
            h.add(X[f:f + 200_000], max_batch=16384)
            print(f"  built {min(f + 200_000, n)} rows, {time.time() - t0:.0f} s", flush=True)
        g0 = h.graph()
        g = O.Graph(n, cap0, M, g0.max_level, g0.entry, g0.levels.copy(), g0.adj0.copy(), g0.upper_row.copy(), g0.adjU.copy())
        if graph_file:
            np.savez(graph_file, levels=g.levels, adj0=g.adj0, upper_row=g.upper_row, adjU=g.adjU,
                     max_level=g.max_level, entry=g.entry)
    adj0 = np.ascontiguousarray(g.adj0, np.uint32).reshape(n, cap0)
    print(f"graph ready: {time.time() - t0:.0f} s, max_level {g.max_level}")
    rng = np.random.default_rng(5)
    qs = rng.integers(0, n, nq)
    pops0, scored = [], []
    for qi in qs:
        r = O.rad_traverse(g, X, X[qi], n_to_score)
        pops0.append(r.pop_nodes[r.pop_levels == 0].astype(np.uint32))
        scored.append(r.slots.astype(np.uint32))
    print(f"traversals: {np.mean([len(p) for p in pops0]):.0f} level-0 pops, {np.mean([len(s) for s in scored]):.0f} scored each")

    # recall@10 of the graph search against brute force
    rec = []
    for qi in qs[:16]:
        a, o = O.scan(X, X[qi])
        qk = ((o.astype(np.int64) - a) << 23) // np.maximum(o, 1)
        truth = np.lexsort((np.arange(n), qk))[:10]
        got = O.graph_search(g, X, X[qi], 10, 64)[0]
        rec.append(len(set(truth.tolist()) & set(np.asarray(got).tolist())) / 10)
    print(f"recall@10 (ef 64): {np.mean(rec):.3f}")
    nc = n // 32
    orders = {}
    orders["identity"] = np.arange(n, dtype=np.uint32)
    ext = np.arange(n, dtype=np.uint64)
    if mode == 1:
        c = ext % nc
        orders["closed-form cluster (c*32+m)"] = (c * 32 + ext // nc).astype(np.uint32)
    elif mode == 2:
        D = 1
        while (1 << (2 * D)) < n:
            D += 1
        leaf = (ext * np.uint64(0x9E3779B1)) & np.uint64((1 << (2 * D)) - 1)
        lid = np.empty(n, np.uint32)
        lid[np.argsort(leaf, kind="stable")] = np.arange(n, dtype=np.uint32)
        orders["closed-form tree order"] = lid
    lid = np.empty(n, np.uint32)
    t1 = time.time()
    H.dfs_order(adj0.ctypes.data_as(C.c_void_p), C.c_uint64(n), C.c_uint32(cap0), C.c_uint32(0),
                lid.ctypes.data_as(C.c_void_p))
    orders[f"DFS nearest-first ({time.time() - t1:.1f}s)"] = lid
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
        assert len(np.unique(lid)) == n
        for G in (128, 256, 480, 1024):
            per_exp, groups = [], []
            for p, s in zip(pops0, scored):
                tot = H.rows_distinct_groups(adj0.ctypes.data_as(C.c_void_p), C.c_uint32(cap0),
                                             lid.ctypes.data_as(C.c_void_p), C.c_uint32(G),
                                             p.ctypes.data_as(C.c_void_p), C.c_uint64(len(p)))
                per_exp.append(tot / max(len(p), 1))
                groups.append(len(np.unique(lid[s] // G)))
            print(f"{name:34s} G={G:5d}: lines/expansion {np.mean(per_exp):5.2f}   groups/traversal {np.mean(groups):8.0f} "
                  f"(fill {n_to_score / np.mean(groups):5.1f})")


if __name__ == "__main__":
    main()
