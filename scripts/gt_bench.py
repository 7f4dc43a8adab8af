#!/usr/bin/env python3
"""Measure the traversal kernel with the per-slot hash table and with the grouped table on one built graph.
    python scripts/gt_bench.py [n_rows] [corpus_mode] [nq] [n_to_score] [ef_add]
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
n_to_score = int(sys.argv[4]) if len(sys.argv) > 4 else 100_000
ef = int(sys.argv[5]) if len(sys.argv) > 5 else 64
src = DeviceIndex(1024, 8, 16, ef)
t0 = time.time()
src.synth_vectors(n, seed=20260101, mode=mode)
print(f"synth {time.time() - t0:.1f}s", flush=True)
X = np.empty((n, 128), np.uint8)
for f in range(0, n, 4_000_000):
    c = min(4_000_000, n - f); X[f:f + c] = src.read_vectors(f, c)
src.close()
idx = DeviceIndex(1024, 8, 16, ef)
t0 = time.time()
for f in range(0, n, 5_000_000):
    idx.add_rows(X[f:f + 5_000_000], seed=777, max_batch=16384)
print(f"build {time.time() - t0:.1f}s max_level {idx.info().max_level}", flush=True)
rng = np.random.default_rng(0)
Q = X[rng.integers(0, n, nq)]
# recall of the graph
from rad_amd import _lib
from rad_amd._lib import ptr, check
import ctypes as C
k = 10
s = np.full((64, k), 0xFFFFFFFF, np.uint32); a = np.zeros((64, k), np.uint32); o = np.zeros((64, k), np.uint32)
cnt = np.zeros(64, np.uint32)
check(_lib.lib().radhip_search(idx._h, ptr(Q[:64]), 64, k, 64, ptr(s), ptr(a), ptr(o), ptr(cnt), None, None))
es, ea, eo, ec = idx.topk(Q[:64], k)
print("recall@10 (ef 64):", np.mean([len(set(s[i]) & set(es[i])) / k for i in range(64)]), flush=True)
del X
for table in (os.environ.get("GT_TABLES", "hash,group").split(",")):
    os.environ["RADHIP_TABLE"] = table
    os.environ["RADHIP_TRAV"] = "4"
    if table == "group":
        info = idx.optimize_layout()
        print(f"layout: {info.seconds:.1f}s groups/row {info.groups_per_row:.2f} degree {info.degree:.2f} id_limit {info.id_limit}", flush=True)
    t = DeviceTraversal(idx, Q, n_to_score)
    for rep in range(3):
        if rep:
            t.reset(Q)
        t.run()
        ms, _ = t.kernel_time(); st = t.stats()
        print(f"{table:6s} table={t.table} rep {rep}: {ms:.1f} ms, {st.n_pops.sum() / ms / 1e6:.3f} G expansions/s, {st.n_scored.sum() / ms / 1e6:.2f} G eval/s, "
              f"{st.n_scored.sum() / st.n_pops.sum():.2f} evals/exp, nbr/exp {st.n_nbr.sum() / st.n_pops.sum():.2f}, "
              f"alg GB/s {(st.n_scored.sum() * 132 + st.n_pops.sum() * 4) / ms / 1e6:.0f} state {t.state_bytes() / 2**30:.1f} GiB; "
              f"per traversal: pops {st.n_pops.mean():.0f} repivots {st.n_repivot.mean():.0f} flushes {st.n_flush.mean():.0f} remids {st.n_remid.mean():.0f}", flush=True)
    t.close()
