#!/usr/bin/env python3
"""recall@10 of the GPU-built HNSW graph vs exact top-k, by corpus mode / size / expansion / batch size.
    python scripts/recall_table.py n mode ef_add max_batch [connectivity]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from rad_amd.device import DeviceIndex
from rad_amd import _lib
from rad_amd._lib import ptr, check
n = int(sys.argv[1]); mode = int(sys.argv[2]); ef_add = int(sys.argv[3]); mb = int(sys.argv[4])
M = int(sys.argv[5]) if len(sys.argv) > 5 else 8
idx = DeviceIndex(1024, M, 2 * M, ef_add)
idx.synth_vectors(n, seed=20260101, mode=mode)
t0 = time.time()
idx.link_resident(seed=777, max_batch=mb)       # the rows are linked where they are: no host copy of the corpus
tb = time.time() - t0
nq, k = 256, 10
Q = np.concatenate([idx.read_vectors(int(r), 1) for r in np.random.default_rng(1).integers(0, n, nq)])
es, ea, eo, ec = idx.topk(Q, k)
out = []
for ef in (64, 128, 400):
    s = np.full((nq, k), 0xFFFFFFFF, np.uint32); a = np.zeros((nq, k), np.uint32); o = np.zeros((nq, k), np.uint32); cnt = np.zeros(nq, np.uint32)
    check(_lib.lib().radhip_search(idx._h, ptr(Q), nq, k, ef, ptr(s), ptr(a), ptr(o), ptr(cnt), None, None))
    out.append(f"ef {ef}: {np.mean([len(set(s[i]) & set(es[i])) / k for i in range(nq)]):.3f}")
deg = (idx.read_graph()[1] != 0xFFFFFFFF).sum(1).mean() if n <= 20_000_000 else float('nan')
print(f"n={n} mode={mode} connectivity={M} ef_add={ef_add} max_batch={mb}: build {tb:.1f}s max_level {idx.info().max_level} degree {deg:.2f} recall@10 " + ", ".join(out), flush=True)
