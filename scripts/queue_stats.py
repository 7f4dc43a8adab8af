#!/usr/bin/env python3
"""Queue maintenance counts of the traversal kernel per pop (re-pivots, flushes, re-mids).
    python scripts/queue_stats.py [n_rows] [corpus_mode] [nq] [n_to_score]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
nts = int(sys.argv[4]) if len(sys.argv) > 4 else 100_000
idx = DeviceIndex(1024, 8, 16, 64); idx.synth_vectors(n, seed=20260101, mode=mode)
X = idx.read_vectors(0, n); idx.close()
idx = DeviceIndex(1024, 8, 16, 64)
for f in range(0, n, 5_000_000):
    idx.add_rows(X[f:f + 5_000_000], seed=777, max_batch=16384)
Q = X[np.random.default_rng(0).integers(0, n, nq)].copy()
t = DeviceTraversal(idx, Q, nts); t.run(); st = t.stats()
p = st.n_pops.astype(float)
print(f"mode {mode} n {n}: pops {p.mean():.0f}, scored/pop {st.n_scored.mean() / p.mean():.2f}, nbr/pop {st.n_nbr.mean() / p.mean():.2f}, "
      f"pops per re-pivot {p.sum() / st.n_repivot.sum():.1f}, per flush {p.sum() / st.n_flush.sum():.1f}, per re-mid {p.sum() / max(1, st.n_remid.sum()):.1f}")
