#!/usr/bin/env python3
"""How unequal are the traversals of one batch?  Four traversals share a wavefront from start to finish, so a
wavefront runs as long as its longest traversal, and a launch as long as its last wavefront.
    python scripts/pop_imbalance.py [n_rows] [nq]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
os.environ.setdefault("RADHIP_TRAV", "4")
idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=20260101, mode=2)
idx.link_resident(seed=777, max_batch=16384)
rng = np.random.default_rng(0)
t = DeviceTraversal(idx, idx.read_vectors(int(rng.integers(0, n - nq)), nq), 100_000)
assert t.run() == 0
ms, nl = t.kernel_time()
st = t.stats()
p = st.n_pops.astype(np.float64)
w = p.reshape(-1, 4)
print(f"{nq} traversals, kernel {ms / nl:.1f} ms; pops per traversal: mean {p.mean():.0f} std {p.std():.0f} ({p.std() / p.mean():.3f}) min {p.min():.0f} "
      f"p10 {np.percentile(p, 10):.0f} p90 {np.percentile(p, 90):.0f} max {p.max():.0f}")
print(f"per wavefront (4 consecutive traversals): mean of max {w.max(1).mean():.0f} = {w.max(1).mean() / p.mean():.3f} x the mean traversal "
      f"-> row-slots idle inside wavefronts: {1 - p.mean() / w.max(1).mean():.3f}")
ws = np.sort(w.reshape(-1, 4), axis=None)
srt = np.sort(p).reshape(-1, 4)
print(f"if the four of a wavefront were equally long (sorted by length): idle {1 - p.mean() / srt.max(1).mean():.3f}")
print(f"longest wavefront {w.max(1).max():.0f} pops = {w.max(1).max() / w.max(1).mean():.2f} x the mean wavefront")

# ---- can the length of a traversal be predicted from its first pops?  (longest-first scheduling needs an estimate)
for probe in (500, 1500, 4000):
    t.reset(idx.read_vectors(int(np.random.default_rng(0).integers(0, n - nq)), nq))
    t.run(max_pops=probe)
    s1 = t.stats()
    rate = s1.n_scored.astype(np.float64) / np.maximum(s1.n_pops, 1)
    pred = 100_000 / np.maximum(rate, 1e-9)
    t.run()
    fin = t.stats().n_pops.astype(np.float64)
    c = np.corrcoef(pred, fin)[0, 1]
    order = np.argsort(-pred)
    # rank quality: how many of the true longest 10 % are among the predicted longest 20 %
    top_true = set(np.argsort(-fin)[: nq // 10].tolist())
    hit = len(top_true & set(order[: nq // 5].tolist())) / len(top_true)
    print(f"first {probe} pops: correlation of predicted and final pops {c:.3f}; of the longest 10 % of the traversals {hit:.2f} are among the predicted longest 20 %")
