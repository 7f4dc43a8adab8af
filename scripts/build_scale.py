"""One-off: time Index.add (GPU HNSW build) at larger sizes and traverse the built graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal
from rad_amd.index import Index
n = int(sys.argv[1]); ef = int(sys.argv[2]) if len(sys.argv) > 2 else 64
src = DeviceIndex(1024, 8, 16, 64)
src.synth_vectors(n, seed=20260101, mode=1)
X = np.empty((n, 128), np.uint8)
for f in range(0, n, 4_000_000):
    c = min(4_000_000, n - f); X[f:f + c] = src.read_vectors(f, c)
src.close()
idx = Index(ndim=1024, connectivity=8, expansion_add=ef, max_batch=16384)
t0 = time.time()
step = 5_000_000
for f in range(0, n, step):
    idx.add(np.arange(f, min(n, f + step)), X[f:f + step])
    print(f"  added {min(n, f + step)} in {time.time() - t0:.1f} s", flush=True)
tb = time.time() - t0
dev = idx.device_index()
cap = dev.traversal_capacity()
rng = np.random.default_rng(0)
Q = X[rng.integers(0, n, cap)]
t = DeviceTraversal(dev, Q, 100_000)
t.run()
ms, _ = t.kernel_time(); st = t.stats()
print(f"n={n} ef_add={ef}: build {tb:.1f} s ({n / tb:.0f} inserts/s), max_level {idx.max_level}; {cap} traversals to 100k: {ms:.1f} ms, "
      f"{st.n_pops.sum() / ms / 1e3:.0f} M expansions/s, {st.n_scored.sum() / ms / 1e6:.2f} G eval/s, {st.n_scored.sum() / st.n_pops.sum():.2f} evals/expansion")
m = idx.search(Q[:256], count=10, expansion=128)
ex = idx.search(Q[:32], count=10, exact=True)
print("recall@10 (ef=128) vs exact:", np.mean([len(set(m.slots[i]) & set(ex.slots[i])) / 10 for i in range(32)]))
