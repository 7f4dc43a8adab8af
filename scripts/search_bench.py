"""Throughput of Index.search (search_kernel) for large query batches on a GPU-built graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.index import Index
from rad_amd.device import DeviceIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
src = DeviceIndex(1024, 8, 16, 64)
src.synth_vectors(n, seed=1, mode=1)
X = np.concatenate([src.read_vectors(f, min(1_000_000, n - f)) for f in range(0, n, 1_000_000)])
src.close()
hnsw = Index(ndim=1024, connectivity=8, expansion_add=64, max_batch=16384)
t0 = time.time(); hnsw.add(np.arange(n), X); tb = time.time() - t0
print(f"built {n} in {tb:.1f} s ({n / tb / 1e6:.2f} M inserts/s), max_level {hnsw.max_level}", flush=True)
rng = np.random.default_rng(0)
for nq in (4096, 65536, 262144):
    Q = X[rng.integers(0, n, nq)]
    for ef in (64, 400):
        hnsw.search(Q[:256], count=10, expansion=ef)
        t0 = time.time(); m = hnsw.search(Q, count=10, expansion=ef); dt = time.time() - t0
        print(f"nq {nq} ef {ef}: {dt * 1e3:.1f} ms wall, {nq / dt / 1e6:.2f} M queries/s, {m.computed_distances / dt / 1e9:.2f} G eval/s, "
              f"{m.visited_members / dt / 1e6:.0f} M expansions/s, evals/query {m.computed_distances / nq:.0f}", flush=True)
