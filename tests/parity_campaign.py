"""Manual validation campaign (test infrastructure: it uses the oracle; not collected by pytest).
At the bench configuration: N traversals at 100M rows, GPU statistics (scored,
expansions, neighbours seen — any divergence from the sequential semantics changes them) against
the oracle's threaded runner, for both traversal kernels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal
from oracle import rad_oracle as O
O.build()
n, nts = 100_000_000, 100_000
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=20260101, mode=1); idx.synth_graph(seed=777)
X = np.empty((n, 128), np.uint8)
for f in range(0, n, 10_000_000):
    X[f:f + 10_000_000] = idx.read_vectors(f, 10_000_000)
levels, adj0, upper_row, adjU = idx.read_graph()
inf = idx.info()
g = O.Graph(n, 16, 8, int(inf.max_level), int(inf.entry), levels, adj0, upper_row, adjU)
rng = np.random.default_rng(99)
Q = X[rng.integers(0, n, N)].copy()
t0 = time.time(); ws, wp, wn = O.rad_traverse_many(g, X, Q, nts, os.cpu_count() or 16); print(f"oracle: {N} traversals in {time.time() - t0:.1f} s", flush=True)
for k in ("4", "1"):
    os.environ["RADHIP_TRAV"] = k
    t = DeviceTraversal(idx, Q, nts); t.run(); st = t.stats()
    bad = np.flatnonzero((st.n_scored != ws) | (st.n_pops != wp) | (st.n_nbr != wn))
    print(f"{t.kernel}: {N - bad.size} of {N} traversals match the oracle's counters" + (f"; first mismatches {bad[:8]}" if bad.size else ""), flush=True)
    t.close()
    assert bad.size == 0
