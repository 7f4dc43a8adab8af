"""debug: trav4 vs oracle on a mid-size closed-form graph with a long traversal (deep queue)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
os.environ["RADHIP_TRAV"] = "4"
from oracle import rad_oracle as O
from rad_amd.device import DeviceIndex, DeviceTraversal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
nts = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
O.build()
idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=5, mode=mode)
idx.synth_graph(seed=9)
X = O.synth_rows(0, n, n, 1024, 5, mode)
g = O.synth_graph(n, 8, 16, 9)
Q = X[np.random.default_rng(3).integers(0, n, 8)]
t = DeviceTraversal(idx, Q, nts, log_pops=True)
try:
    t.run()
except Exception as e:
    print("run error:", e)
st = t.stats()
print("status", st.status, "scored", st.n_scored, "pops", st.n_pops, "repivot", st.n_repivot, "flush", st.n_flush)
for i in range(Q.shape[0]):
    want = O.rad_traverse(g, X, Q[i], nts)
    nodes, levels = t.pop_log(i)
    m = min(len(nodes), len(want.pop_nodes))
    d = np.nonzero((nodes[:m] != want.pop_nodes[:m]) | (levels[:m] != want.pop_levels[:m]))[0]
    s, a, o = t.results(i)
    print(f"q{i}: pops {len(nodes)} vs {len(want.pop_nodes)}; first pop divergence {d[0] if d.size else None}; scored {len(s)} vs {len(want.slots)}")
