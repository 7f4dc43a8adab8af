"""Experiment (negative result, profiles/r01/README.md): does a cluster-contiguous renumbering of the
corpus speed the traversal up?  Relabels the synthetic graph on the host (numpy), loads both layouts
and times the traversal kernel on each.  The queue order depends on slot ids (tie-break), so the two
layouts are different workloads with the same statistics."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
nts = 100_000
src = DeviceIndex(1024, 8, 16, 64)
src.synth_vectors(n, seed=3, mode=1); src.synth_graph(seed=4)
X = np.concatenate([src.read_vectors(f, min(2_000_000, n - f)) for f in range(0, n, 2_000_000)])
levels, adj0, upper_row, adjU = src.read_graph()
inf = src.info(); cap = src.traversal_capacity()
def run(idx, Q, tag):
    t = DeviceTraversal(idx, Q, nts); t.run(); ms, _ = t.kernel_time(); st = t.stats()
    print(f"{tag}: {ms:.1f} ms  {st.n_pops.sum()/ms/1e3:.0f} M exp/s  {st.n_scored.sum()/ms/1e6:.2f} G eval/s  evals/pop {st.n_scored.sum()/st.n_pops.sum():.2f}", flush=True)
    t.close()
qrows = np.random.default_rng(0).integers(0, n, cap)
run(src, X[qrows], "scattered ids (as generated)")
# cluster-contiguous relabel: cluster c = slot % nc, member m = slot // nc  ->  new = c * cs + m
nc = n // 32
old = np.arange(n, dtype=np.int64)
c, m = old % nc, old // nc
cs = (n - c + nc - 1) // nc
start = np.concatenate([[0], np.cumsum(((n - np.arange(nc) + nc - 1) // nc))[:-1]])
new_of_old = (start[c] + m).astype(np.uint32)
order = np.argsort(new_of_old)           # old slot at each new position
def remap(a):
    out = a.copy(); mask = a != 0xFFFFFFFF; out[mask] = new_of_old[a[mask]]; return out
X2 = X[order]; levels2 = levels[order]; adj02 = remap(adj0)[order]
# upper rows: keep row storage, remap targets, permute the per-node base pointers
upper_row2 = upper_row[order]; adjU2 = remap(adjU)
dst = DeviceIndex(1024, 8, 16, 64)
dst.load_vectors(X2); dst.load_graph(levels2, adj02, upper_row2, adjU2, int(inf.max_level), int(new_of_old[inf.entry]))
run(dst, X2[new_of_old[qrows]], "cluster-contiguous ids")
