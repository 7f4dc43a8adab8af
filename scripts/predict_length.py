import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal
n, nq = 20_000_000, 32768
os.environ["RADHIP_TRAV"] = "4"
idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=20260101, mode=2)
idx.link_resident(seed=777, max_batch=16384)
Q = idx.read_vectors(5, nq)
t = DeviceTraversal(idx, Q, 100_000)
t.run()
st = t.stats()
p = st.n_pops.astype(np.float64)
qpop = np.unpackbits(Q, axis=1).sum(1).astype(np.float64)
print("corr(pops, query popcount) =", round(float(np.corrcoef(p, qpop)[0, 1]), 3))
print("corr(pops, upper-level visits) =", round(float(np.corrcoef(p, st.n_upper.astype(np.float64))[0, 1]), 3))
# the query IS a corpus row here (row 5 + i): its own level in the graph, its degree
levels, adj0, upper_row, adjU = idx.read_graph()
rows = np.arange(5, 5 + nq)
deg = (adj0[rows] != 0xFFFFFFFF).sum(1).astype(np.float64)
print("corr(pops, level-0 degree of the query's own node) =", round(float(np.corrcoef(p, deg)[0, 1]), 3))
# first 2000 pops: new nodes per pop AND neighbours per pop
t.reset(Q); t.run(max_pops=2000); s1 = t.stats()
r1 = s1.n_scored / np.maximum(s1.n_pops, 1); r2 = s1.n_nbr / np.maximum(s1.n_pops, 1)
print("after 2000 pops: corr(pops, scored/pop) =", round(float(np.corrcoef(p, r1)[0, 1]), 3), " corr(pops, neighbours/pop) =", round(float(np.corrcoef(p, r2)[0, 1]), 3))
A = np.stack([r1, r2, qpop, np.ones(nq)], 1); coef = np.linalg.lstsq(A, p, rcond=None)[0]
print("linear fit on (scored/pop, nbr/pop, popcount): corr =", round(float(np.corrcoef(p, A @ coef)[0, 1]), 3))
