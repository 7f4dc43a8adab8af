"""Exploration harness (not the bench): time the traversal kernel at a given size."""
import argparse, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=100_000_000)
ap.add_argument("--ndim", type=int, default=1024)
ap.add_argument("--M", type=int, default=8)
ap.add_argument("--nq", type=str, default="cap")
ap.add_argument("--nts", type=str, default="100000")
ap.add_argument("--mode", type=int, default=1)
a = ap.parse_args()
t0 = time.time()
idx = DeviceIndex(a.ndim, a.M, 2 * a.M, 64)
idx.synth_vectors(a.n, seed=3, mode=a.mode)
t1 = time.time()
idx.synth_graph(seed=4)
t2 = time.time()
inf = idx.info()
print(f"n={a.n} synth rows {t1-t0:.2f}s graph {t2-t1:.2f}s max_level={inf.max_level} dev_bytes={inf.device_bytes/1e9:.2f} GB", flush=True)
B = inf.row_stride
for nts in [int(x) for x in a.nts.split(",")]:
    for nq in [idx.traversal_capacity() * int(x[3:] or 1) if x.startswith("cap") else int(x) for x in a.nq.split(",")]:
        Q = idx.read_vectors(12345, nq)
        t3 = time.time()
        t = DeviceTraversal(idx, Q, nts)
        t4 = time.time()
        t.run()
        t5 = time.time()
        ms, _ = t.kernel_time()
        st = t.stats()
        ev, pops, nbr = int(st.n_scored.sum()), int(st.n_pops.sum()), int(st.n_nbr.sum())
        gbs = (ev * (B + 4) + pops * 4) / (ms * 1e-3) / 1e9
        print(f"nts={nts} nq={nq}: create {t4-t3:.2f}s wall {t5-t4:.3f}s kernel {ms:.2f} ms  pops={pops} evals={ev} nbr={nbr} "
              f"evals/pop={ev/pops:.2f} | {pops/ms/1e3:.2f} M exp/s {ev/ms/1e6:.3f} G eval/s  alg {gbs:.1f} GB/s ({gbs/8000*100:.1f}% of 8TB/s) "
              f"state {t.state_bytes()/1e9:.2f} GB status={np.bincount(st.status+8)[8:]} repivots/trav={st.n_repivot.mean():.0f} flushes/trav={st.n_flush.mean():.0f}", flush=True)
        t.close()
