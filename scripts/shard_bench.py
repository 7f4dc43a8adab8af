#!/usr/bin/env python3
"""The row-sharded traversal's product loop on ONE GPU (RCCL communicator of world 1: kernels + collectives on
one stream, no second rank): what a frontier step costs before any inter-GPU latency.
    python scripts/shard_bench.py [n_rows] [nq] [n_to_score] [corpus_mode]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceShard, DeviceTraversal, RcclComm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
nts = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
mode = int(sys.argv[4]) if len(sys.argv) > 4 else 2
src = DeviceIndex(1024, 8, 16, 64); src.synth_vectors(n, seed=20260101, mode=mode)
X = np.empty((n, 128), np.uint8)
for f in range(0, n, 4_000_000):
    c = min(4_000_000, n - f); X[f:f + c] = src.read_vectors(f, c)
src.close()
idx = DeviceIndex(1024, 8, 16, 64)
for f in range(0, n, 5_000_000):
    idx.add_rows(X[f:f + 5_000_000], seed=777, max_batch=16384)
Q = X[np.random.default_rng(0).integers(0, n, nq)]
del X
ref = DeviceTraversal(idx, Q[:256], nts); ref.run(); want = ref.stats(); ref.close()
comm = RcclComm(0, 1, RcclComm.unique_id(), 0)
sh = DeviceShard(idx, 0, 1, 0, n, Q, nts)
print(f"engine {sh.engine}, state {sh.state_bytes() / 2**30:.1f} GiB", flush=True)
for rep in range(2):
    if rep: sh.reset(Q)
    t0 = time.perf_counter(); steps = sh.run(comm); dt = time.perf_counter() - t0
    st = sh.stats()
    ok = int(((st.n_pops[:256] == want.n_pops) & (st.n_scored[:256] == want.n_scored) & (st.n_nbr[:256] == want.n_nbr)).sum())
    print(f"rep {rep}: {nq} traversals to {nts}: {steps} frontier steps in {dt:.2f} s = {dt / steps * 1e6:.0f} us/step, "
          f"{st.n_pops.sum() / dt / 1e6:.1f} M expansions/s, {st.n_scored.sum() / dt / 1e6:.1f} M eval/s; parity vs the single-GPU kernel {ok}/256", flush=True)
