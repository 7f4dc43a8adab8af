#!/usr/bin/env python3
"""The rate of the traversal kernel on a full device (the marginal rate between two batch sizes of ONE launch) against the whole-step
rate of the two-object pipeline: how much of a launch's tail the overlapped launches recover.
    python scripts/marginal_rate.py [rows=100000000] [expansion_add=64]          (RADHIP_TABLE picks the table)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nts = 100_000
idx = DeviceIndex(1024, 8, 16, ef)
idx.synth_vectors(n, seed=20260101, mode=2)
t0 = time.perf_counter(); idx.link_resident(seed=777, max_batch=16384); print(f"build {time.perf_counter() - t0:.1f} s ({n} rows, expansion_add {ef})", flush=True)
rng = np.random.default_rng(4242)
Q = idx.read_vectors(int(rng.integers(0, n - 131072)), 131072)
res = {}
for nq in (65536, 131072):
    t = DeviceTraversal(idx, Q[:nq], nts, slots=True)
    t.run(0)                                  # warm-up
    ms_l, pops = [], 0
    for rep in range(2):
        t.reset(Q[:nq]); assert t.run(0) == 0
        k, l = t.kernel_time(); ms_l.append(k / l)
        pops = int(t.stats().n_pops.sum())
    res[nq] = (min(ms_l), pops)
    print(f"one launch of {nq:6d} traversals ({t.table} table, {t.slots} rows' tables, {t.state_bytes() / 1e9:.1f} GB): {min(ms_l):8.1f} ms, {pops / (min(ms_l) * 1e-3) / 1e9:.3f} G expansions/s", flush=True)
    t.close()
(m1, p1), (m2, p2) = res[65536], res[131072]
marg = (p2 - p1) / ((m2 - m1) * 1e-3) / 1e9
print(f"marginal rate between the two (the kernel on a full device): {marg:.3f} G expansions/s; tail of a launch ~ {m1 - p1 / (marg * 1e9) * 1e3:.0f} ms")
# the pipeline: two objects, 65536 per batch
batches = [Q[:65536], Q[65536:]] * 4
A = DeviceTraversal(idx, batches[0], nts, slots=True, own_stream=True)
B = DeviceTraversal(idx, batches[1], nts, slots=True, own_stream=True)
A.run(0); B.run(0)
objs, pops = [A, B], 0
w0 = time.perf_counter()
for i, b in enumerate(batches):
    o = objs[i & 1]
    if i >= 2:
        assert o.finish() == 0; pops += int(o.stats().n_pops.sum())
    o.reset(b); o.start()
for o in objs:
    assert o.finish() == 0; pops += int(o.stats().n_pops.sum())
wall = time.perf_counter() - w0
print(f"two objects, {len(batches)} batches of 65536, launches overlapped: {pops / wall / 1e9:.3f} G expansions/s whole steps = {pops / wall / 1e9 / marg:.3f} of the marginal rate")
