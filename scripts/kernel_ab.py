#!/usr/bin/env python3
"""One variant of the traversal kernel on the bench workload's shape: two objects with per-row state on two streams, launches
overlapped (what bench.py times).  RADHIP_LIB picks the build, RADHIP_TABLE the table.
    python scripts/kernel_ab.py [rows=20000000] [nq=65536] [batches=7] [expansion_add=64] [connectivity=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 7
ef = int(sys.argv[4]) if len(sys.argv) > 4 else 64
M = int(sys.argv[5]) if len(sys.argv) > 5 else 8
nts = 100_000
LABEL = os.path.basename(os.path.dirname(os.environ['RADHIP_LIB'])) if os.environ.get('RADHIP_LIB') else '_build'
idx = DeviceIndex(1024, M, 2 * M, ef)
idx.synth_vectors(n, seed=20260101, mode=2)
t0 = time.perf_counter(); idx.link_resident(seed=777, max_batch=16384); tb = time.perf_counter() - t0
rng = np.random.default_rng(4242)
batches = [idx.read_vectors(int(rng.integers(0, n - nq)), nq) for _ in range(nb)]
if os.environ.get("KERNEL_AB_CHAIN"):
    # all timed batches in ONE launch (a ring of scored lists), after a one-batch warm-up launch
    allq = np.concatenate(batches[1:])
    T = DeviceTraversal(idx, allq, nts, list_ring=2 * nq)
    T.reset(batches[0]); T.run(0)
    w0 = time.perf_counter()
    T.reset(allq); assert T.run(0) == 0
    wall = time.perf_counter() - w0
    st = T.stats(); pops, evals = int(st.n_pops.sum()), int(st.n_scored.sum())
    k, l = T.kernel_time()
    h = T.result_hashes((nb - 2) * nq, 1024)
    print(f"{LABEL:14s} table {T.table:8s} rows {n} build {tb:5.1f} s: {pops / wall / 1e9:.3f} G expansions/s whole steps "
          f"({wall / (nb - 1) * 1e3:7.1f} ms per batch of {nq}; ONE launch of {nb - 1} batches: {k / l:8.1f} ms = {pops / (k / l * 1e-3) / 1e9:.3f} G by kernel time; "
          f"{evals / max(pops, 1):.2f} evals/expansion; hash {int(np.bitwise_xor.reduce(h)):016x})", flush=True)
    sys.exit(0)
A = DeviceTraversal(idx, batches[0], nts, slots=True, own_stream=True)
B = DeviceTraversal(idx, batches[0], nts, slots=True, own_stream=True)
A.run(0); B.run(0)                                     # warm-up (also brings every row's epoch past its first use)
objs = [A, B]
pops = evals = 0
kms = []
def done(o):
    global pops, evals
    assert o.finish() == 0
    st = o.stats(); pops += int(st.n_pops.sum()); evals += int(st.n_scored.sum())
    k, l = o.kernel_time(); kms.append(k / l)
w0 = time.perf_counter()
for i, b in enumerate(batches[1:]):
    o = objs[i & 1]
    if i >= 2: done(o)
    o.reset(b); o.start()
for i in range(max(0, nb - 3), nb - 1): done(objs[i & 1])
wall = time.perf_counter() - w0
h = objs[(nb - 2) & 1].result_hashes(0, 1024)
print(f"{LABEL:14s} table {A.table:8s} rows {n} build {tb:5.1f} s: {pops / wall / 1e9:.3f} G expansions/s whole steps "
      f"({wall / (nb - 1) * 1e3:7.1f} ms per batch of {nq}; one launch alone in flight ~{np.mean(kms):7.1f} ms; {evals / max(pops, 1):.2f} evals/expansion; "
      f"hash {int(np.bitwise_xor.reduce(h)):016x})", flush=True)
