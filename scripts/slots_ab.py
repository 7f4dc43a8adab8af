#!/usr/bin/env python3
"""A/B on one box: state per traversal against state per resident row (RADHIP_TRAV_SLOTS), and two objects on two streams whose
launches overlap (start / finish) against one launch after the other.  python scripts/slots_ab.py [rows=20000000] [nq=65536] [batches=6]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 6
nts = 100_000
idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=20260101, mode=2)
t0 = time.perf_counter(); idx.link_resident(seed=777, max_batch=16384); print(f"build {time.perf_counter() - t0:.1f} s ({n} rows)", flush=True)
rng = np.random.default_rng(4242)
batches = [idx.read_vectors(int(rng.integers(0, n - nq)), nq) for _ in range(nb)]

def seq(slots):
    t = DeviceTraversal(idx, batches[0], nts, slots=slots)
    t.run(0)                                   # warm-up
    pops = 0; ms = []
    w0 = time.perf_counter()
    for b in batches[1:]:
        t.reset(b); assert t.run(0) == 0
        k, l = t.kernel_time(); ms.append(k / l)
        pops += int(t.stats().n_pops.sum())
    wall = time.perf_counter() - w0
    h = t.result_hashes(0, 2048)
    print(f"{'rows' if slots else 'traversals'}-state sequential: {t.state_bytes() / 1e9:6.1f} GB, slots {t.slots}, kernel {np.mean(ms):7.1f} ms per launch, "
          f"{pops / wall / 1e9:.3f} G expansions/s whole steps, {pops / (sum(ms) * 1e-3) / 1e9:.3f} G by kernel time", flush=True)
    t.close()
    return h

def overlapped():
    A = DeviceTraversal(idx, batches[0], nts, slots=True, own_stream=True)
    B = DeviceTraversal(idx, batches[0], nts, slots=True, own_stream=True)
    A.run(0); B.run(0)
    objs = [A, B]
    pops = 0
    w0 = time.perf_counter()
    first = None
    for i, b in enumerate(batches[1:]):
        o = objs[i & 1]
        if i >= 2:
            assert o.finish() == 0
            pops += int(o.stats().n_pops.sum())
        o.reset(b); o.start()
        if first is None: first = o
    last = None
    for i in range(max(0, len(batches) - 3), len(batches) - 1):
        o = objs[i & 1]
        assert o.finish() == 0
        pops += int(o.stats().n_pops.sum())
        last = o
    wall = time.perf_counter() - w0
    h = last.result_hashes(0, 2048)
    print(f"two objects, launches overlapped: {2 * A.state_bytes() / 1e9:6.1f} GB, {pops / wall / 1e9:.3f} G expansions/s whole steps "
          f"({wall / (len(batches) - 1) * 1e3:.1f} ms per batch)", flush=True)
    A.close(); B.close()
    return h

h0 = seq(False)
h1 = seq(True)
h2 = overlapped()
print("hashes of the last batch agree:", bool(np.array_equal(h0, h1) and np.array_equal(h1, h2)))
