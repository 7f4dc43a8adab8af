#!/usr/bin/env python3
"""One built graph (20M hierarchical rows), one batch of traversals, two launches (profiling runs: RADHIP_TRAV_STATIC etc.
come from the environment).      python scripts/one_batch.py [connectivity = 8] [nq = 32768]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
n = 20_000_000
os.environ.setdefault("RADHIP_TRAV", "4")
idx = DeviceIndex(1024, M, 2 * M, 64)
idx.synth_vectors(n, seed=20260101, mode=2)
idx.link_resident(seed=777, max_batch=16384)
t = DeviceTraversal(idx, idx.read_vectors(5, nq), 100_000)
for rep in range(2):
    if rep:
        t.reset(idx.read_vectors(77, nq))
    t.run()
    ms, nl = t.kernel_time(); st = t.stats()
    print(f"M={M} nq={nq} static={os.environ.get('RADHIP_TRAV_STATIC', 'default')}: {ms / nl:.1f} ms per launch, {int(st.n_pops.sum()) * nl / ms / 1e3:.0f} M expansions/s", flush=True)
