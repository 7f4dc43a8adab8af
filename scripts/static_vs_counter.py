#!/usr/bin/env python3
"""trav4_kernel on one built graph (20M hierarchical rows): rows that take their traversals from the counter against
rows that keep the traversal their block index names (RADHIP_TRAV_STATIC=1), at two and four resident rounds per launch.
    python scripts/wide_grid.py [connectivity = 16]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal
n, M = 20_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 16
idx = DeviceIndex(1024, M, 2 * M, 64)
idx.synth_vectors(n, seed=20260101, mode=2)
idx.link_resident(seed=777, max_batch=16384)
rng = np.random.default_rng(1)
os.environ["RADHIP_TRAV"] = "4"
for nq, grid in ((32768, 0), (32768, 1), (65536, 0), (65536, 1)):
    if grid: os.environ["RADHIP_TRAV_STATIC"] = "1"
    else: os.environ.pop("RADHIP_TRAV_STATIC", None)
    t = DeviceTraversal(idx, idx.read_vectors(5, nq), 100_000)
    t.run()
    ms, _ = t.kernel_time(); st = t.stats()
    print(f"M={M} nq {nq} {'rows take traversals from the counter' if not grid else 'static rows'}: {ms:.1f} ms, {int(st.n_pops.sum()) / ms / 1e3:.0f} M expansions/s; flushes {st.n_flush.mean():.0f} re-pivots {st.n_repivot.mean():.0f} re-mids {st.n_remid.mean():.0f}", flush=True)
    t.close()
