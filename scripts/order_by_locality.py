#!/usr/bin/env python3
"""Does the ORDER of the traversals of a batch matter?  Rows take traversals in index order, so neighbours in the batch run
at the same time: if they explore the same region of the graph their fingerprint and adjacency reads can meet in L2 / MALL.
Queries sorted by the graph-locality layout id of their nearest node (one ef = 16 search each) against the batch as it comes.
    python scripts/order_by_locality.py [n_rows = 20M] [nq = 65536]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd import _lib
from rad_amd._lib import check, ptr
from rad_amd.device import DeviceIndex, DeviceTraversal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
os.environ["RADHIP_TRAV"] = "4"
idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=20260101, mode=2)
idx.link_resident(seed=777, max_batch=16384)
t0 = time.time(); lay = idx.optimize_layout(); lid = idx.read_layout()
print(f"layout {time.time() - t0:.1f} s (groups/row {lay.groups_per_row:.2f})", flush=True)
rng = np.random.default_rng(3)
Q = idx.read_vectors(int(rng.integers(0, n - nq)), nq)
t0 = time.time()
s = np.full((nq, 1), 0xFFFFFFFF, np.uint32); a = np.zeros((nq, 1), np.uint32); o = np.zeros((nq, 1), np.uint32); c = np.zeros(nq, np.uint32)
check(_lib.lib().radhip_search(idx._h, ptr(Q), nq, 1, 16, ptr(s), ptr(a), ptr(o), ptr(c), None, None))
near = s[:, 0]
print(f"nearest node of {nq} queries (ef 16): {time.time() - t0:.2f} s", flush=True)
orders = {"as it comes": np.arange(nq), "sorted by the layout id of the nearest node": np.argsort(lid[near], kind="stable"),
          "sorted by the nearest node's slot": np.argsort(near, kind="stable"), "shuffled": rng.permutation(nq)}
t = DeviceTraversal(idx, Q, 100_000)
ref = None
for rep in range(2):
    for name, od in orders.items():
        t.reset(Q[od]); t.run()
        ms, nl = t.kernel_time(); st = t.stats()
        pops = int(st.n_pops.sum())
        inv = np.empty(nq, np.int64); inv[od] = np.arange(nq)
        sig = st.n_pops[inv]                       # back in the original order: the same traversals, whatever the order
        if ref is None: ref = sig
        assert np.array_equal(sig, ref)
        print(f"rep {rep} {name:45s}: {ms / nl:7.1f} ms  {pops / (ms / nl) / 1e6:.3f} G expansions/s", flush=True)
