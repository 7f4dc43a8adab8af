#!/usr/bin/env python3
"""The reference's own index parameters (examples/DUDEZ_example.ipynb:165-166, 183-189: 1024-bit Morgan fingerprints,
connectivity = 16 -> level-0 rows of 32 slots, expansion_add = 400) on the synthetic hierarchical corpus: build time,
recall@10, and the rate and algorithmic fraction of both traversal kernels (trav_kernel: one traversal per wavefront, a 64-lane row;
trav4_kernel's WIDE form: four per wavefront, a row walked in chunks of 16).      python scripts/notebook_shape.py [n_rows] [connectivity] [expansion_add] [ndim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd import _lib
from rad_amd._lib import check, ptr
from rad_amd.device import DeviceIndex, DeviceTraversal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ef_add = int(sys.argv[3]) if len(sys.argv) > 3 else 400
ndim = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
nts = 100_000
idx = DeviceIndex(ndim, M, 2 * M, ef_add)
idx.synth_vectors(n, seed=20260101, mode=2)
t0 = time.time()
idx.link_resident(seed=777, max_batch=16384)
tb = time.time() - t0
rng = np.random.default_rng(1)
Q = np.concatenate([idx.read_vectors(int(r), 1) for r in rng.integers(0, n, 256)])
es, _a, _o, _c = idx.topk(Q, 10)
rec = {}
for ef in (128, 400):
    s = np.full((256, 10), 0xFFFFFFFF, np.uint32); a = np.zeros((256, 10), np.uint32); o = np.zeros((256, 10), np.uint32); c = np.zeros(256, np.uint32)
    check(_lib.lib().radhip_search(idx._h, ptr(Q), 256, 10, ef, ptr(s), ptr(a), ptr(o), ptr(c), None, None))
    rec[ef] = float(np.mean([len(set(s[i]) & set(es[i])) / 10 for i in range(256)]))
print(f"n={n} {ndim}-bit connectivity={M} (level-0 width {2 * M}) expansion_add={ef_add}: Index build {tb:.1f} s ({n / tb / 1e6:.2f} M inserts/s), "
      f"recall@10 ef128 {rec[128]:.3f} ef400 {rec[400]:.3f}", flush=True)
B = idx.info().row_stride
for kern, mult, table in (("1", 1, None), ("1", 2, None), ("4", 1, None), ("4", 2, None), ("4", 4, None), ("4", 2, "group"), ("4", 3, "group")):
    os.environ["RADHIP_TRAV"] = kern
    if table: os.environ["RADHIP_TABLE"] = table
    else: os.environ.pop("RADHIP_TABLE", None)
    cap = idx.traversal_capacity()
    nq = cap * mult
    try:
        t = DeviceTraversal(idx, idx.read_vectors(int(rng.integers(0, n - nq)), nq), nts)
    except Exception as e:   # (state of the larger batches may not fit)
        print(f"  {nq} traversals (RADHIP_TRAV={kern}): {e}", flush=True)
        continue
    t.run()
    t.reset(idx.read_vectors(int(rng.integers(0, n - nq)), nq))     # second batch: the timed one
    t.run()
    ms, _ = t.kernel_time(); st = t.stats()
    pops, ev = int(st.n_pops.sum()), int(st.n_scored.sum())
    alg = (ev * (B + 4) + pops * 4) / (ms * 1e-3) / 1e9
    print(f"  {nq} traversals to {nts} ({t.kernel}, table {t.table}): {ms:.1f} ms, {pops / ms / 1e3:.0f} M expansions/s, {ev / ms / 1e6:.2f} G evals/s, "
          f"{ev / pops:.1f} evals/expansion, algorithmic {alg:.0f} GB/s = {alg / 8000:.3f} of 8 TB/s; state {t.state_bytes() / nq / 1e6:.2f} MB/traversal", flush=True)
    t.close()
