#!/usr/bin/env python3
"""Which row of which wavefront worked on which traversal (how profiles/r03/which_row_worked_on_which_traversal.log was made).
Needs a library built with a one-line debug patch in save_row of rad_amd/csrc/traverse4.inc, after the header fields are
written:      H->n_upper = blockIdx.x * 4u + g;
(the statistics' n_upper field then carries the row's number).   python scripts/owner_debug.py <connectivity> <nq>"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal
M = int(sys.argv[1]); nq = int(sys.argv[2]); n = 20_000_000
os.environ["RADHIP_TRAV"] = "4"; os.environ["RADHIP_TRAV_STATIC"] = "0"
idx = DeviceIndex(1024, M, 2 * M, 64)
idx.synth_vectors(n, seed=20260101, mode=2)
idx.link_resident(seed=777, max_batch=16384)
t = DeviceTraversal(idx, idx.read_vectors(5, nq), 100_000)
t.run()
ms, nl = t.kernel_time(); st = t.stats()
own = st.n_upper.astype(np.int64)
rows, cnt = np.unique(own, return_counts=True)
waves, wcnt = np.unique(own // 4, return_counts=True)
print(f"M={M} nq={nq}: {ms/nl:.1f} ms; rows that worked {len(rows)}; traversals per row: " + ", ".join(f"{k}:{int((cnt==k).sum())}" for k in range(1, cnt.max()+1)))
print(f"  wavefronts that worked {len(waves)}; traversals per wavefront min {wcnt.min()} mean {wcnt.mean():.2f} max {wcnt.max()}; histogram " + ", ".join(f"{k}:{int((wcnt==k).sum())}" for k in sorted(set(wcnt.tolist()))))
p = st.n_pops.astype(np.float64)
work = np.zeros(own.max() // 4 + 1); np.add.at(work, own // 4, p)
roww = np.zeros(own.max() + 1); np.add.at(roww, own, p)
live = roww.reshape(-1, 4).max(1) if len(roww) % 4 == 0 else None
if live is not None:
    print(f"  wave-rounds if a wavefront lives as long as its busiest row: {live.sum()/1e6:.1f} M; ideal (all rows busy) {p.sum()/4e6:.1f} M; longest row {roww.max():.0f} pops")
hog = rows[cnt > 2]
print("  rows with more than two traversals, by position in their wavefront:", {int(k): int((hog % 4 == k).sum()) for k in range(4)})
one = rows[cnt == 1]
print("  rows with exactly one traversal, by position:", {int(k): int((one % 4 == k).sum()) for k in range(4)})
