#!/usr/bin/env python3
"""A/B of the visited/scored tables of trav4_kernel on ONE built graph, in one process, alternating:
    python scripts/ab_table.py [n_rows] [corpus_mode] [nq] [n_to_score] [tables: hash,bucket,...] [reps]
Prints kernel ms per launch and expansions/s for every table, and checks that all of them return the same
counters.  RADHIP_TABLE is read at traversal create, so one index serves all variants."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from rad_amd.device import DeviceIndex, DeviceTraversal

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 32768
nts = int(sys.argv[4]) if len(sys.argv) > 4 else 100_000
tables = (sys.argv[5] if len(sys.argv) > 5 else "hash,bucket").split(",")
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
os.environ.setdefault("RADHIP_TRAV", "4")

idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=20260101, mode=mode)
t0 = time.time()
idx.link_resident(seed=777, max_batch=16384)
print(f"build {time.time() - t0:.1f} s ({n} rows, corpus mode {mode})", flush=True)
rng = np.random.default_rng(0)
Qs = [idx.read_vectors(int(rng.integers(0, n - nq)), nq) for _ in range(reps + 1)]
ref = None
for rep in range(reps + 1):
    for tb in tables:
        if tb == "bucket":
            os.environ.pop("RADHIP_TABLE", None)
        else:
            os.environ["RADHIP_TABLE"] = tb
        t = DeviceTraversal(idx, Qs[rep], nts)
        assert t.run() == 0
        ms, nl = t.kernel_time()
        st = t.stats()
        pops = int(st.n_pops.sum())
        sig = (pops, int(st.n_scored.sum()), int(st.n_nbr.sum()))
        if tb == tables[0]:
            ref = sig
        assert sig == ref, (tb, sig, ref)
        if rep:
            print(f"rep {rep} {t.table:8s} {ms / nl:8.1f} ms  {pops / (ms * 1e-3) / 1e9:.3f} G expansions/s  "
                  f"({pops / nq:.0f} pops, {sig[1] / max(pops, 1):.2f} evals/pop; state {t.state_bytes() / nq / 1e6:.2f} MB/traversal; "
                  f"upper-level visits max {int(st.n_upper.max())} mean {st.n_upper.mean():.0f}; flushes {st.n_flush.mean():.0f} re-pivots {st.n_repivot.mean():.0f} re-mids {st.n_remid.mean():.0f})", flush=True)
        t.close()
