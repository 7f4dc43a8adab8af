#!/usr/bin/env python3
"""Parameter sweep of trav4_kernel's queue: `prepare` builds 20M-row graphs of both corpora once and parks them in
/tmp; `run <tag>` loads them through the library named by RADHIP_LIB and times the traversal kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
n, nq, nts = 20_000_000, 16384, 100_000
if sys.argv[1] == "prepare":
    from rad_amd.device import DeviceIndex
    for mode in (1, 2):
        src = DeviceIndex(1024, 8, 16, 64); src.synth_vectors(n, seed=20260101, mode=mode)
        X = np.empty((n, 128), np.uint8)
        for f in range(0, n, 4_000_000):
            X[f:f + 4_000_000] = src.read_vectors(f, min(4_000_000, n - f))
        src.close()
        idx = DeviceIndex(1024, 8, 16, 64)
        for f in range(0, n, 5_000_000):
            idx.add_rows(X[f:f + 5_000_000], seed=777, max_batch=16384)
        lv, a0, ur, aU = idx.read_graph()
        inf = idx.info()
        np.save(f"/tmp/qs_X{mode}.npy", X)
        np.savez(f"/tmp/qs_g{mode}.npz", levels=lv, adj0=a0, upper_row=ur, adjU=aU, max_level=inf.max_level, entry=inf.entry)
        idx.close()
    print("prepared")
else:
    os.environ["RADHIP_TRAV"] = "4"; os.environ["RADHIP_TABLE"] = "hash"
    from rad_amd.device import DeviceIndex, DeviceTraversal
    out = []
    for mode in (1, 2):
        X = np.load(f"/tmp/qs_X{mode}.npy", mmap_mode="r")
        z = np.load(f"/tmp/qs_g{mode}.npz")
        idx = DeviceIndex(1024, 8, 16, 64)
        idx.load_vectors(np.asarray(X))
        idx.load_graph(z["levels"], z["adj0"], z["upper_row"], z["adjU"], int(z["max_level"]), int(z["entry"]))
        Q = np.asarray(X[np.sort(np.random.default_rng(0).integers(0, n, nq))])
        t = DeviceTraversal(idx, Q, nts)
        ms = []
        for rep in range(3):
            if rep: t.reset(Q)
            t.run(); ms.append(t.kernel_time()[0])
        st = t.stats()
        out.append(f"mode {mode}: {min(ms):.1f} ms (flushes {st.n_flush.mean():.0f} remids {st.n_remid.mean():.0f} repivots {st.n_repivot.mean():.0f})")
        t.close(); idx.close()
    print(sys.argv[2], " | ".join(out), flush=True)
