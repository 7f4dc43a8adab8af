"""Measurements for BASELINE.json configs other than the bench line (configs[0], [1], [4]) and the
K1 scan roofline.  Writes a markdown table to stdout.  Run on an MI355X."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd import _lib
from rad_amd.device import DeviceIndex, DeviceTraversal
from rad_amd.index import Index
from rad_amd.hnsw_service import LocalHNSWService
from rad_amd.traverser import RADTraverser

L = _lib.lib()
rows = []

def synth(n, ndim, seed, mode):
    d = DeviceIndex(ndim, 8, 16, 64)
    d.synth_vectors(n, seed=seed, mode=mode)
    x = d.read_vectors(0, n)
    d.close()
    return x

# ---- K1 scan roofline at 100M x 1024-bit -------------------------------------------------
idx = DeviceIndex(1024, 8, 16, 64)
n = 100_000_000
idx.synth_vectors(n, seed=3, mode=1)
q = idx.read_vectors(7, 8)
for nq in (1, 4, 8):
    chunk = 25_000_000                       # results are copied back: bound the host buffers
    ms = 0.0
    for f in range(0, n, chunk):
        idx.scan(q[:nq], f, chunk)
        ms += L.radhip_last_kernel_ms()
    gbs = n * 128 / (ms * 1e-3) / 1e9
    rows.append(("K1 scan", f"100M x 1024-bit, {nq} queries/pass", f"{ms:.1f} ms kernel", f"{n * nq / ms / 1e6:.1f} G eval/s",
                 f"{gbs:.0f} GB/s read = {gbs / 80:.1f} % of 8 TB/s (+ {8 * nq} B/row written)"))
idx.close()

# ---- config[0]: plumbing (host RADTraverser over a GPU-built index) ------------------------
for (n, M) in ((20_000, 16), (20_000, 8)):
    X = synth(n, 1024, 1234, 1)
    hnsw = Index(ndim=1024, dtype="b1", metric="tanimoto", connectivity=M, expansion_add=400)
    t0 = time.time(); hnsw.add(np.arange(n), X); tb = time.time() - t0
    def score(smiles, n=n):   # deterministic pseudo docking score
        return float((hash(smiles) % 100003) / 1000.0)
    svc = LocalHNSWService(hnsw)
    svc._transform_to_smiles_format = lambda data: [x if i % 2 == 0 else f"S{x}" for i, x in enumerate(int(v) for v in data)]
    tr = RADTraverser(hnsw_service=svc, scoring_fn=score)
    tr.prime()
    t0 = time.time(); tr.traverse(n_workers=1, n_to_score=n); dt = time.time() - t0
    st = tr.get_traversal_stats()["coordination"]
    pops = st["hnsw_proxy"]["total_neighbor_queries"]
    rows.append(("config[0] plumbing", f"{n} x 1024-bit, connectivity={M}, expansion_add=400, user scoring_fn, n_workers=1",
                 f"Index.add {tb:.2f} s on GPU", f"{pops / dt:.0f} expansions/s (host, Python)", f"{st['scored_molecules']} scored in {dt:.2f} s"))

# ---- config[1]: 1M x 1024-bit, connectivity 8 -------------------------------------------------
for mode, tag in ((1, "clustered sparse (~7 %)"), (0, "Bernoulli(0.5)")):
    n = 1_000_000
    X = synth(n, 1024, 1, mode)
    hnsw = Index(ndim=1024, connectivity=8, expansion_add=64, max_batch=4096)
    t0 = time.time(); hnsw.add(np.arange(n), X); tb = time.time() - t0
    rows.append(("config[1] build", f"1M x 1024-bit {tag}, connectivity=8, expansion_add=64, batches of 4096", f"{tb:.1f} s", f"{n / tb:.0f} inserts/s", ""))
    rng = np.random.default_rng(0)
    Q = X[rng.integers(0, n, 4096)]
    for ef in (64, 400):
        t0 = time.time(); m = hnsw.search(Q, count=10, expansion=ef); dt = time.time() - t0
        ex = hnsw.search(Q[:64], count=10, exact=True)
        rec = np.mean([len(set(m.slots[i]) & set(ex.slots[i])) / 10 for i in range(64)])
        rows.append(("config[1] search", f"1M {tag}, 4096 queries, k=10, ef={ef}", f"{dt * 1e3:.0f} ms wall (incl. copies)",
                     f"{m.computed_distances / dt / 1e9:.2f} G eval/s, {m.visited_members / dt / 1e6:.1f} M expansions/s", f"recall@10 vs exact scan = {rec:.3f}"))
    dev = hnsw.device_index()
    cap = dev.traversal_capacity()
    tq = X[rng.integers(0, n, cap)]
    t = DeviceTraversal(dev, tq, 100_000)
    t.run()
    ms, _ = t.kernel_time(); st = t.stats()
    rows.append(("config[1] traversal", f"1M {tag} (graph built on the GPU), {cap} traversals to n_to_score=100k", f"{ms:.1f} ms kernel",
                 f"{st.n_pops.sum() / ms / 1e3:.0f} M expansions/s, {st.n_scored.sum() / ms / 1e6:.2f} G eval/s",
                 f"{(st.n_scored.sum() * 132 + st.n_pops.sum() * 4) / ms / 1e6:.0f} GB/s algorithmic, {st.n_scored.sum() / st.n_pops.sum():.2f} evals/expansion"))
    t.close()

# ---- config[4]: 2048-bit, connectivity 32, level-0 width 64 ------------------------------------
idx = DeviceIndex(2048, 32, 64, 400)
n = 20_000_000
idx.synth_vectors(n, seed=5, mode=1)
idx.synth_graph(seed=6)
cap = idx.traversal_capacity()
for nq in (cap // 4, cap):
    tq = idx.read_vectors(1000, nq)
    t = DeviceTraversal(idx, tq, 100_000)
    t.run()
    ms, _ = t.kernel_time(); st = t.stats()
    rows.append(("config[4]", f"20M x 2048-bit, connectivity=32 (level-0 width 64), synthetic graph, {nq} traversals (trav_kernel, 1 per wave) to 100k",
                 f"{ms:.1f} ms kernel", f"{st.n_pops.sum() / ms / 1e3:.0f} M expansions/s, {st.n_scored.sum() / ms / 1e6:.2f} G eval/s",
                 f"{(st.n_scored.sum() * 260 + st.n_pops.sum() * 4) / ms / 1e6:.0f} GB/s algorithmic = {(st.n_scored.sum() * 260 + st.n_pops.sum() * 4) / ms / 1e6 / 80:.1f} % of 8 TB/s, {st.n_scored.sum() / st.n_pops.sum():.1f} evals/expansion"))
    t.close()

print("| what | configuration | time | throughput | notes |")
print("|---|---|---|---|---|")
for r in rows:
    print("| " + " | ".join(r) + " |")
