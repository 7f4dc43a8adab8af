"""Parity and graph quality at the sizes the bench really runs:
  * Index.add with the bench's batch size (max_batch = 16384) against the oracle builder, adjacency bit-exact
    (the rocPRIM request sort, the 16-bit source offsets and req_cap only matter at such batches);
  * the traversal kernel on a GPU-BUILT graph of 20M rows against the oracle, full results;
  * recall@10 of the batched GPU build against the oracle's sequential build (max_batch = 1) on the
    hierarchical corpus, ground truth from the exact top-k kernel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NO_SLOT = 0xFFFFFFFF


def test_index_add_at_bench_batch_size_matches_oracle(gpu, oracle):
    from rad_amd.device import DeviceIndex
    n, ndim, M, ef = 300_000, 1024, 8, 64
    X = oracle.synth_rows(0, n, n, ndim, 20260101, 2)
    h = oracle.Hnsw(ndim, M, 2 * M, ef, seed=777)
    h.add(X, max_batch=16384)
    g = h.graph()
    idx = DeviceIndex(ndim, M, 2 * M, ef)
    idx.add_rows(X, seed=777, max_batch=16384)
    levels, adj0, upper_row, adjU = idx.read_graph()
    inf = idx.info()
    assert inf.max_level == g.max_level and inf.entry == g.entry
    assert np.array_equal(levels, g.levels) and np.array_equal(upper_row, g.upper_row)
    bad = np.nonzero((adj0 != g.adj0).any(1))[0]
    assert bad.size == 0, f"{bad.size} level-0 rows differ, first {bad[:8]}"
    assert np.array_equal(adjU, g.adjU)
    # the last batches of this build hold 16384 nodes each: the schedule really reached the bench's batch size
    assert n // 16 > 16384


def test_traversal_on_gpu_built_graph_20m_matches_oracle(gpu, oracle, monkeypatch):
    from rad_amd.device import DeviceIndex, DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "4")        # the bench kernel
    n, nts = 20_000_000, 100_000
    src = DeviceIndex(1024, 8, 16, 64)
    src.synth_vectors(n, seed=20260101, mode=2)
    X = np.empty((n, 128), np.uint8)
    for f in range(0, n, 4_000_000):
        X[f:f + 4_000_000] = src.read_vectors(f, min(4_000_000, n - f))
    src.close()
    idx = DeviceIndex(1024, 8, 16, 64)
    for f in range(0, n, 5_000_000):
        idx.add_rows(X[f:f + 5_000_000], seed=777, max_batch=16384)
    levels, adj0, upper_row, adjU = idx.read_graph()
    inf = idx.info()
    g = oracle.Graph(n, 16, 8, int(inf.max_level), int(inf.entry), levels, adj0, upper_row, adjU)
    Q = X[np.random.default_rng(3).integers(0, n, 6)].copy()
    t = DeviceTraversal(idx, Q, nts, log_pops=True)
    assert t.kernel == "trav4_kernel" and t.run() == 0
    st = t.stats()
    for i in range(Q.shape[0]):
        want = oracle.rad_traverse(g, X, Q[i], nts)
        s, a, o = t.results(i)
        nodes, lv = t.pop_log(i)
        assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(lv, want.pop_levels), i
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
        assert st.n_pops[i] == want.n_pops and st.n_nbr[i] == want.n_nbr
    assert (st.n_remid > 0).all()                 # long enough to exercise the three-level queue's refills


def _recall(idx, Q, k=10, ef=128):
    import ctypes as C
    from rad_amd import _lib
    from rad_amd._lib import check, ptr
    nq = Q.shape[0]
    s = np.full((nq, k), NO_SLOT, np.uint32); a = np.zeros((nq, k), np.uint32); o = np.zeros((nq, k), np.uint32)
    cnt = np.zeros(nq, np.uint32)
    check(_lib.lib().radhip_search(idx._h, ptr(Q), nq, k, ef, ptr(s), ptr(a), ptr(o), ptr(cnt), None, None))
    es, _a, _o, _c = idx.topk(Q, k)
    return float(np.mean([len(set(s[i]) & set(es[i])) / k for i in range(nq)]))


def test_batched_build_recall_matches_sequential_build(gpu, oracle):
    """the graph the bench traverses is a graph: recall@10 of the 16384-batch GPU build stays within 0.03 of the
    oracle's classical sequential insert (and above 0.9) on a corpus with neighbourhood structure at every
    scale.  (At 300k rows the last batches are 6 % of the graph each; at 1M rows the gap is 0.01 — 0.914 vs
    0.923 at ef 128, 0.864 vs 0.844 at ef 64: profiles/r02/README.md, scripts/recall_table.py.)"""
    from rad_amd.device import DeviceIndex
    n, ndim, M, ef = 300_000, 1024, 8, 64
    X = oracle.synth_rows(0, n, n, ndim, 20260101, 2)
    Q = X[np.random.default_rng(9).integers(0, n, 200)].copy()
    batched = DeviceIndex(ndim, M, 2 * M, ef)
    batched.add_rows(X, seed=777, max_batch=16384)
    h = oracle.Hnsw(ndim, M, 2 * M, ef, seed=777)
    h.add(X, max_batch=1)                                 # sequential insert on the CPU
    g = h.graph()
    seq = DeviceIndex(ndim, M, 2 * M, ef)
    seq.load_vectors(X)
    seq.load_graph(g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
    rb, rs = _recall(batched, Q), _recall(seq, Q)
    assert rb >= rs - 0.03 and rb >= 0.9, (rb, rs)


def test_batched_build_recall_at_1m_rows(gpu, oracle):
    """At 1M rows the 16384-batch build was measured within 0.01 of the sequential insert (0.914 vs 0.923 at ef 128,
    profiles/r02/README.md; the sequential build takes 6 minutes of CPU, so it is not repeated here): the batched
    build alone must stay at that level."""
    from rad_amd.device import DeviceIndex
    n = 1_000_000
    idx = DeviceIndex(1024, 8, 16, 64)
    idx.synth_vectors(n, seed=20260101, mode=2)
    idx.link_resident(seed=777, max_batch=16384)
    Q = np.concatenate([idx.read_vectors(int(r), 1) for r in np.random.default_rng(9).integers(0, n, 256)])
    r = _recall(idx, Q)
    assert r >= 0.90, r
