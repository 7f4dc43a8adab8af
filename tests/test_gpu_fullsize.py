"""BASELINE-size checks of the traversal (100M x 1024-bit rows resident in HBM, n_to_score = 100k):
size-independent properties, and full oracle parity of 48 traversals at 100M (the corpus and graph are
copied back to the host: 20 GB) and of 3 at 20M."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def big(gpu):
    from rad_amd.device import DeviceIndex
    idx = DeviceIndex(1024, 8, 16, 64)
    idx.synth_vectors(100_000_000, seed=20260101, mode=1)
    idx.synth_graph(seed=777)
    return idx


def _run(idx, Q, nts, want_kernel=None):
    from rad_amd.device import DeviceTraversal
    t = DeviceTraversal(idx, Q, nts)
    if want_kernel:
        assert t.kernel == want_kernel
    assert t.run() == 0
    out = [t.results(i) for i in range(Q.shape[0])]
    st = t.stats()
    t.close()
    return out, st


def test_fullsize_properties(big, monkeypatch):
    monkeypatch.setenv("RADHIP_TRAV", "4")   # the bench kernel (a full batch selects it by itself)
    n, nts = 100_000_000, 100_000
    rng = np.random.default_rng(11)
    Q = np.concatenate([big.read_vectors(int(r), 1) for r in rng.integers(0, n, 7)])
    res, st = _run(big, Q, nts)
    assert (st.status == 1).all() and (st.n_scored >= nts).all() and (st.n_scored < nts + 16).all()
    for i, (s, a, o) in enumerate(res):
        # P1: a node is scored once; slots are valid
        assert np.unique(s).size == s.size and int(s.max()) < n
        # P2: the stored integer scores equal an independent gather-Tanimoto (K2) of the same rows
        ga, go = big.gather(Q[i:i + 1], s, np.array([0, s.size], np.uint64))
        assert np.array_equal(ga, a) and np.array_equal(go, o)
        assert (a <= o).all()
    # P3: stopping earlier yields a prefix of the longer traversal
    short, st2 = _run(big, Q[:3], 30_000)
    for i in range(3):
        k = short[i][0].size
        assert 30_000 <= k < 30_016
        for x, y in zip(short[i], res[i]):
            assert np.array_equal(x, y[:k])
    # P4: determinism, and independence from the position / size of the batch a query rides in
    again, _ = _run(big, Q[[4, 0, 0, 2, 6]], nts)
    for j, i in enumerate([4, 0, 0, 2, 6]):
        for x, y in zip(again[j], res[i]):
            assert np.array_equal(x, y)
    # P5: the one-traversal-per-wave kernel gives the same answer as the four-per-wave kernel
    monkeypatch.setenv("RADHIP_TRAV", "1")
    other, _ = _run(big, Q[:2], nts, "trav_kernel")
    for i in range(2):
        for x, y in zip(other[i], res[i]):
            assert np.array_equal(x, y)


def test_oracle_parity_at_20m(gpu, oracle):
    from rad_amd.device import DeviceIndex, DeviceTraversal
    n, nts = 20_000_000, 100_000
    idx = DeviceIndex(1024, 8, 16, 64)
    idx.synth_vectors(n, seed=5, mode=1)
    idx.synth_graph(seed=6)
    X = np.empty((n, 128), np.uint8)
    for f in range(0, n, 5_000_000):
        X[f:f + 5_000_000] = idx.read_vectors(f, 5_000_000)
    levels, adj0, upper_row, adjU = idx.read_graph()
    inf = idx.info()
    g = oracle.Graph(n, 16, 8, int(inf.max_level), int(inf.entry), levels, adj0, upper_row, adjU)
    Q = X[[123, 19_999_999, 7_654_321]].copy()
    t = DeviceTraversal(idx, Q, nts, log_pops=True)
    assert t.run() == 0
    for i in range(3):
        want = oracle.rad_traverse(g, X, Q[i], nts)
        s, a, o = t.results(i)
        nodes, lv = t.pop_log(i)
        assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(lv, want.pop_levels)
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)


def test_oracle_parity_at_100m(big, oracle, monkeypatch):
    """Bit-exact parity with the oracle AT the bench configuration (100M rows, n_to_score = 100k):
    the corpus and the graph are copied back from HBM (20 GB of host memory) and 48 traversals are
    compared in full — expansion order, scored order, integer counts — for the four-per-wave
    kernel, 6 of them for the one-per-wave kernel too."""
    from rad_amd.device import DeviceTraversal
    n, nts = 100_000_000, 100_000
    X = np.empty((n, 128), np.uint8)
    for f in range(0, n, 10_000_000):
        X[f:f + 10_000_000] = big.read_vectors(f, 10_000_000)
    levels, adj0, upper_row, adjU = big.read_graph()
    inf = big.info()
    g = oracle.Graph(n, 16, 8, int(inf.max_level), int(inf.entry), levels, adj0, upper_row, adjU)
    rng = np.random.default_rng(2026)
    rows = rng.integers(0, n, 48)
    Q = X[rows].copy()
    want = [oracle.rad_traverse(g, X, Q[i], nts) for i in range(Q.shape[0])]

    def check(t, idxs):
        for j, i in enumerate(idxs):
            s, a, o = t.results(j)
            nodes, lv = t.pop_log(j)
            assert np.array_equal(nodes, want[i].pop_nodes) and np.array_equal(lv, want[i].pop_levels), i
            assert np.array_equal(s, want[i].slots) and np.array_equal(a, want[i].and_cnt) and np.array_equal(o, want[i].or_cnt), i

    monkeypatch.setenv("RADHIP_TRAV", "4")
    t = DeviceTraversal(big, Q, nts, log_pops=True)
    assert t.kernel == "trav4_kernel" and t.run() == 0
    check(t, range(48))
    t.close()
    monkeypatch.setenv("RADHIP_TRAV", "1")
    sel = [0, 7, 13, 21, 34, 47]
    t1 = DeviceTraversal(big, Q[sel], nts, log_pops=True)
    assert t1.run() == 0
    check(t1, sel)
    t1.close()


def test_bench_workload_itself_full_results_vs_oracle(gpu, oracle, monkeypatch):
    """The bench workload itself: 100M hierarchical rows, the HNSW graph BUILT on the GPU with the
    bench's parameters (connectivity 8, expansion_add 64, batches of 16384, rows linked where they are resident),
    n_to_score = 100k — 16 traversals compared with the oracle in full: expansion order, scored order, both counts;
    plus the order-sensitive result hash the bench's parity sample uses."""
    from rad_amd.device import DeviceIndex, DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "4")
    monkeypatch.delenv("RADHIP_TABLE", raising=False)
    n, nts = 100_000_000, 100_000
    idx = DeviceIndex(1024, 8, 16, 64)
    idx.synth_vectors(n, seed=20260101, mode=2)
    idx.link_resident(seed=777, max_batch=16384)
    X = np.empty((n, 128), np.uint8)
    for f in range(0, n, 10_000_000):
        X[f:f + 10_000_000] = idx.read_vectors(f, 10_000_000)
    levels, adj0, upper_row, adjU = idx.read_graph()
    inf = idx.info()
    g = oracle.Graph(n, 16, 8, int(inf.max_level), int(inf.entry), levels, adj0, upper_row, adjU)
    rows = np.random.default_rng(77).integers(0, n, 16)
    Q = X[rows].copy()
    t = DeviceTraversal(idx, Q, nts, log_pops=True)
    assert t.kernel == "trav4_kernel" and t.table == "bucket" and t.run() == 0
    hashes = t.result_hashes()
    st = t.stats()
    for i in range(16):
        want = oracle.rad_traverse(g, X, Q[i], nts)
        s, a, o = t.results(i)
        nodes, lv = t.pop_log(i)
        assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(lv, want.pop_levels), i
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt), i
        assert int(hashes[i]) == oracle.result_hash(want.slots, want.and_cnt, want.or_cnt), i
        assert st.n_pops[i] == want.n_pops and st.n_nbr[i] == want.n_nbr
    assert (st.n_remid > 0).all()
    t.close()
    # a batch larger than the device holds resident: the rows of the persistent wavefronts take its traversals from a
    # counter — the LAST ones of the batch are taken over by rows that have finished others (hash of the whole scored list +
    # the three counters against the oracle)
    nqb = 2 * idx.traversal_capacity() + 4096
    Qb = X[np.random.default_rng(78).integers(0, n, nqb)].copy()
    tb = DeviceTraversal(idx, Qb, nts)
    assert tb.kernel == "trav4_kernel" and tb.run() == 0
    hb = tb.result_hashes(nqb - 6, 6)
    sb = tb.stats()
    for j in range(6):
        i = nqb - 6 + j
        want = oracle.rad_traverse(g, X, Qb[i], nts)
        assert int(hb[j]) == oracle.result_hash(want.slots, want.and_cnt, want.or_cnt), i
        assert sb.n_pops[i] == want.n_pops and sb.n_nbr[i] == want.n_nbr and sb.n_scored[i] == len(want.slots)
    tb.close()
    idx.close()
