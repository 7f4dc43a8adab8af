"""GPU parity tests (through the C ABI) of K1 scan, K2 gather, the synthetic
generators and the K3 RAD traversal against the CPU oracle.  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk_index(ndim, M, cap0=0):
    from rad_amd.device import DeviceIndex
    return DeviceIndex(ndim, M, cap0, 64)


@pytest.mark.parametrize("ndim,mode", [(1024, 1), (1024, 0), (2048, 1), (64, 0), (512, 1), (100, 0)])
def test_synth_rows_match_oracle(gpu, oracle, ndim, mode):
    idx = _mk_index(ndim, 8)
    n = 5000
    idx.synth_vectors(n, seed=11, mode=mode)
    got = idx.read_vectors(0, n)
    want = oracle.synth_rows(0, n, n, ndim, 11, mode)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n,M,cap0", [(4096, 8, 16), (20000, 8, 16), (3000, 4, 8), (50000, 16, 32), (70000, 32, 64)])
def test_synth_graph_matches_oracle(gpu, oracle, n, M, cap0):
    idx = _mk_index(1024, M, cap0)
    idx.synth_vectors(n, seed=5, mode=1)
    idx.synth_graph(seed=9)
    levels, adj0, upper_row, adjU = idx.read_graph()
    g = oracle.synth_graph(n, M, cap0, 9)
    assert idx.info().max_level == g.max_level
    assert np.array_equal(levels, g.levels)
    assert np.array_equal(adj0, g.adj0)
    assert np.array_equal(upper_row, g.upper_row)
    assert np.array_equal(adjU, g.adjU)
    assert np.array_equal(idx.get_top_level_nodes(), g.top_level())


@pytest.mark.parametrize("ndim", [64, 256, 1024, 2048, 1000])
@pytest.mark.parametrize("nq", [1, 3, 5, 6, 7, 8, 11])
def test_scan_matches_oracle(gpu, oracle, ndim, nq):
    rng = np.random.default_rng(ndim + nq)
    n = 3001
    rb = (ndim + 7) // 8
    X = rng.integers(0, 256, (n, rb), dtype=np.uint8)
    if ndim % 8:
        X[:, -1] &= (1 << (ndim % 8)) - 1
    X[7] = 0  # all-zero row: or may be 0 against an all-zero query
    Q = X[rng.integers(0, n, nq)].copy()
    Q[0] = 0
    idx = _mk_index(ndim, 8)
    idx.load_vectors(X)
    a, o = idx.scan(Q)
    for i in range(nq):
        wa, wo = oracle.scan(X, Q[i])
        assert np.array_equal(a[i], wa) and np.array_equal(o[i], wo)
    # sub-range
    a2, o2 = idx.scan(Q[:2], first=100, count=777)
    assert np.array_equal(a2, a[:2, 100:877]) and np.array_equal(o2, o[:2, 100:877])


@pytest.mark.parametrize("ndim", [64, 1024, 2048])
def test_gather_matches_oracle(gpu, oracle, ndim):
    rng = np.random.default_rng(ndim)
    n, nq = 4000, 5
    X = rng.integers(0, 256, (n, (ndim + 7) // 8), dtype=np.uint8)
    Q = X[:nq]
    sizes = [0, 1, 17, 256, 1000]
    slots = np.concatenate([rng.integers(0, n, s) for s in sizes]).astype(np.uint32)
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint64)
    idx = _mk_index(ndim, 8)
    idx.load_vectors(X)
    a, o = idx.gather(Q, slots, off)
    for i in range(nq):
        wa, wo = oracle.gather(X, Q[i], slots[int(off[i]):int(off[i + 1])])
        assert np.array_equal(a[int(off[i]):int(off[i + 1])], wa)
        assert np.array_equal(o[int(off[i]):int(off[i + 1])], wo)


def _check_traversal(oracle, idx, graph, X, Q, n_to_score):
    from rad_amd.device import DeviceTraversal
    t = DeviceTraversal(idx, Q, n_to_score, log_pops=True)
    assert t.run() == 0
    st = t.stats()
    for i in range(Q.shape[0]):
        want = oracle.rad_traverse(graph, X, Q[i], n_to_score)
        s, a, o = t.results(i)
        nodes, levels = t.pop_log(i)
        assert st.n_pops[i] == want.n_pops, (i, st.n_pops[i], want.n_pops)
        assert np.array_equal(nodes, want.pop_nodes), f"traversal {i}: expansion order differs"
        assert np.array_equal(levels, want.pop_levels)
        assert np.array_equal(s, want.slots), f"traversal {i}: scored order differs"
        assert np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
        assert st.n_nbr[i] == want.n_nbr
        assert st.status[i] in (1, 2)
    t.close()


@pytest.mark.parametrize("grid,static", [(1, "0"), (2, "0"), (3, "1")])
@pytest.mark.parametrize("table", ["bucket", "hash", "group", "local"])
def test_rows_take_traversals_from_the_counter(gpu, oracle, monkeypatch, grid, static, table):
    """trav4_kernel launches as many wavefronts as the device holds resident and their rows take the traversals of the batch
    one after the other; a grid of one or two wavefronts (test hook) makes every row work through many traversals of
    different lengths — and resume them in rounds — with the same results as one row per traversal."""
    from rad_amd.device import DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "4")
    monkeypatch.setenv("RADHIP_TRAV_STATIC", static)
    if table != "bucket":
        monkeypatch.setenv("RADHIP_TABLE", table)
    n, nq = 6000, 37
    idx = _mk_index(1024, 8, 16)
    idx.synth_vectors(n, seed=5, mode=2)
    idx.link_resident(seed=9, max_batch=512)
    X = idx.read_vectors(0, n)
    levels, adj0, upper_row, adjU = idx.read_graph()
    inf = idx.info()
    g = oracle.Graph(n, 16, 8, int(inf.max_level), int(inf.entry), levels, adj0, upper_row, adjU)
    rng = np.random.default_rng(3)
    Q = X[rng.integers(0, n, nq)].copy()
    Q[3] = 0
    nts = 900
    want = [oracle.rad_traverse(g, X, Q[i], nts) for i in range(nq)]
    monkeypatch.setenv("RADHIP_TEST_GRID", str(grid))
    t = DeviceTraversal(idx, Q, nts, log_pops=True)
    assert t.kernel == "trav4_kernel"
    if static == "0":   # in rounds: every launch, every row lets go of and takes up traversals that are half done
        rounds = 0
        while t.run(max_pops=40) and rounds < 200:
            rounds += 1
        assert rounds >= 3
    else:
        assert t.run() == 0
    st = t.stats()
    for i in range(nq):
        s, a, o = t.results(i)
        nodes, lv = t.pop_log(i)
        assert np.array_equal(nodes, want[i].pop_nodes) and np.array_equal(lv, want[i].pop_levels), i
        assert np.array_equal(s, want[i].slots) and np.array_equal(a, want[i].and_cnt) and np.array_equal(o, want[i].or_cnt), i
        assert st.n_pops[i] == want[i].n_pops and st.n_nbr[i] == want[i].n_nbr
    t.close()


@pytest.mark.parametrize("ndim,n,M,cap0,n_to_score", [
    (1024, 4096, 8, 16, 1000),
    (1024, 30000, 8, 16, 5000),
    (2048, 9000, 32, 64, 3000),
    (64, 1000, 4, 8, 300),
    (1024, 5000, 8, 16, 5000),      # score everything: queue drains
])
def test_traversal_synthetic_graph(gpu, oracle, trav_mode, ndim, n, M, cap0, n_to_score):
    idx = _mk_index(ndim, M, cap0)
    idx.synth_vectors(n, seed=3, mode=1)
    idx.synth_graph(seed=4)
    X = oracle.synth_rows(0, n, n, ndim, 3, 1)
    g = oracle.synth_graph(n, M, cap0, 4)
    Q = X[[0, 17, n // 2, n - 1, 5, 6, 7]].copy()
    _check_traversal(oracle, idx, g, X, Q, n_to_score)


@pytest.mark.parametrize("ndim,n,M,ef", [(1024, 3000, 8, 64), (64, 1000, 4, 20)])
def test_traversal_built_graph(gpu, oracle, trav_mode, ndim, n, M, ef):
    """Graph built by the oracle's usearch-shaped builder, loaded via load_graph."""
    X = oracle.synth_rows(0, n, n, ndim, 21, 1 if ndim >= 512 else 0)
    h = oracle.Hnsw(ndim, M, 2 * M, ef, seed=7)
    h.add(X, max_batch=1)
    g = h.graph()
    idx = _mk_index(ndim, M, 2 * M)
    idx.load_vectors(X)
    idx.load_graph(g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
    rng = np.random.default_rng(1)
    Q = np.concatenate([X[:4], rng.integers(0, 256, (3, X.shape[1]), dtype=np.uint8)])
    _check_traversal(oracle, idx, g, X, Q, 700)
    # adjacency reads through the C ABI
    for slot in (0, 1, n - 1, g.entry):
        for lv in range(int(g.levels[slot]) + 1):
            assert np.array_equal(idx.get_neighbors(slot, lv), g.neighbors(slot, lv))
    assert np.array_equal(idx.get_top_level_nodes(), g.top_level())


def test_traversal_resume_in_rounds(gpu, oracle, trav_mode):
    """Bounded rounds (max_pops) must give the same result as one run."""
    from rad_amd.device import DeviceTraversal
    n, ndim = 20000, 1024
    idx = _mk_index(ndim, 8, 16)
    idx.synth_vectors(n, seed=3, mode=1)
    idx.synth_graph(seed=4)
    X = oracle.synth_rows(0, n, n, ndim, 3, 1)
    Q = X[:9].copy()
    t1 = DeviceTraversal(idx, Q, 4000)
    assert t1.run() == 0
    t2 = DeviceTraversal(idx, Q, 4000)
    rounds = 0
    while t2.run(max_pops=37) > 0:
        rounds += 1
        assert rounds < 10000
    assert rounds > 3
    for i in range(Q.shape[0]):
        for x, y in zip(t1.results(i), t2.results(i)):
            assert np.array_equal(x, y)
    # reset re-arms the same state
    t2.reset(Q[::-1].copy())
    assert t2.run() == 0
    for i in range(Q.shape[0]):
        for x, y in zip(t1.results(i), t2.results(Q.shape[0] - 1 - i)):
            assert np.array_equal(x, y)


def test_device_keys_match_host_restatement(gpu):
    """The device's float-reciprocal q and comparison-tree key (common.h *_dev) against the host
    restatement that tests/test_key_order.py pins to the Redis order: every (and, or) with
    or <= 2048, and slots around every power of ten."""
    from rad_amd import _lib
    from rad_amd._lib import check, ptr
    L = _lib.lib()
    idx = _mk_index(1024, 8)
    ors = np.arange(0, 2049, dtype=np.uint32)
    a = np.concatenate([np.arange(0, o + 1, dtype=np.uint32) for o in ors])
    o = np.concatenate([np.full(o + 1, o, dtype=np.uint32) for o in ors])
    rng = np.random.default_rng(0)
    edge = np.array([0, 1, 9, 10, 11, 99, 100, 999, 1000, 9999, 10000, 99999, 100000, 999999, 1000000,
                     9999999, 10000000, 99999999, 100000000, 999999999], np.uint32)
    slot = np.concatenate([edge, rng.integers(0, 1_000_000_000, a.size - edge.size).astype(np.uint32)])
    level = rng.integers(0, 16, a.size).astype(np.uint32)
    out = np.empty(a.size, np.uint64)
    check(L.radhip_debug_device_keys(idx._h, ptr(a), ptr(o), ptr(slot), ptr(level), a.size, ptr(out)))
    q = np.where(o > 0, ((o.astype(np.int64) - a.astype(np.int64)) << 23) // np.maximum(o.astype(np.int64), 1), 0)
    assert np.array_equal(out >> np.uint64(38), q.astype(np.uint64))
    for i in np.concatenate([np.arange(edge.size), rng.integers(0, a.size, 5000)]):
        assert int(out[i]) == L.radhip_rad_key(int(a[i]), int(o[i]), int(slot[i]), int(level[i]))
    assert idx.traversal_capacity() >= 256 * 8


def test_staging_sort_network(gpu):
    """trav4_kernel's register-resident 256-slot bitonic network (4 keys per lane, DPP / ds_bpermute exchanges)
    against numpy on every fill level, with duplicates, extremes and junk behind the valid keys."""
    from rad_amd import _lib
    from rad_amd._lib import check, ptr
    L = _lib.lib()
    idx = _mk_index(1024, 8)
    cap = int(L.radhip_debug_staging_capacity())
    assert 64 <= cap <= 256
    rng = np.random.default_rng(5)
    counts = np.array(list(range(1, cap + 1)) + [cap] * 64 + [min(c, cap) for c in (1, 2, 3, 63, 64, 65, 127, 128, 129)], np.uint32)
    keys = rng.integers(0, 1 << 62, (counts.size, 256), dtype=np.uint64)
    keys[5::7] >>= np.uint64(40)                                   # many equal high words: the low word decides
    keys[3::11, ::2] = keys[3::11, 1::2]                           # exact duplicates
    keys[cap + 1, :8] = np.array([0, 1, (1 << 62) - 1, 0xFFFFFFFF, 0x100000000, 0xFFFFFFFFFFFFFFFE, 0, 1], np.uint64)
    out = np.empty_like(keys)
    check(L.radhip_debug_sort_staging(idx._h, ptr(keys), ptr(counts), counts.size, ptr(out)))
    for b, n in enumerate(counts.tolist()):
        want = np.sort(keys[b, :n])
        assert np.array_equal(out[b, :n], want), (b, n)
        assert (out[b, n:cap] == np.uint64(0xFFFFFFFFFFFFFFFF)).all(), (b, n)


@pytest.mark.parametrize("kernel", ["trav4", "trav1"])
def test_traversal_deep_queue_paths(gpu, oracle, kernel, monkeypatch):
    """n_to_score large enough that the pivot queue flushes hundreds of sorted runs and re-pivots
    hundreds of times (bench-scale code paths), for both traversal kernels, still bit-exact."""
    from rad_amd.device import DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "1" if kernel == "trav1" else "4")
    n, ndim, M, cap0, nts = 400_000, 1024, 8, 16, 60_000
    idx = _mk_index(ndim, M, cap0)
    idx.synth_vectors(n, seed=13, mode=1)
    idx.synth_graph(seed=14)
    X = oracle.synth_rows(0, n, n, ndim, 13, 1)
    g = oracle.synth_graph(n, M, cap0, 14)
    Q = X[[11, 70_001, 399_999, 5, 123_456]].copy()
    t = DeviceTraversal(idx, Q, nts, log_pops=True)
    assert t.run() == 0
    st = t.stats()
    assert st.n_flush.min() > 50 and st.n_repivot.min() > 50
    for i in range(Q.shape[0]):
        want = oracle.rad_traverse(g, X, Q[i], nts)
        s, a, o = t.results(i)
        nodes, levels = t.pop_log(i)
        assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(levels, want.pop_levels)
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
    t.close()
    # the same in bounded rounds (state persisted / restored between launches)
    t2 = DeviceTraversal(idx, Q[:2], nts)
    rounds = 0
    while t2.run(max_pops=1500) > 0:
        rounds += 1
        assert rounds < 1000
    for i in range(2):
        want = oracle.rad_traverse(g, X, Q[i], nts, log_pops=False)
        s, a, o = t2.results(i)
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt)


@pytest.mark.parametrize("kernel,ndim,M,cap0", [("trav4", 1024, 8, 16), ("trav1", 1024, 8, 16), ("trav1", 2048, 32, 64)])
def test_traversal_crowded_visited_table(gpu, oracle, kernel, ndim, M, cap0, monkeypatch):
    """Thousands of short traversals whose visited tables end half full (2 * (n_to_score + 64 + n_top)
    just fits the smallest table of 1024 buckets): lanes of one expansion keep meeting on the same
    and on neighbouring empty buckets.  Regression test of the in-expansion claim protocol — a lane
    that loses bucket h and walks on to h+1 must lose h+1 as well when another lane of the same
    expansion owns it already, or a visited entry is overwritten and its node scored twice."""
    from rad_amd.device import DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "1" if kernel == "trav1" else "4")
    n, nq = 120_000, 4096
    idx = _mk_index(ndim, M, cap0)
    idx.synth_vectors(n, seed=31, mode=1)
    idx.synth_graph(seed=32)
    X = oracle.synth_rows(0, n, n, ndim, 31, 1)
    g = oracle.synth_graph(n, M, cap0, 32)
    n_top = len(idx.get_top_level_nodes())
    nts = 512 - 64 - n_top
    assert nts > 200
    rng = np.random.default_rng(5)
    Q = X[rng.choice(n, nq, replace=False)].copy()
    t = DeviceTraversal(idx, Q, nts, log_pops=True)
    assert t.run() == 0
    st = t.stats()
    bad = []
    for i in range(nq):
        want = oracle.rad_traverse(g, X, Q[i], nts)
        s, a, o = t.results(i)
        nodes, levels = t.pop_log(i)
        ok = (len(np.unique(s)) == len(s) and np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt)
              and np.array_equal(o, want.or_cnt) and np.array_equal(nodes, want.pop_nodes)
              and np.array_equal(levels, want.pop_levels) and st.n_nbr[i] == want.n_nbr)
        if not ok:
            bad.append(i)
    t.close()
    assert not bad, f"{len(bad)} of {nq} traversals differ from the oracle, first {bad[:8]}"


def test_reset_reuses_tables_across_epochs(gpu, oracle, trav_mode):
    """reset() re-arms the state without clearing the visited tables (epoch tags): many batches in a
    row — past the epoch wrap-around — must each match the oracle, on levels 0 and above."""
    from rad_amd.device import DeviceTraversal
    n = 6000
    X = oracle.synth_rows(0, n, n, 1024, 31, 1)
    g = oracle.synth_graph(n, 8, 16, 32)
    idx = _mk_index(1024, 8, 16)
    idx.load_vectors(X)
    idx.load_graph(g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
    rng = np.random.default_rng(5)
    t = DeviceTraversal(idx, X[:6], 400)
    for batch in range(140):                       # EPOCH_LIMIT is 127
        Q = X[rng.integers(0, n, 6)].copy()
        if batch:
            t.reset(Q)
        else:
            t.reset(Q)
        assert t.run() == 0
        if batch % 9 == 0 or batch > 120:
            for i in (0, 5):
                want = oracle.rad_traverse(g, X, Q[i], 400, log_pops=False)
                s, a, o = t.results(i)
                assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt), (batch, i)


def test_kernel_choice_follows_batch_size(gpu, monkeypatch):
    """Auto dispatch: one traversal per wavefront while the whole batch is resident with that
    kernel, four per wavefront beyond; wide rows always one per wavefront; env overrides."""
    from rad_amd.device import DeviceTraversal
    monkeypatch.delenv("RADHIP_TRAV", raising=False)
    monkeypatch.delenv("RADHIP_NO_TRAV4", raising=False)
    idx = _mk_index(1024, 8, 16)
    idx.synth_vectors(50_000, seed=1, mode=1)
    idx.synth_graph(seed=2)
    cap4 = idx.traversal_capacity()
    assert cap4 % 4 == 0 and cap4 >= 1024
    Q = idx.read_vectors(0, 4)
    t = DeviceTraversal(idx, Q, 100)
    assert t.kernel == "trav_kernel"
    t.close()
    big = idx.read_vectors(0, cap4)
    t = DeviceTraversal(idx, big, 100)
    assert t.kernel == "trav4_kernel"
    assert t.run() == 0
    t.close()
    monkeypatch.setenv("RADHIP_TRAV", "4")
    t = DeviceTraversal(idx, Q, 100)
    assert t.kernel == "trav4_kernel"
    t.close()
    monkeypatch.setenv("RADHIP_TRAV", "1")
    t = DeviceTraversal(idx, big[:cap4 // 2], 100)
    assert t.kernel == "trav_kernel"
    t.close()
    monkeypatch.delenv("RADHIP_TRAV")
    wide = _mk_index(1024, 32, 64)
    wide.synth_vectors(50_000, seed=1, mode=1)
    wide.synth_graph(seed=2)
    t = DeviceTraversal(wide, wide.read_vectors(0, 4), 100)
    assert t.kernel == "trav_kernel"
    t.close()


def _brute_topk(oracle, X, q, k, first, count):
    a, o = oracle.scan(X[first:first + count], q)
    qk = ((o.astype(np.int64) - a) << 23) // np.maximum(o.astype(np.int64), 1)
    order = np.lexsort((np.arange(count), qk))[:k]
    return (order + first).astype(np.uint32), a[order], o[order]


@pytest.mark.parametrize("ndim", [64, 200, 1024, 2048])
@pytest.mark.parametrize("k,nq", [(1, 3), (10, 9), (100, 8), (500, 2), (1984, 1)])
def test_topk_matches_brute_force(gpu, oracle, ndim, k, nq):
    """radhip_tanimoto_topk against a brute-force scan in (distance, slot) order: many exact ties
    (duplicate rows, all-zero rows), sub-ranges, k larger than the range."""
    rng = np.random.default_rng(ndim * 7 + k)
    n = 20_011
    rb = (ndim + 7) // 8
    bits = rng.random((n, rb * 8)) < (0.5 if ndim <= 64 else 0.07)
    bits[:, ndim:] = False
    X = np.ascontiguousarray(np.packbits(bits, axis=1))
    X[rng.integers(0, n, 3000)] = X[rng.integers(0, n, 3000)]     # duplicates: equal distances, slot decides
    X[rng.integers(0, n, 50)] = 0
    Q = X[rng.integers(0, n, nq)].copy()
    Q[-1] = 0
    idx = _mk_index(ndim, 8)
    idx.load_vectors(X)
    for first, count in ((0, n), (777, 9000), (n - 37, 37), (5, 1)):
        s, a, o, c = idx.topk(Q, k, first, count)
        for i in range(nq):
            ws, wa, wo = _brute_topk(oracle, X, Q[i], k, first, count)
            m = ws.size
            assert c[i] == m
            assert np.array_equal(s[i, :m], ws), (ndim, k, i, first, count)
            assert np.array_equal(a[i, :m], wa) and np.array_equal(o[i, :m], wo)
            assert (s[i, m:] == 0xFFFFFFFF).all() and (a[i, m:] == 0).all()
    with pytest.raises(Exception):
        idx.topk(Q, 0)
    with pytest.raises(Exception):
        idx.topk(Q, 5000)
