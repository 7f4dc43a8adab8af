"""Edge cases of the device path against the oracle: tiny and degenerate graphs, ragged batch
sizes, single-level indexes (every node is a top-level node), all-zero fingerprints, odd
dimensions, n_to_score larger than the index, capacity errors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NO_SLOT = 0xFFFFFFFF


def _check(oracle, idx, g, X, Q, nts):
    from rad_amd.device import DeviceTraversal
    t = DeviceTraversal(idx, Q, nts, log_pops=True)
    assert t.run() == 0
    for i in range(Q.shape[0]):
        want = oracle.rad_traverse(g, X, Q[i], min(nts, g.n))
        s, a, o = t.results(i)
        nodes, levels = t.pop_log(i)
        assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(levels, want.pop_levels), i
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt), i
    t.close()


def _dev(X, g, M, cap0):
    from rad_amd.device import DeviceIndex
    idx = DeviceIndex(X.shape[1] * 8 if True else 0, M, cap0, 32)
    idx.load_vectors(X)
    idx.load_graph(g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
    return idx


@pytest.mark.parametrize("n", [1, 2, 3, 17])
def test_tiny_indexes_built_on_gpu(gpu, oracle, trav_mode, n):
    from rad_amd.index import Index
    X = oracle.synth_rows(0, n, n, 1024, 2, 1)
    h = oracle.Hnsw(1024, 8, 16, 32, seed=9)
    h.add(X, max_batch=4)
    g = h.graph()
    idx = Index(ndim=1024, connectivity=8, expansion_add=32, seed=9, max_batch=4)
    idx.add(np.arange(n), X)
    lv, a0, ur, aU = idx.device_index().read_graph()
    assert np.array_equal(lv, g.levels) and np.array_equal(a0, g.adj0) and np.array_equal(aU, g.adjU)
    _check(oracle, idx.device_index(), g, X, X[:min(n, 3)], 10)
    m = idx.search(X[:1], count=5)
    assert int(m.counts[0]) == min(5, n) and int(m.slots[0, 0]) == 0


@pytest.mark.parametrize("nq", [1, 2, 3, 5, 7, 9])
def test_ragged_batch_sizes(gpu, oracle, trav_mode, nq):
    """Batches that do not fill the four traversal rows of a wavefront."""
    n = 3000
    X = oracle.synth_rows(0, n, n, 1024, 4, 1)
    g = oracle.synth_graph(n, 8, 16, 5)
    idx = _dev(X, g, 8, 16)
    _check(oracle, idx, g, X, X[100:100 + nq].copy(), 500)


def test_single_level_index_every_node_is_top_level(gpu, oracle, trav_mode):
    """max_level == 0: get_top_level_nodes returns every node, prime scores all of them and the
    traversal starts on level 0 (rad/traverser.py:157 start level = max(0, max_level - 1))."""
    n, M, cap0 = 150, 8, 16
    rng = np.random.default_rng(0)
    X = oracle.synth_rows(0, n, n, 1024, 6, 1)
    adj0 = np.full((n, cap0), NO_SLOT, np.uint32)
    for i in range(n):
        nb = rng.choice(np.delete(np.arange(n), i), size=rng.integers(0, cap0 + 1), replace=False)
        adj0[i, :nb.size] = nb                                   # ragged rows, some empty
    g = oracle.Graph(n, cap0, M, 0, 0, np.zeros(n, np.int8), adj0, np.full(n, NO_SLOT, np.uint32), np.empty((0, M), np.uint32))
    idx = _dev(X, g, M, cap0)
    assert idx.get_top_level_nodes().tolist() == list(range(n))
    for nts in (10, 149, 150, 10_000):                           # n_to_score above the index size
        _check(oracle, idx, g, X, X[[0, 77]].copy(), nts)


def test_empty_rows_and_upper_levels(gpu, oracle, trav_mode):
    """Nodes whose adjacency row is empty on some level still descend (stated deviation)."""
    n, M, cap0 = 64, 4, 8
    X = oracle.synth_rows(0, n, n, 256, 8, 0)
    levels = np.zeros(n, np.int8)
    levels[[0, 5, 9]] = 2
    levels[[1, 2, 3, 30]] = 1
    upper_row = np.full(n, NO_SLOT, np.uint32)
    rows = 0
    for i in range(n):
        if levels[i] > 0:
            upper_row[i] = rows
            rows += int(levels[i])
    adjU = np.full((rows, M), NO_SLOT, np.uint32)
    adjU[upper_row[0] + 1, :2] = [5, 9]          # level 2 of node 0
    adjU[upper_row[5] + 1, :1] = [0]             # level 2 of node 5;  node 9 has an EMPTY level-2 row
    adjU[upper_row[0], :3] = [1, 2, 5]           # level 1
    adjU[upper_row[5], :2] = [3, 30]
    adjU[upper_row[1], :1] = [0]                 # nodes 2, 3, 9, 30 have empty level-1 rows
    adj0 = np.full((n, cap0), NO_SLOT, np.uint32)
    for i in range(0, n, 2):                     # odd nodes have empty level-0 rows
        adj0[i, :4] = [(i + 1) % n, (i + 2) % n, (i + 7) % n, (i + 20) % n]
    g = oracle.Graph(n, cap0, M, 2, 0, levels, adj0, upper_row, adjU)
    idx = _dev(X, g, M, cap0)
    _check(oracle, idx, g, X, X[:6].copy(), 64)


@pytest.mark.parametrize("ndim", [100, 1000, 1536, 2048])
def test_odd_dimensions_and_zero_vectors(gpu, oracle, trav_mode, ndim):
    n = 2500
    rb = (ndim + 7) // 8
    rng = np.random.default_rng(ndim)
    X = (rng.random((n, rb * 8)) < 0.08)
    X[:, ndim:] = False
    X = np.packbits(X.astype(np.uint8), axis=1, bitorder="little")[:, :rb]
    X[7] = 0                                      # all-zero rows: or == 0 against an all-zero query
    X[8] = 0
    g = oracle.synth_graph(n, 8, 16, 3)
    idx = _dev(X, g, 8, 16)
    Q = X[[0, 7, 99]].copy()                      # includes the all-zero query
    _check(oracle, idx, g, X, Q, 600)
    a, o = idx.scan(Q[1:2])
    assert int(o[0, 8]) == 0 and int(a[0, 8]) == 0


def test_invalid_arguments_are_rejected(gpu, oracle):
    from rad_amd._lib import RadHipError
    from rad_amd.device import DeviceIndex, DeviceTraversal
    idx = DeviceIndex(1024, 8, 16, 32)
    with pytest.raises(RadHipError):
        DeviceTraversal(idx, np.zeros((1, 128), np.uint8), 10)      # no vectors / graph yet
    X = oracle.synth_rows(0, 500, 500, 1024, 1, 1)
    g = oracle.synth_graph(500, 8, 16, 1)
    idx.load_vectors(X)
    idx.load_graph(g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
    with pytest.raises(ValueError):
        DeviceTraversal(idx, np.zeros((1, 64), np.uint8), 10)       # wrong row width
    with pytest.raises(RadHipError):
        idx.get_neighbors(500, 0)
    with pytest.raises(RadHipError):
        idx.gather(X[:1], np.array([10_000], np.uint32), np.array([0, 1], np.uint64))
    with pytest.raises(RadHipError):
        DeviceIndex(4096, 8)                                        # ndim > 2048


def test_search_rows_shorter_than_k_are_padded(gpu, oracle):
    """A query whose graph search reaches fewer than k nodes gets counts < k and a row padded with
    NO_SLOT / zero counts (once the tail was whatever the device buffer held)."""
    from rad_amd.index import Index
    n, ndim, M = 40, 64, 2
    X = oracle.synth_rows(0, n, n, ndim, 3, 0)
    # two components: nodes 0..19 form a ring, 20..39 another; entry in the first
    levels = np.zeros(n, np.int8)
    adj0 = np.full((n, 2 * M), NO_SLOT, np.uint32)
    for i in range(n):
        base = 0 if i < 20 else 20
        adj0[i, :2] = [base + (i - base + 1) % 20, base + (i - base + 19) % 20]
    idx = Index(ndim=ndim, connectivity=M, connectivity_base=2 * M)
    idx.load_graph(None, X, levels, adj0, np.full(n, NO_SLOT, np.uint32), np.full((1, M), NO_SLOT, np.uint32), 0, 0)
    for _ in range(3):   # repeated: the tail must not depend on what earlier calls left in device memory
        m = idx.search(X[:3], count=30, expansion=64)
        assert (m.counts == 20).all()
        assert (m.slots[:, 20:] == NO_SLOT).all() and (m.slots[:, :20] < 20).all()
        assert (m.keys[:, 20:] == 0).all()


def test_add_survives_visited_table_overflow_and_stays_extendable(gpu, oracle):
    """A corpus of near-duplicates makes an insert's layer search visit far more nodes than expansion_add
    suggests (equal distances admit every smaller slot): the per-insert visited table overflows.  The library
    re-runs the batch with a larger table instead of failing — the graph is the oracle's, and the index can
    still be extended afterwards (add() used to be left half-committed)."""
    from rad_amd.index import Index
    rng = np.random.default_rng(5)
    n, ndim, M, ef = 9000, 64, 4, 8
    base = np.packbits(rng.integers(0, 2, (3, ndim), dtype=np.uint8), axis=1)
    X = base[rng.integers(0, 3, n)].copy()                    # three distinct fingerprints, 3000 copies each
    X[::7, 0] ^= rng.integers(0, 4, X[::7, 0].shape, dtype=np.uint8)
    h = oracle.Hnsw(ndim, M, 2 * M, ef, seed=3)
    h.add(X[:6000], max_batch=256)
    h.add(X[6000:], max_batch=256)
    g = h.graph()
    idx = Index(ndim=ndim, connectivity=M, expansion_add=ef, seed=3, max_batch=256)
    idx.add(np.arange(6000), X[:6000])
    idx.add(np.arange(6000, n), X[6000:])                     # a second call after whatever the first one hit
    levels, adj0, upper_row, adjU = idx.device_index().read_graph()
    assert len(idx) == n and np.array_equal(levels, g.levels)
    assert np.array_equal(adj0, g.adj0) and np.array_equal(adjU, g.adjU)
    assert idx.get_node_ids_from_keys([0, 8999]).tolist() == [0, 8999]


def test_long_traversal_1_2m_scored_four_per_wavefront(gpu, oracle, monkeypatch):
    """n_to_score = 1.2M in trav4_kernel: ~10^4 staging flushes per traversal.  The run table and the key pool
    are garbage-collected on the device (the 8192-entry run table used to overflow here)."""
    from rad_amd.device import DeviceIndex, DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "4")
    n, nts = 3_000_000, 1_200_000
    idx = DeviceIndex(1024, 8, 16, 64)
    idx.synth_vectors(n, seed=5, mode=1)
    idx.synth_graph(seed=9)
    X = oracle.synth_rows(0, n, n, 1024, 5, 1)
    g = oracle.synth_graph(n, 8, 16, 9)
    Q = X[[7, 1_500_000, 2_999_999]].copy()
    t = DeviceTraversal(idx, Q, nts)
    assert t.kernel == "trav4_kernel" and t.run() == 0
    st = t.stats()
    assert (st.n_flush > 4000).all()
    for i in range(3):
        want = oracle.rad_traverse(g, X, Q[i], nts, log_pops=False)
        s, a, o = t.results(i)
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
        assert st.n_pops[i] == want.n_pops


def test_traversal_that_lives_on_the_upper_levels(gpu, oracle, trav_mode):
    """Most nodes of this graph sit above level 0 and the upper rows are wide: the (node, level >= 1) visited
    set outgrows the estimate it was sized by.  The library re-arms the batch with more room
    instead of failing with RADHIP_E_CAPACITY; results equal the oracle's."""
    import test_gpu_fuzz as F
    from rad_amd.device import DeviceIndex, DeviceTraversal
    rng = np.random.default_rng(12)
    n, M, cap0, max_level = 60000, 16, 16, 6
    g = F._random_graph(oracle, rng, n, M, cap0, max_level, 0.0)
    lv = g.levels.copy()
    X = F._random_rows(rng, n, 1024, 0.0)
    idx = DeviceIndex(1024, M, cap0, 32)
    idx.load_vectors(X)
    idx.load_graph(g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
    Q = X[rng.integers(0, n, 5)].copy()
    nts = 30000
    t = DeviceTraversal(idx, Q, nts, log_pops=True)
    assert t.run() == 0
    for i in range(5):
        want = oracle.rad_traverse(g, X, Q[i], nts)
        s, a, o = t.results(i)
        nodes, levels = t.pop_log(i)
        assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(levels, want.pop_levels)
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
    assert (lv > 0).mean() > 0.3
