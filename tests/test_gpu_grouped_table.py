"""GPU parity of the grouped visited / scored table (2 bits per node, keyed by the index's graph-locality
layout; rad_amd/csrc/traverse4.inc, layout.hip) against the oracle and against the per-slot hash table:
the same pops, the same scored order, the same counts — for the layout the library computes and for
layouts chosen to hurt (every id on one chunk position, ids spread so that every node has its own chunk)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _index(oracle, n, ndim, M, ef, mode, seed=3):
    from rad_amd.device import DeviceIndex
    X = oracle.synth_rows(0, n, n, ndim, seed, mode)
    idx = DeviceIndex(ndim, M, 2 * M, ef, device=0)
    idx.add_rows(X, seed=11, max_batch=512)
    levels, adj0, upper_row, adjU = idx.read_graph()
    inf = idx.info()
    g = oracle.Graph(n, 2 * M, M, int(inf.max_level), int(inf.entry), levels, adj0, upper_row, adjU)
    return idx, X, g


def _check(oracle, idx, X, g, Q, n_to_score, expect_table):
    from rad_amd.device import DeviceTraversal
    t = DeviceTraversal(idx, Q, n_to_score, log_pops=True)
    assert t.table == expect_table and t.kernel == "trav4_kernel"
    assert t.run() == 0
    for i in range(Q.shape[0]):
        want = oracle.rad_traverse(g, X, Q[i], n_to_score)
        s, a, o = t.results(i)
        nodes, levels = t.pop_log(i)
        assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(levels, want.pop_levels), f"pops differ, query {i}"
        assert np.array_equal(s, want.slots), f"scored order differs, query {i}"
        assert np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
    return t


@pytest.mark.parametrize("mode,n,n_to_score", [(2, 20000, 3000), (1, 12000, 2500), (0, 6000, 1500)])
def test_grouped_table_matches_oracle_and_hash_table(gpu, oracle, monkeypatch, mode, n, n_to_score):
    monkeypatch.setenv("RADHIP_TRAV", "4")
    idx, X, g = _index(oracle, n, 1024, 8, 48, mode)
    Q = X[np.random.default_rng(1).integers(0, n, 37)]     # 37: a ragged last wavefront
    monkeypatch.setenv("RADHIP_TABLE", "hash")
    _check(oracle, idx, X, g, Q, n_to_score, "hash")
    monkeypatch.delenv("RADHIP_TABLE")
    _check(oracle, idx, X, g, Q, n_to_score, "bucket")     # the kernel's default table
    info = idx.optimize_layout()
    assert info.valid and info.group == 384 and info.id_limit >= n
    lid = idx.read_layout()
    assert np.unique(lid).size == n and int(lid.max()) < info.id_limit
    monkeypatch.setenv("RADHIP_TABLE", "group")
    t = _check(oracle, idx, X, g, Q, n_to_score, "grouped")
    # a second batch on the same state (epoch bump, no clearing)
    Q2 = X[np.random.default_rng(2).integers(0, n, 37)]
    t.reset(Q2)
    assert t.run() == 0
    for i in (0, 17, 36):
        want = oracle.rad_traverse(g, X, Q2[i], n_to_score)
        s, a, o = t.results(i)
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)


@pytest.mark.parametrize("M,ndim,n,n_to_score", [(16, 1024, 12000, 2500), (32, 2048, 6000, 1500), (12, 512, 8000, 2000)])
def test_grouped_table_on_wide_rows(gpu, oracle, monkeypatch, M, ndim, n, n_to_score):
    """adjacency rows of 24 / 32 / 64 slots: trav4_kernel's WIDE form (a pop walks its row in chunks of 16) with the
    grouped table against the oracle, and against the bucket table of the same form"""
    monkeypatch.setenv("RADHIP_TRAV", "4")
    idx, X, g = _index(oracle, n, ndim, M, 48, 2)
    Q = X[np.random.default_rng(4).integers(0, n, 21)]
    monkeypatch.delenv("RADHIP_TABLE", raising=False)
    _check(oracle, idx, X, g, Q, n_to_score, "bucket")
    monkeypatch.setenv("RADHIP_TABLE", "group")            # (computes the layout on first use)
    _check(oracle, idx, X, g, Q, n_to_score, "grouped")
    # an adversarial layout: every id on one chunk position -> the table fills up, the batch re-runs on the bucket table
    lid = (np.arange(n, dtype=np.uint32) * 384).astype(np.uint32)
    idx.set_layout(lid)
    from rad_amd.device import DeviceTraversal
    t = DeviceTraversal(idx, Q, n_to_score, log_pops=True)
    assert t.run() == 0
    for i in (0, 10, 20):
        want = oracle.rad_traverse(g, X, Q[i], n_to_score)
        s, a, o = t.results(i)
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)


def test_layout_quality_on_hierarchical_corpus(gpu, oracle):
    idx, X, g = _index(oracle, 30000, 1024, 8, 64, 2)
    info = idx.optimize_layout()
    # neighbours of a row span clearly fewer groups of 384 ids than the row has entries
    assert info.groups_per_row < 0.6 * info.degree, (info.groups_per_row, info.degree)


@pytest.mark.parametrize("layout", ["identity", "reversed", "random", "one-position", "own-chunk"])
def test_results_do_not_depend_on_the_layout(gpu, oracle, monkeypatch, layout):
    monkeypatch.setenv("RADHIP_TRAV", "4")
    monkeypatch.setenv("RADHIP_TABLE", "group")
    n, n_to_score = 9000, 2000
    idx, X, g = _index(oracle, n, 1024, 8, 48, 2, seed=9)
    i = np.arange(n, dtype=np.uint64)
    lid = {"identity": i, "reversed": n - 1 - i, "random": np.random.default_rng(4).permutation(n).astype(np.uint64),
           "one-position": i * 384,      # every node in chunk position 0 of its own group: all probes collide on one position
           "own-chunk": i * 48}[layout]  # every node alone in a chunk: the table degenerates to one entry per node
    idx.set_layout(lid.astype(np.uint32))
    Q = X[np.random.default_rng(5).integers(0, n, 9)]
    _check(oracle, idx, X, g, Q, n_to_score, "grouped")


def test_chunk_position_overflow_falls_back_to_the_hash_table(gpu, oracle, monkeypatch):
    """n_to_score chunks on ONE chunk position exceed the lines of the table: the kernel reports it and the
    library re-runs the batch on the per-slot table — same results, no error."""
    from rad_amd.device import DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "4")
    monkeypatch.setenv("RADHIP_TABLE", "group")
    n, n_to_score = 9000, 6000
    idx, X, g = _index(oracle, n, 1024, 8, 48, 2, seed=9)
    idx.set_layout((np.arange(n, dtype=np.uint64) * 384).astype(np.uint32))
    Q = X[[5, 4000]]
    t = DeviceTraversal(idx, Q, n_to_score)
    assert t.table == "grouped"
    assert t.run() == 0
    assert t.table == "hash"
    for i in range(2):
        want = oracle.rad_traverse(g, X, Q[i], n_to_score)
        s, a, o = t.results(i)
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)


def test_traversal_object_does_not_survive_an_add(gpu, oracle):
    from rad_amd._lib import RadHipError
    from rad_amd.device import DeviceTraversal
    idx, X, g = _index(oracle, 3000, 1024, 8, 32, 2)
    t = DeviceTraversal(idx, X[:4], 500)
    idx.add_rows(oracle.synth_rows(3000, 500, 3500, 1024, 3, 2), seed=11, max_batch=64)
    with pytest.raises(RadHipError):
        t.run()
    with pytest.raises(RadHipError):
        t.reset(X[:4])


def test_replacing_the_layout_invalidates_grouped_traversals(gpu, oracle, monkeypatch):
    """A traversal bound to the grouped table holds the layout's device arrays: installing another layout frees
    them, so the object must refuse to run (RADHIP_E_STATE) instead of reading freed memory."""
    from rad_amd import _lib
    from rad_amd._lib import RadHipError
    from rad_amd.device import DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "4")
    idx, X, g = _index(oracle, 8000, 1024, 8, 48, 2)
    idx.optimize_layout()
    monkeypatch.setenv("RADHIP_TABLE", "group")
    Q = X[:5].copy()
    t = DeviceTraversal(idx, Q, 500)
    assert t.table == "grouped" and t.run() == 0
    idx.set_layout(np.random.default_rng(5).permutation(8000).astype(np.uint32))
    for call in (lambda: t.reset(Q), lambda: t.run()):
        with pytest.raises(RadHipError) as ei:
            call()
        assert ei.value.code == _lib.E_STATE
    t2 = DeviceTraversal(idx, Q, 500)            # a new object sees the new layout
    assert t2.table == "grouped" and t2.run() == 0
    for i in range(5):
        assert np.array_equal(t2.results(i)[0], oracle.rad_traverse(g, X, Q[i], 500).slots)
