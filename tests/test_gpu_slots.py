"""Round 4: heavy traversal state per RESIDENT ROW of trav4_kernel instead of per traversal (RADHIP_TRAV_SLOTS), and launches
that do not wait (radhip_traversal_start / _finish on the object's own stream).  Results must be those of the oracle —
rad/coordination_service.py:290-413 restated — whichever row worked on whichever traversal, however often a row's tables
were reused (epochs) or cleared by the row itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mk(ndim, M, cap0, n, seed=5, mode=2, ef=64, batch=512):
    from rad_amd.device import DeviceIndex
    idx = DeviceIndex(ndim, M, cap0, ef)
    idx.synth_vectors(n, seed=seed, mode=mode)
    idx.link_resident(seed=9, max_batch=batch)
    return idx


def _oracle_graph(oracle, idx, n, M, cap0):
    X = idx.read_vectors(0, n)
    levels, adj0, upper_row, adjU = idx.read_graph()
    inf = idx.info()
    return X, oracle.Graph(n, cap0, M, int(inf.max_level), int(inf.entry), levels, adj0, upper_row, adjU)


def _check(t, want, nq, logs=True):
    st = t.stats()
    for i in range(nq):
        s, a, o = t.results(i)
        assert np.array_equal(s, want[i].slots) and np.array_equal(a, want[i].and_cnt) and np.array_equal(o, want[i].or_cnt), i
        assert st.n_pops[i] == want[i].n_pops and st.n_nbr[i] == want[i].n_nbr and st.n_scored[i] == len(want[i].slots), i
        if logs:
            nodes, lv = t.pop_log(i)
            assert np.array_equal(nodes, want[i].pop_nodes) and np.array_equal(lv, want[i].pop_levels), i


@pytest.mark.parametrize("slots,grid,epoch_max", [(4, 1, None), (8, 2, None), (8, 2, 2), (12, 3, 1), (4, 1, 3)])
@pytest.mark.parametrize("table", ["bucket", "group", "local"])
def test_slots_fewer_than_traversals(gpu, oracle, monkeypatch, slots, grid, epoch_max, table):
    """37 traversals of different lengths on 4 / 8 / 12 rows' worth of state: every row takes several traversals one after the
    other and reuses its tables under a new epoch; with one, two or three epochs the rows clear their own tables in between
    (what happens at 1B rows, where a bucket entry has ONE epoch bit left).  Scored lists, counts and pop logs == oracle."""
    from rad_amd.device import DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "4")
    if table != "bucket":
        monkeypatch.setenv("RADHIP_TABLE", table)
    monkeypatch.setenv("RADHIP_TEST_SLOTS", str(slots))
    monkeypatch.setenv("RADHIP_TEST_GRID", str(grid))
    if epoch_max is not None:
        monkeypatch.setenv("RADHIP_TEST_EPOCH_MAX", str(epoch_max))
    n, nq, nts = 6000, 37, 900
    idx = _mk(1024, 8, 16, n)
    X, g = _oracle_graph(oracle, idx, n, 8, 16)
    rng = np.random.default_rng(3)
    Q = X[rng.integers(0, n, nq)].copy()
    Q[3] = 0
    want = [oracle.rad_traverse(g, X, Q[i], nts) for i in range(nq)]
    t = DeviceTraversal(idx, Q, nts, log_pops=True, slots=True)
    assert t.kernel == "trav4_kernel" and t.slots == slots and t.table == {"group": "grouped"}.get(table, table)
    assert t.run() == 0
    _check(t, want, nq)
    # a second and a third batch on the same object: the rows' epochs carry on from where the last batch left them
    for rep in range(2):
        Q2 = X[rng.integers(0, n, nq)].copy()
        want2 = [oracle.rad_traverse(g, X, Q2[i], nts) for i in range(nq)]
        t.reset(Q2)
        assert t.run() == 0
        _check(t, want2, nq)
    t.close()
    idx.close()


def test_slot_batches_refuse_what_needs_state_per_traversal(gpu, monkeypatch):
    """a batch with per-row state runs to completion: no max_pops, no parked targets (their state would have to outlive
    the row's next traversal)"""
    from rad_amd.device import DeviceTraversal
    from rad_amd._lib import RadHipError
    monkeypatch.setenv("RADHIP_TRAV", "4")
    monkeypatch.setenv("RADHIP_TEST_SLOTS", "4")
    n = 3000
    idx = _mk(1024, 8, 16, n)
    Q = idx.read_vectors(0, 16)
    t = DeviceTraversal(idx, Q, 300, slots=True)
    assert t.slots == 4
    with pytest.raises(RadHipError):
        t.run(max_pops=10)
    with pytest.raises(RadHipError):
        t.set_targets(np.full(16, 100, np.uint64))
    assert t.run() == 0
    # without the flag, or when the batch fits the rows anyway, the state stays per traversal
    t2 = DeviceTraversal(idx, Q, 300)
    assert t2.slots == 0
    t2.close()
    t.close()
    idx.close()


def test_wide_rows_with_slots(gpu, oracle, monkeypatch):
    """the WIDE form (adjacency rows of 32 slots) with per-row state"""
    from rad_amd.device import DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "4")
    monkeypatch.setenv("RADHIP_TEST_SLOTS", "8")
    monkeypatch.setenv("RADHIP_TEST_GRID", "2")
    monkeypatch.setenv("RADHIP_TEST_EPOCH_MAX", "2")
    n, nq, nts = 5000, 29, 700
    idx = _mk(1024, 16, 32, n, ef=100)
    X, g = _oracle_graph(oracle, idx, n, 16, 32)
    rng = np.random.default_rng(11)
    Q = X[rng.integers(0, n, nq)].copy()
    want = [oracle.rad_traverse(g, X, Q[i], nts) for i in range(nq)]
    t = DeviceTraversal(idx, Q, nts, log_pops=True, slots=True)
    assert t.kernel == "trav4_kernel" and t.slots == 8
    assert t.run() == 0
    _check(t, want, nq)
    t.close()
    idx.close()


def test_two_objects_overlap_and_agree(gpu, oracle, monkeypatch):
    """two traversal objects with per-row state on two streams, batches started back to back (the second batch's wavefronts
    start while the first batch's last traversals still run): every batch equals the oracle and equals the same batch run
    alone; the busy interval of the pair is shorter than the sum of the launches."""
    from rad_amd.device import DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "4")
    monkeypatch.setenv("RADHIP_TEST_SLOTS", "16")
    n, nq, nts = 8000, 64, 1500
    idx = _mk(1024, 8, 16, n)
    X, g = _oracle_graph(oracle, idx, n, 8, 16)
    rng = np.random.default_rng(17)
    batches = [X[rng.integers(0, n, nq)].copy() for _ in range(4)]
    want = [[oracle.rad_traverse(g, X, b[i], nts) for i in range(nq)] for b in batches]
    A = DeviceTraversal(idx, batches[0], nts, slots=True, own_stream=True)
    B = DeviceTraversal(idx, batches[1], nts, slots=True, own_stream=True)
    assert A.slots == 16 and B.slots == 16
    A.start()
    B.start()
    assert A.finish() == 0
    _check(A, want[0], nq, logs=False)
    A.reset(batches[2])
    A.start()
    assert B.finish() == 0
    _check(B, want[1], nq, logs=False)
    B.reset(batches[3])
    B.start()
    assert A.finish() == 0
    _check(A, want[2], nq, logs=False)
    assert B.finish() == 0
    _check(B, want[3], nq, logs=False)
    hB = B.result_hashes()
    assert A.elapsed_to(B) > 0.0
    # the same batch alone, state per traversal
    C = DeviceTraversal(idx, batches[3], nts)
    assert C.slots == 0 and C.run() == 0
    assert np.array_equal(C.result_hashes(), hB)
    # start twice / finish without start are refused
    from rad_amd._lib import RadHipError
    A.reset(batches[0])
    A.start()
    with pytest.raises(RadHipError):
        A.start()
    with pytest.raises(RadHipError):
        A.reset(batches[1])
    assert A.finish() == 0
    with pytest.raises(RadHipError):
        A.finish()
    for t in (A, B, C):
        t.close()
    idx.close()


def test_slots_at_a_real_resident_round(gpu, oracle):
    """no test hook: 20000 traversals on the device's own rows (16384 on an MI355X) — the batch is larger than a resident
    round, the flag applies by itself; a sample of traversals from the head and the tail of the batch against the oracle"""
    from rad_amd.device import DeviceTraversal
    n, nq, nts = 200_000, 20_000, 2000
    idx = _mk(1024, 8, 16, n, batch=4096)
    X, g = _oracle_graph(oracle, idx, n, 8, 16)
    rng = np.random.default_rng(23)
    Q = X[rng.integers(0, n, nq)].copy()
    t = DeviceTraversal(idx, Q, nts, slots=True)
    assert t.kernel == "trav4_kernel"
    cap = idx.traversal_capacity()
    assert t.slots == (cap if nq > cap else 0)
    assert t.run() == 0
    t2 = DeviceTraversal(idx, Q, nts)
    assert t2.slots == 0 and t2.run() == 0
    assert np.array_equal(t.result_hashes(), t2.result_hashes())
    assert t.state_bytes() < t2.state_bytes()
    st = t.stats()
    for i in list(range(0, 6)) + list(range(nq - 6, nq)):
        w = oracle.rad_traverse(g, X, Q[i], nts)
        s, a, o = t.results(i)
        assert np.array_equal(s, w.slots) and np.array_equal(a, w.and_cnt) and np.array_equal(o, w.or_cnt), i
        assert st.n_pops[i] == w.n_pops
    t.close()
    t2.close()
    idx.close()


@pytest.mark.parametrize("slots,grid,ring", [(4, 1, 8), (8, 2, 16), (8, 1, 16), (16, 4, 32)])
@pytest.mark.parametrize("table", ["bucket", "local"])
def test_chained_batches_on_a_ring_of_scored_lists(gpu, oracle, monkeypatch, slots, grid, ring, table):
    """Several batches in ONE launch (radhip_traversal_create_ring): 75 traversals on 4 / 8 / 16 rows' worth of state and a ring of
    8 / 16 / 32 scored lists — every list slot is reused several times, and with so few rows a row regularly draws a traversal
    whose slot is still being written by a traversal on ANOTHER ROW OF ITS OWN WAVEFRONT (the gate must not block the wavefront).
    Counts of ALL traversals and the complete lists and pop logs of the last `ring` == oracle; the lists of the earlier ones are
    refused, not returned stale."""
    from rad_amd.device import DeviceTraversal
    from rad_amd._lib import RadHipError
    monkeypatch.setenv("RADHIP_TRAV", "4")
    if table != "bucket":
        monkeypatch.setenv("RADHIP_TABLE", table)
    monkeypatch.setenv("RADHIP_TEST_SLOTS", str(slots))
    monkeypatch.setenv("RADHIP_TEST_GRID", str(grid))
    n, nq, nts = 6000, 75, 700
    idx = _mk(1024, 8, 16, n)
    X, g = _oracle_graph(oracle, idx, n, 8, 16)
    rng = np.random.default_rng(21)
    Q = X[rng.integers(0, n, nq)].copy()
    want = [oracle.rad_traverse(g, X, Q[i], nts) for i in range(nq)]
    t = DeviceTraversal(idx, Q, nts, log_pops=True, list_ring=ring)
    assert t.kernel == "trav4_kernel" and t.slots == slots and t.list_ring == ring
    assert t.run() == 0
    st = t.stats()
    assert set(st.status.tolist()) <= {1, 2}
    for i in range(nq):
        assert st.n_pops[i] == want[i].n_pops and st.n_nbr[i] == want[i].n_nbr and st.n_scored[i] == len(want[i].slots), i
        nodes, lv = t.pop_log(i)                                            # (pop logs are kept per traversal)
        assert np.array_equal(nodes, want[i].pop_nodes) and np.array_equal(lv, want[i].pop_levels), i
    for i in range(nq - ring, nq):
        s, a, o = t.results(i)
        assert np.array_equal(s, want[i].slots) and np.array_equal(a, want[i].and_cnt) and np.array_equal(o, want[i].or_cnt), i
    h = t.result_hashes(nq - ring, ring)
    assert [int(x) for x in h] == [oracle.result_hash(w.slots, w.and_cnt, w.or_cnt) for w in want[nq - ring:]]
    with pytest.raises(RadHipError):
        t.results(nq - ring - 1)
    with pytest.raises(RadHipError):
        t.result_hashes(0, nq)
    # a shorter batch on the same object (reset with fewer queries), then a full one again
    Q2 = X[rng.integers(0, n, 30)].copy()
    want2 = [oracle.rad_traverse(g, X, Q2[i], nts) for i in range(30)]
    t.reset(Q2)
    assert t.run() == 0
    st2 = t.stats()
    assert len(st2.n_pops) == 30 and [int(x) for x in st2.n_pops] == [w.n_pops for w in want2]
    for i in range(max(0, 30 - ring), 30):
        s, a, o = t.results(i)
        assert np.array_equal(s, want2[i].slots), i
    t.reset(Q)
    assert t.run() == 0
    assert [int(x) for x in t.stats().n_pops] == [w.n_pops for w in want]
    t.close()
    idx.close()
