"""GPU side of the sharded traversal: per-traversal targets / park / resume and frontier keys
of trav_kernel against the oracle, two shards driven through the federated rounds on one GPU
(threads + a barrier standing in for the ranks), and the RCCL communicator with world = 1."""
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _device_shard(O, X, g, M, cap0):
    from rad_amd.device import DeviceIndex
    idx = DeviceIndex(X.shape[1] * 8, M, cap0, 64)
    idx.load_vectors(X)
    idx.load_graph(g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
    return idx


def test_targets_park_resume_and_frontier(gpu, oracle, trav_mode):
    from rad_amd import _lib
    from rad_amd.device import DeviceTraversal
    from sharded_util import make_shards
    (X, g), = make_shards(oracle, 1, 8000, 1024, 8, 16, 9)
    idx = _device_shard(oracle, X, g, 8, 16)
    Q = X[[1, 50, 4000]].copy()
    t = DeviceTraversal(idx, Q, 2000)
    key = _lib.lib().radhip_rad_key
    for tg in ([100, 300, 50], [100, 700, 900], [2000, 2000, 2000]):
        t.set_targets(np.array(tg, np.uint64))
        t.run()
        keys, scored = t.frontier()
        st = t.stats()
        for i in range(3):
            want = oracle.rad_traverse(g, X, Q[i], tg[i], log_pops=False)
            s, a, o = t.results(i)
            assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
            assert int(scored[i]) == want.slots.shape[0] and st.n_pops[i] == want.n_pops
            assert int(keys[i]) == (0xFFFFFFFFFFFFFFFF if want.frontier is None else key(*want.frontier))
    assert set(t.stats().status.tolist()) <= {1, 2}


def test_two_shards_federated_rounds_on_one_gpu(gpu, oracle, trav_mode):
    from rad_amd.device import DeviceTraversal
    from rad_amd.sharded import ShardedTraversal, allocate_targets
    from sharded_util import OracleLocalTraversal, make_shards
    world, n_per, nts = 2, 6000, 900
    shards = make_shards(oracle, world, n_per, 1024, 8, 16, 5)
    Q = oracle.synth_rows(0, 5, world * n_per, 1024, 5, 1)
    barrier = threading.Barrier(world)
    slots_box = [None] * world
    out = [None] * world
    errs = []

    def allgather_for(rank):
        def allgather(a):
            slots_box[rank] = np.asarray(a, np.uint64).copy()
            barrier.wait(60)
            res = np.stack(slots_box)
            barrier.wait(60)
            return res
        return allgather

    def run(rank):
        try:
            X, g = shards[rank]
            idx = _device_shard(oracle, X, g, 8, 16)
            local = DeviceTraversal(idx, Q, nts)
            st = ShardedTraversal(local, allgather_for(rank), rank, world, nts, local_cap=nts)
            sc, fr = st.run()
            out[rank] = (sc, fr, st.rounds, [local.results(i) for i in range(5)])
        except Exception as e:  # pragma: no cover
            errs.append(e)
            barrier.abort()
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not errs, errs
    # oracle lock-step reference of the same rounds
    locals_ = [OracleLocalTraversal(oracle, g, X, Q, nts) for X, g in shards]
    for l in locals_:
        l.set_targets(np.full(5, -(-nts // world), np.uint64))
    rounds = 0
    while True:
        [l.run() for l in locals_]
        fs = [l.frontier() for l in locals_]
        scm, frm = np.stack([f[1] for f in fs]), np.stack([f[0] for f in fs])
        rounds += 1
        tg, done = allocate_targets(scm, frm, nts, nts)
        if done.all():
            break
        for r, l in enumerate(locals_):
            l.set_targets(tg[r])
    for r in range(world):
        sc, fr, nr, res = out[r]
        assert nr == rounds and np.array_equal(sc, scm) and np.array_equal(fr, frm)
        for i in range(5):
            for x, y in zip(res[i], locals_[r].results(i)):
                assert np.array_equal(x, y)


def test_rccl_comm_world1(gpu):
    from rad_amd.device import RcclComm
    uid = RcclComm.unique_id()
    assert len(uid) == 128
    c = RcclComm(0, 1, uid, 0)
    a = np.arange(1000, dtype=np.uint64) * 3
    out = c.allgather_u64(a)
    assert out.shape == (1, 1000) and np.array_equal(out[0], a)
    c.close()


# ------------------------------------------------------------------ row-sharded traversal (north star)
def _lockstep(shards):
    """The per-step exchange of rad_amd.sharded.RowShardedTraversal for ranks that live in one process:
    all-gather = stack, reduce-scatter = sum over ranks of the block of each rank."""
    world = len(shards)
    scores = [np.zeros((s.nq, s.width), np.uint32) for s in shards]
    steps = 0
    while True:
        stepped = [s.step(scores[r]) for r, s in enumerate(shards)]
        steps += 1
        if sum(live for _req, live in stepped) == 0:
            return steps
        req_all = np.stack([req for req, _live in stepped])
        outs = [s.evaluate(req_all) for s in shards]
        scores = [sum(outs[k][r] for k in range(world)) for r in range(world)]


def _row_sharded_setup(oracle, n, nts, world, nq, mode=2, seed=7):
    X = oracle.synth_rows(0, n, n, 1024, seed, mode)
    idx_rows = []
    from rad_amd.device import DeviceIndex
    full = DeviceIndex(1024, 8, 16, 48)
    full.add_rows(X, seed=3, max_batch=256)                 # ONE graph over all rows, built on the GPU
    levels, adj0, upper_row, adjU = full.read_graph()
    inf = full.info()
    g = oracle.Graph(n, 16, 8, int(inf.max_level), int(inf.entry), levels, adj0, upper_row, adjU)
    Qall = X[np.random.default_rng(seed).integers(0, n, world * nq)].copy()
    return X, g, full, Qall


@pytest.fixture(params=["row", "row-spec0", "row-spec1", "wave", "thread", "thread-spec1", "thread-spec2"])
def engine(request, monkeypatch):
    """the three step kernels of shard.hip: sixteen lanes per traversal (the default), the single-GPU traversal kernel
    cut at the fingerprint read, and the thread-per-traversal restatement of the oracle's stepper — row and thread
    without speculation (the thread engine's default) and with one or two queue heads expanded speculatively per step
    (two is the row engine's default): the committed state must not depend on it"""
    name, _, spec = request.param.partition("-spec")
    monkeypatch.setenv("RADHIP_SHARD_ENGINE", name)
    if spec:
        monkeypatch.setenv("RADHIP_SHARD_SPEC", spec)
    else:
        monkeypatch.delenv("RADHIP_SHARD_SPEC", raising=False)
    return name


@pytest.mark.parametrize("world", [2, 3])
def test_row_sharded_equals_single_gpu_traversal(gpu, oracle, world, engine):
    """`world` ranks' worth of step / evaluation kernels on one GPU, each rank keeping ONLY its rows
    (radhip_index_keep_rows): every query's scored order, counts and pop log equal the oracle's traversal of
    the whole corpus, and the single-GPU kernel's."""
    from rad_amd.device import DeviceIndex, DeviceShard, DeviceTraversal
    n, nts, nq = 20000, 1500, 5
    X, g, full, Qall = _row_sharded_setup(oracle, n, nts, world, nq)
    single = DeviceTraversal(full, Qall, nts, log_pops=True)
    assert single.run() == 0
    levels, adj0, upper_row, adjU = g.levels, g.adj0, g.upper_row, g.adjU
    shards, idxs = [], []
    rows = n // world
    for r in range(world):
        first = r * rows
        count = rows if r < world - 1 else n - first
        idx = DeviceIndex(1024, 8, 16, 48)
        idx.load_vectors(X)
        idx.load_graph(levels, adj0, upper_row, adjU, g.max_level, g.entry)
        idx.keep_rows(first, count)                          # this rank holds its rows only
        assert idx.info().has_vectors
        idxs.append(idx)
        shards.append(DeviceShard(idx, r, world, first, count, Qall, nts, log_pops=True))
        assert shards[-1].engine == engine
    steps = _lockstep(shards)
    assert steps > 10
    for r in range(world):
        st = shards[r].stats()
        assert set(st.status.tolist()) <= {1, 2}
        for q in range(nq):
            t = r * nq + q
            want = oracle.rad_traverse(g, X, Qall[t], nts)
            s, a, o = shards[r].results(q)
            nodes, lv = shards[r].pop_log(q)
            assert np.array_equal(s, want.slots), (r, q)
            assert np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
            assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(lv, want.pop_levels)
            s1, a1, o1 = single.results(t)
            assert np.array_equal(s, s1) and np.array_equal(a, a1) and np.array_equal(o, o1)
            assert st.n_pops[q] == want.n_pops and st.n_nbr[q] == want.n_nbr


def test_row_sharded_native_loop_world1_rccl(gpu, oracle, engine):
    """radhip_shard_run — the product loop (kernels + RCCL collectives on one stream, device buffers) — with
    a real RCCL communicator of world 1, to a drained queue (n_to_score = n)."""
    from rad_amd.device import DeviceShard, RcclComm
    n, nq = 3000, 4
    X, g, full, Qall = _row_sharded_setup(oracle, n, n, 1, nq, mode=1, seed=4)
    comm = RcclComm(0, 1, RcclComm.unique_id(), 0)
    sh = DeviceShard(full, 0, 1, 0, n, Qall, n, log_pops=True)
    assert sh.engine == engine
    steps = sh.run(comm)
    assert steps > 10
    for q in range(nq):
        want = oracle.rad_traverse(g, X, Qall[q], n)
        s, a, o = sh.results(q)
        assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
    step_ms, eval_ms, nsteps, xbytes = sh.timing()
    assert nsteps == steps and xbytes > 0 and step_ms > 0


def test_sharded_index_refuses_whole_corpus_entry_points(gpu, oracle):
    from rad_amd._lib import RadHipError
    from rad_amd.device import DeviceTraversal
    X, g, full, Qall = _row_sharded_setup(oracle, 2000, 100, 2, 1)
    full.keep_rows(1000, 1000)
    for call in (lambda: full.scan(Qall[:1]), lambda: DeviceTraversal(full, Qall[:1], 10), lambda: full.read_vectors(0, 1),
                 lambda: full.keep_rows(0, 10)):
        with pytest.raises(RadHipError):
            call()


# ------------------------------------------------------------------ shard-native setup
def _check_against_oracle(oracle, shards, g, X, Qall, nq, nts):
    for r, sh in enumerate(shards):
        st = sh.stats()
        assert set(st.status.tolist()) <= {1, 2}
        for q in range(nq):
            want = oracle.rad_traverse(g, X, Qall[r * nq + q], nts)
            s_, a, o = sh.results(q)
            nodes, lv = sh.pop_log(q)
            assert np.array_equal(s_, want.slots), (r, q)
            assert np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
            assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(lv, want.pop_levels)
            assert st.n_pops[q] == want.n_pops and st.n_nbr[q] == want.n_nbr


@pytest.mark.parametrize("world", [2, 3])
def test_ranks_created_with_only_their_rows(gpu, oracle, world, monkeypatch):
    """Every rank's index is CREATED with its rows only (radhip_index_load_vectors_shard): no rank ever allocates,
    stages or uploads another rank's rows; the graph over the whole corpus comes from load_graph.  Results equal
    the oracle's traversal of the whole corpus."""
    from rad_amd._lib import RadHipError
    from rad_amd.device import DeviceIndex, DeviceShard
    monkeypatch.delenv("RADHIP_SHARD_ENGINE", raising=False)
    n, nts, nq = 20000, 1500, 4
    X, g, full, Qall = _row_sharded_setup(oracle, n, nts, world, nq)
    full_bytes = full.info().device_bytes
    full.close()
    rows = n // world
    shards, idxs = [], []
    for r in range(world):
        first = r * rows
        count = rows if r < world - 1 else n - first
        idx = DeviceIndex(1024, 8, 16, 48)
        idx.load_vectors_shard(X[first:first + count], first, n)
        idx.load_graph(g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
        sh = DeviceShard(idx, r, world, first, count, Qall, nts, log_pops=True)
        inf = idx.info()
        assert inf.sharded == 1 and inf.shard_first == first and inf.shard_rows == count and inf.n == n
        # rows of this rank + the whole adjacency, nothing else: well below a full index
        assert inf.device_bytes <= count * 128 + (full_bytes - n * 128) + 4096
        with pytest.raises(RadHipError):
            idx.read_vectors(0, 1)                       # whole-corpus entry points refuse a shard
        with pytest.raises(RadHipError):
            DeviceShard(idx, r, world, 0, n, Qall, nts)  # rows it does not hold
        idxs.append(idx); shards.append(sh)
    assert _lockstep(shards) > 10
    _check_against_oracle(oracle, shards, g, X, Qall, nq, nts)


def test_synthetic_shards_and_closed_form_graph(gpu, oracle, monkeypatch):
    """config[3]'s setup in small: every rank generates ITS rows of the closed-form corpus on the device
    (radhip_index_synth_vectors_shard) and the closed-form graph over all n_total nodes (no row is read for it) —
    no collective, no host copy, nothing of the other ranks' rows."""
    from rad_amd.device import DeviceIndex, DeviceShard
    monkeypatch.delenv("RADHIP_SHARD_ENGINE", raising=False)
    n, nts, nq, world = 30000, 2000, 3, 3
    X = oracle.synth_rows(0, n, n, 1024, 1234, 1)
    g = oracle.synth_graph(n, 8, 16, 99)
    Qall = X[np.random.default_rng(5).integers(0, n, world * nq)].copy()
    rows = n // world
    shards, idxs = [], []
    for r in range(world):
        first = r * rows
        count = rows if r < world - 1 else n - first
        idx = DeviceIndex(1024, 8, 16, 64)
        idx.synth_vectors_shard(count, first, n, seed=1234, mode=1)
        idx.synth_graph(seed=99)
        inf = idx.info()
        assert inf.n == n and inf.shard_rows == count and inf.max_level == g.max_level
        lv, a0, ur, aU = idx.read_graph()
        assert np.array_equal(a0, g.adj0) and np.array_equal(lv, g.levels) and np.array_equal(aU, g.adjU)
        idxs.append(idx)
        shards.append(DeviceShard(idx, r, world, first, count, Qall, nts, log_pops=True))
    assert _lockstep(shards) > 10
    _check_against_oracle(oracle, shards, g, X, Qall, nq, nts)


def test_link_resident_equals_add(gpu, oracle):
    """radhip_index_link_resident (rows already in HBM) builds the graph radhip_index_add builds from host rows."""
    from rad_amd.device import DeviceIndex
    n = 40000
    X = oracle.synth_rows(0, n, n, 1024, 20260101, 2)
    a = DeviceIndex(1024, 8, 16, 48)
    a.add_rows(X, seed=777, max_batch=1024)
    b = DeviceIndex(1024, 8, 16, 48)
    b.synth_vectors(n, seed=20260101, mode=2)
    b.link_resident(seed=777, max_batch=1024)
    ga, gb = a.read_graph(), b.read_graph()
    for x, y in zip(ga, gb):
        assert np.array_equal(x, y)
    ia, ib = a.info(), b.info()
    assert (ia.max_level, ia.entry, ia.n) == (ib.max_level, ib.entry, ib.n)
    # extending: half by add, the rest resident
    c = DeviceIndex(1024, 8, 16, 48)
    c.load_vectors(X)
    c.link_resident(seed=777, max_batch=1024)
    for x, y in zip(ga, c.read_graph()):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("slots,spec", [(4, "0"), (8, "2"), (5, "1")])
def test_slots_take_the_traversals_of_a_batch(gpu, oracle, monkeypatch, slots, spec):
    """RADHIP_SHARD_SLOTS: fewer slots (queue + sets) than traversals; a slot whose traversal is done takes the next one,
    its sets keep the old entries as stale (epochs).  Results are by traversal number and equal the oracle's."""
    from rad_amd.device import DeviceShard, RcclComm
    monkeypatch.setenv("RADHIP_SHARD_ENGINE", "row")
    monkeypatch.setenv("RADHIP_SHARD_SPEC", spec)
    monkeypatch.setenv("RADHIP_SHARD_SLOTS", str(slots))
    n, nq, nts = 30000, 23, 2500
    X, g, full, Qall = _row_sharded_setup(oracle, n, nts, 1, nq, mode=2, seed=17)
    comm = RcclComm(0, 1, RcclComm.unique_id(), 0)
    sh = DeviceShard(full, 0, 1, 0, n, Qall, nts, log_pops=True)
    assert sh.engine == "row" and sh.slots == slots and sh.nq == nq
    steps = sh.run(comm)
    assert steps > 10
    st = sh.stats()
    assert set(st.status.tolist()) <= {1, 2}
    for q in range(nq):
        want = oracle.rad_traverse(g, X, Qall[q], nts)
        s_, a, o = sh.results(q)
        nodes, lv = sh.pop_log(q)
        assert np.array_equal(s_, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt), q
        assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(lv, want.pop_levels), q
        assert st.n_pops[q] == want.n_pops and st.n_nbr[q] == want.n_nbr
    # a second batch on the same state
    Q2 = X[np.random.default_rng(5).integers(0, n, nq)].copy()
    sh.reset(Q2)
    sh.run(comm)
    for q in (0, nq // 2, nq - 1):
        want = oracle.rad_traverse(g, X, Q2[q], nts)
        s_, a, o = sh.results(q)
        assert np.array_equal(s_, want.slots) and np.array_equal(a, want.and_cnt), q
    with pytest.raises(Exception):
        sh.step(np.zeros((sh.nq, sh.width), np.uint32))            # the host-staged pieces need one slot per traversal
    sh.close(); comm.close()


@pytest.mark.parametrize("eng", ["row", "thread"])
def test_speculation_cuts_steps_not_results(gpu, oracle, monkeypatch, eng):
    """Speculative score prefetch: fewer frontier steps, identical committed state."""
    from rad_amd.device import DeviceShard, RcclComm
    monkeypatch.setenv("RADHIP_SHARD_ENGINE", eng)
    n, nq, nts = 60000, 16, 6000
    X, g, full, Qall = _row_sharded_setup(oracle, n, nts, 1, nq, mode=2, seed=11)
    comm = RcclComm(0, 1, RcclComm.unique_id(), 0)
    steps, res = {}, {}
    for spec in (0, 1, 2):
        monkeypatch.setenv("RADHIP_SHARD_SPEC", str(spec))
        sh = DeviceShard(full, 0, 1, 0, n, Qall, nts, log_pops=True)
        assert sh.width == 16 * (1 + min(spec, 1))
        steps[spec] = sh.run(comm)
        res[spec] = [sh.results(q) + sh.pop_log(q) for q in range(nq)]
        depth, asked, used, hits = sh.speculation()
        assert depth == spec
        if spec:
            assert asked > 0 and 0 < used <= asked and hits > 0
        else:
            assert asked == used == hits == 0
        sh.close()
    for q in range(nq):
        want = oracle.rad_traverse(g, X, Qall[q], nts)
        for spec in (0, 1, 2):
            s_, a, o, nodes, lv = res[spec][q]
            assert np.array_equal(s_, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt)
            assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(lv, want.pop_levels)
    assert steps[1] < steps[0] * 0.8 and steps[2] <= steps[1], steps


def test_pair_of_groups_on_two_streams(gpu, oracle, monkeypatch):
    """radhip_shard_run_pair: two groups of traversals, each with its own stream and communicator, stepped in
    one loop; every group's results are those of its own run."""
    from rad_amd.device import DeviceShard, RcclComm
    monkeypatch.delenv("RADHIP_SHARD_ENGINE", raising=False)
    n, nq, nts = 30000, 6, 3000
    X, g, full, Qall = _row_sharded_setup(oracle, n, nts, 2, nq, mode=2, seed=21)
    ca = RcclComm(0, 1, RcclComm.unique_id(), 0)
    cb = RcclComm(0, 1, RcclComm.unique_id(), 0)
    a = DeviceShard(full, 0, 1, 0, n, Qall[:nq], nts, log_pops=True)
    b = DeviceShard(full, 0, 1, 0, n, Qall[nq:], nts // 2, log_pops=True, own_stream=True)
    steps = a.run_pair(ca, b, cb)
    assert steps > 10
    for grp, (sh, Q, t) in enumerate(((a, Qall[:nq], nts), (b, Qall[nq:], nts // 2))):
        for q in range(nq):
            want = oracle.rad_traverse(g, X, Q[q], t)
            s_, aa, oo = sh.results(q)
            nodes, lv = sh.pop_log(q)
            assert np.array_equal(s_, want.slots), (grp, q)
            assert np.array_equal(aa, want.and_cnt) and np.array_equal(oo, want.or_cnt)
            assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(lv, want.pop_levels)


@pytest.mark.parametrize("eng", ["row", "thread"])
def test_loop_fails_safe(gpu, oracle, monkeypatch, eng):
    """A host-side failure in the middle of radhip_shard_run (injected) and a device-side one (a queue that is
    too small): both return an error promptly, leak nothing that blocks the next run, and a fresh shard on the
    same index works afterwards."""
    from rad_amd import _lib
    from rad_amd._lib import RadHipError
    from rad_amd.device import DeviceShard, RcclComm
    monkeypatch.setenv("RADHIP_SHARD_ENGINE", eng)
    n, nq, nts = 20000, 8, 4000
    X, g, full, Qall = _row_sharded_setup(oracle, n, nts, 1, nq, mode=2, seed=31)
    comm = RcclComm(0, 1, RcclComm.unique_id(), 0)
    monkeypatch.setenv("RADHIP_SHARD_TEST_FAIL_AT", "7")
    sh = DeviceShard(full, 0, 1, 0, n, Qall, nts)
    with pytest.raises(RadHipError) as ei:
        sh.run(comm)
    assert ei.value.code == _lib.E_HIP and "injected" in str(ei.value) and "after 7 steps" in str(ei.value)
    sh.close()
    monkeypatch.delenv("RADHIP_SHARD_TEST_FAIL_AT")
    monkeypatch.setenv("RADHIP_SHARD_TEST_HEAP_CAP", "96")
    sh = DeviceShard(full, 0, 1, 0, n, Qall, nts)
    with pytest.raises(RadHipError) as ei:
        sh.run(comm)
    assert ei.value.code == _lib.E_CAPACITY
    sh.close()
    monkeypatch.delenv("RADHIP_SHARD_TEST_HEAP_CAP")
    sh = DeviceShard(full, 0, 1, 0, n, Qall, nts)
    assert sh.run(comm) > 10
    want = oracle.rad_traverse(g, X, Qall[0], nts)
    assert np.array_equal(sh.results(0)[0], want.slots)
    info = comm.info()
    assert info["comm_count"] == 1 and info["comm_rank"] == 0 and info["rccl_version_code"] > 0 and ":" in info["pci_bus_id"]


def test_eight_virtual_ranks_uneven_last_shard_and_an_idle_rank(gpu, oracle, monkeypatch):
    """EIGHT ranks' worth of step / evaluation kernels in lock step on one GPU, each rank created with its
    rows only; the corpus size is not divisible by eight (the last shard is longer); traversals differ in length, so some rank
    has no live traversal left while others still work and keeps stepping with nothing to do.  Every scored list, count and pop log equals the oracle's traversal of the whole corpus."""
    from rad_amd.device import DeviceIndex, DeviceShard
    monkeypatch.delenv("RADHIP_SHARD_ENGINE", raising=False)
    world, n, nts, nq = 8, 20003, 1200, 3
    X, g, full, Qall = _row_sharded_setup(oracle, n, nts, world, nq, seed=13)
    full.close()
    rows = n // world
    assert n % world != 0
    shards, idxs = [], []
    for r in range(world):
        first = r * rows
        count = rows if r < world - 1 else n - first
        idx = DeviceIndex(1024, 8, 16, 48)
        idx.load_vectors_shard(X[first:first + count], first, n)
        idx.load_graph(g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
        idxs.append(idx)
        shards.append(DeviceShard(idx, r, world, first, count, Qall, nts, log_pops=True))
    assert shards[-1].index.info().shard_rows == n - 7 * rows > rows
    # lock step, recording who is live at every step
    scores = [np.zeros((s.nq, s.width), np.uint32) for s in shards]
    steps, live_hist = 0, []
    while True:
        stepped = [s.step(scores[r]) for r, s in enumerate(shards)]
        steps += 1
        live_hist.append([live for _req, live in stepped])
        if sum(live_hist[-1]) == 0:
            break
        req_all = np.stack([req for req, _live in stepped])
        outs = [s.evaluate(req_all) for s in shards]
        scores = [sum(outs[k][r] for k in range(world)) for r in range(world)]
    live_hist = np.array(live_hist)
    assert steps > 10
    # some rank has been idle for a stretch while others were still live (traversal lengths differ)
    idle_while_others_work = ((live_hist == 0) & (live_hist.sum(1, keepdims=True) > 0)).sum(0)
    assert idle_while_others_work.max() >= 1, idle_while_others_work
    _check_against_oracle(oracle, shards, g, X, Qall, nq, nts)
    for s in shards:
        s.close()
    for i in idxs:
        i.close()
