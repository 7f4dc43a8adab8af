"""Multi-GPU paths on CPU, world_size 2 over torch.distributed (gloo), the oracle standing in for the
device kernels:
  * the row-sharded traversal (ONE global graph, rows and traversals partitioned, per-step exchange of
    frontier candidates and their scores): the result of every query must EQUAL the single-process oracle
    traversal of the same global corpus and graph — scored order, counts and pop log;
  * the plain-TCP process group the bench uses instead of torch (rad_amd/rendezvous.py);
  * the federated rounds (labelled alternative) and their allocation function."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_allocate_targets_properties():
    from rad_amd.sharded import KEY_EMPTY, allocate_targets
    sc = np.array([[10, 5, 0], [10, 7, 0], [10, 0, 0]], np.uint64)          # [world=3, nq=3]
    fr = np.array([[50, 9, KEY_EMPTY], [40, 9, KEY_EMPTY], [60, KEY_EMPTY, KEY_EMPTY]], np.uint64)
    t, done = allocate_targets(sc, fr, 100, local_cap=1000)
    assert done.tolist() == [False, False, True]
    # query 0: remaining 70 over 3 live shards: 23 each + 1 extra to the best frontier (rank 1)
    assert t[:, 0].tolist() == [33, 34, 33]
    # query 1: shard 2 is empty; remaining 88 over 2: 44 each; equal keys -> no extra needed
    assert t[:, 1].tolist() == [49, 51, 0]
    # query 2: every queue empty -> done, targets unchanged
    assert t[:, 2].tolist() == [0, 0, 0]
    # budget met -> done
    t2, d2 = allocate_targets(np.array([[60], [40]], np.uint64), np.array([[1], [2]], np.uint64), 100, 1000)
    assert d2.tolist() == [True] and t2[:, 0].tolist() == [60, 40]
    # a shard at its local capacity is skipped
    t3, d3 = allocate_targets(np.array([[50], [10]], np.uint64), np.array([[1], [2]], np.uint64), 100, 50)
    assert t3[:, 0].tolist() == [50, 50] and not d3[0]
    # remainder goes to the best frontiers first, ties by rank
    t4, _ = allocate_targets(np.zeros((4, 1), np.uint64), np.array([[7], [3], [3], [9]], np.uint64), 10, 100)
    assert t4[:, 0].tolist() == [2, 3, 3, 2]


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle import rad_oracle as O
    from rad_amd.sharded import ShardedTraversal
    from sharded_util import OracleLocalTraversal, make_shards
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)

    def allgather(a):
        t = torch.from_numpy(np.ascontiguousarray(a, np.uint64).view(np.int64))
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return np.stack([o.numpy().view(np.uint64) for o in outs])
    n_per, ndim, M, cap0, nts = 6000, 1024, 8, 16, 900
    X, g = make_shards(O, world, n_per, ndim, M, cap0, 5)[rank]
    Q = O.synth_rows(0, 5, world * n_per, ndim, 5, 1)            # same queries on every rank
    local = OracleLocalTraversal(O, g, X, Q, local_cap=nts)
    st = ShardedTraversal(local, allgather, rank, world, nts, local_cap=nts)
    sc, fr = st.run()
    np.savez(out_path, scored=sc, frontier=fr, rounds=st.rounds,
             slots=np.concatenate([local.results(i)[0] for i in range(5)]),
             counts=np.array([local.results(i)[0].shape[0] for i in range(5)]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_traversal_world2_gloo(tmp_path, oracle):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    ctx = mp.get_context("spawn")
    procs = []
    for r in range(world):
        p = ctx.Process(target=_worker, args=(r, world, port, str(tmp_path / f"r{r}.npz")))
        p.start()
        procs.append(p)
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    r0, r1 = np.load(tmp_path / "r0.npz"), np.load(tmp_path / "r1.npz")
    # every rank saw the same all-gathered state and took the same number of rounds
    assert np.array_equal(r0["scored"], r1["scored"]) and np.array_equal(r0["frontier"], r1["frontier"])
    assert int(r0["rounds"]) == int(r1["rounds"]) >= 1
    sc = r0["scored"].astype(np.int64)
    nts = 900
    # the global budget is met, without more overshoot than one expansion row per shard
    assert (sc.sum(0) >= nts).all() and (sc.sum(0) <= nts + 2 * 16).all()
    # what each rank holds matches what it reported
    assert np.array_equal(r0["counts"], sc[0]) and np.array_equal(r1["counts"], sc[1])
    # the sequential single-process reference of the same rounds gives the same split
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from rad_amd.sharded import ShardedTraversal
    from sharded_util import OracleLocalTraversal, make_shards
    shards = make_shards(oracle, world, 6000, 1024, 8, 16, 5)
    Q = oracle.synth_rows(0, 5, world * 6000, 1024, 5, 1)
    locals_ = [OracleLocalTraversal(oracle, g, X, Q, nts) for X, g in shards]
    # lock-step emulation: drive both shards by hand with the same allocation function
    from rad_amd.sharded import allocate_targets
    first = -(-nts // world)
    for l in locals_:
        l.set_targets(np.full(5, first, np.uint64))
    rounds = 0
    while True:
        for l in locals_:
            l.run()
        fs = [l.frontier() for l in locals_]
        scm = np.stack([f[1] for f in fs])
        frm = np.stack([f[0] for f in fs])
        rounds += 1
        t, done = allocate_targets(scm, frm, nts, nts)
        if done.all():
            break
        for r, l in enumerate(locals_):
            l.set_targets(t[r])
    assert rounds == int(r0["rounds"]) and np.array_equal(scm.astype(np.int64), sc)


# ------------------------------------------------------------------ row-sharded traversal (north star)
_ROW_Q = [11, 4000, 8999, 17, 5555, 2, 7001, 333]


def _row_worker(rank, world, port, out_path, exchange, n=9000, nq=3):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import rad_oracle as O
    from rad_amd.sharded import RowShardedTraversal
    from sharded_util import OracleRowShard
    ndim, M, cap0, nts = 1024, 8, 16, 1200
    X = O.synth_rows(0, n, n, ndim, 5, 2)
    g = O.synth_graph(n, M, cap0, 21)
    Qall = X[_ROW_Q[:world * nq]].copy()            # world * nq queries, rank-major
    rows = n // world
    first = rank * rows
    count = rows if rank < world - 1 else n - first
    local = OracleRowShard(O, g, X[first:first + count], first, Qall, rank, world, nts)
    if exchange == "gloo":
        import torch
        import torch.distributed as dist
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)

        def allgather(a):
            t = torch.from_numpy(np.ascontiguousarray(a, np.uint32).view(np.int32))
            outs = [torch.empty_like(t) for _ in range(world)]
            dist.all_gather(outs, t)
            return np.stack([o.numpy().view(np.uint32) for o in outs])

        def reduce_scatter(a):       # gloo has no reduce_scatter: all-reduce, keep this rank's block
            t = torch.from_numpy(np.ascontiguousarray(a, np.uint32).view(np.int32).copy())
            dist.all_reduce(t)
            return t.numpy().view(np.uint32)[rank]
        finish = lambda: (dist.barrier(), dist.destroy_process_group())
    else:
        from rad_amd.rendezvous import TcpGroup
        grp = TcpGroup(rank, world, "127.0.0.1", port)
        assert grp.broadcast_obj(b"id" if rank == 0 else None) == b"id"
        assert grp.allreduce([rank + 1.0, 2.0], "max").tolist() == [float(world), 2.0]
        allgather, reduce_scatter = grp.allgather_u32, grp.reduce_scatter_sum_u32
        finish = lambda: (grp.barrier(), grp.close())
    drv = RowShardedTraversal(local, allgather, reduce_scatter, rank, world)
    steps = drv.run()
    res = {}
    for q in range(nq):
        s, a, o, pn, pl = local.results(q)
        res.update({f"s{q}": s, f"a{q}": a, f"o{q}": o, f"pn{q}": pn, f"pl{q}": pl})
    np.savez(out_path, steps=steps, bytes=drv.exchanged_bytes, **res)
    finish()


@pytest.mark.parametrize("exchange", ["gloo", "tcp"])
def test_row_sharded_traversal_equals_single_index_world2(tmp_path, oracle, exchange):
    import multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, nq = 2, 3
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_row_worker, args=(r, world, port, str(tmp_path / f"row{r}.npz"), exchange)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    n, nts = 9000, 1200
    X = oracle.synth_rows(0, n, n, 1024, 5, 2)
    g = oracle.synth_graph(n, 8, 16, 21)
    Qall = X[[11, 4000, 8999, 17, 5555, 2]]
    outs = [np.load(tmp_path / f"row{r}.npz") for r in range(world)]
    assert int(outs[0]["steps"]) == int(outs[1]["steps"]) > 10          # every rank stopped at the same step
    for r in range(world):
        for q in range(nq):
            want = oracle.rad_traverse(g, X, Qall[r * nq + q], nts)      # single index, whole corpus
            z = outs[r]
            assert np.array_equal(z[f"s{q}"], want.slots), (r, q)
            assert np.array_equal(z[f"a{q}"], want.and_cnt) and np.array_equal(z[f"o{q}"], want.or_cnt)
            assert np.array_equal(z[f"pn{q}"], want.pop_nodes) and np.array_equal(z[f"pl{q}"], want.pop_levels)


def test_stepper_equals_sequential_traversal(oracle):
    """the oracle's stepper (what the sharded step kernel restates), driven with a local evaluator, IS the
    sequential traversal: same scored order, counts, pop log — including a drained queue"""
    n = 4000
    X = oracle.synth_rows(0, n, n, 1024, 3, 2)
    g = oracle.synth_graph(n, 8, 16, 5)
    for qi, nts in ((7, 600), (100, n), (3999, 1)):
        want = oracle.rad_traverse(g, X, X[qi], nts)
        st = oracle.Stepper(g, nts)
        a = o = None
        while True:
            req = st.step(a, o)
            if req.size == 0:
                break
            a, o = oracle.gather(X, X[qi], req)
        r = st.result()
        assert np.array_equal(r.slots, want.slots) and np.array_equal(r.and_cnt, want.and_cnt) and np.array_equal(r.or_cnt, want.or_cnt)
        assert np.array_equal(r.pop_nodes, want.pop_nodes) and np.array_equal(r.pop_levels, want.pop_levels)
        assert st.status == (1 if len(want.slots) >= nts else 2)


def test_row_sharded_traversal_world4_gloo_uneven_shards(tmp_path, oracle):
    """four ranks over gloo, a corpus that does not divide by four (the last rank holds three rows more),
    two traversals per rank: every rank stops at the same step and returns the single-index traversal's lists."""
    import multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world, nq, n, nts = 4, 2, 9003, 1200
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_row_worker, args=(r, world, port, str(tmp_path / f"row{r}.npz"), "gloo", n, nq)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    X = oracle.synth_rows(0, n, n, 1024, 5, 2)
    g = oracle.synth_graph(n, 8, 16, 21)
    Qall = X[_ROW_Q[:world * nq]]
    outs = [np.load(tmp_path / f"row{r}.npz") for r in range(world)]
    assert len({int(o["steps"]) for o in outs}) == 1 and int(outs[0]["steps"]) > 10
    for r in range(world):
        for q in range(nq):
            want = oracle.rad_traverse(g, X, Qall[r * nq + q], nts)
            z = outs[r]
            assert np.array_equal(z[f"s{q}"], want.slots), (r, q)
            assert np.array_equal(z[f"a{q}"], want.and_cnt) and np.array_equal(z[f"o{q}"], want.or_cnt)
            assert np.array_equal(z[f"pn{q}"], want.pop_nodes) and np.array_equal(z[f"pl{q}"], want.pop_levels)
