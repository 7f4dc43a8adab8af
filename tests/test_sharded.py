"""Multi-GPU path on CPU: the federated-rounds logic of rad_amd/sharded.py with world_size 2
over torch.distributed (gloo), the oracle standing in for the device traversal; plus the pure
allocation function."""
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
