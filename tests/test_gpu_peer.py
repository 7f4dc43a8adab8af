"""Round 4: the peer-mapped corpus — BASELINE's row partitioning with no lock step.  Every rank maps
the row shards of all ranks into one virtual address range and runs the UNCHANGED single-GPU traversal kernel over it; on
the 8-GPU box the peers' shards are read over xGMI, here the "peers" are separate physical allocations on one GPU — in one
process (virtual ranks) and in two processes that hand their dmabuf descriptors over a Unix socket."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _graph_for(oracle, X, n):
    from rad_amd.device import DeviceIndex
    full = DeviceIndex(1024, 8, 16, 48)
    full.load_vectors(X)
    full.link_resident(seed=3, max_batch=512)
    return full


@pytest.mark.parametrize("world,n", [(4, 60_000), (3, 50_001)])
def test_virtual_ranks_map_each_others_shards(gpu, oracle, world, n):
    """`world` indices on one GPU, each created with ITS rows only; after the descriptor exchange every one of them holds the
    whole corpus (own shard + the others' allocations mapped in place) and the single-GPU traversal kernel returns, on each,
    exactly what it returns on an ordinary index — scored lists, counts, pop logs == oracle.  50 001 rows over 3 ranks leaves
    the last rank's shard short (and the 2-MiB granule makes the shards 32768 rows: rank 2 holds none of the corpus)."""
    from rad_amd._lib import RadHipError
    from rad_amd.device import DeviceIndex, DeviceTraversal
    X = oracle.synth_rows(0, n, n, 1024, 77, 2)
    full = _graph_for(oracle, X, n)
    levels, adj0, upper_row, adjU = full.read_graph()
    inf = full.info()
    g = oracle.Graph(n, 16, 8, int(inf.max_level), int(inf.entry), levels, adj0, upper_row, adjU)
    ranks = [DeviceIndex(1024, 8, 16, 48) for _ in range(world)]
    rps = [ix.peer_create(r, world, n) for r, ix in enumerate(ranks)]
    assert len(set(rps)) == 1 and rps[0] % 16384 == 0 and rps[0] * world >= n
    rps = rps[0]
    for r, ix in enumerate(ranks):
        first = r * rps
        ix.peer_fill_rows(X[first:min(first + rps, n)] if first < n else X[:0])
        i = ix.info()
        assert i.sharded == 1 and i.shard_first == first and i.shard_rows == max(0, min(rps, n - first))
        with pytest.raises(RadHipError):
            ix.peer_seal()                                   # the peers' shards are not mapped yet
    fds = [ix.peer_export() for ix in ranks]
    for r, ix in enumerate(ranks):
        for p in range(world):
            if p != r:
                ix.peer_import(p, fds[p])
    for fd in fds:
        os.close(fd)
    rng = np.random.default_rng(5)
    nq, nts = 6, 1500
    Q = X[rng.integers(0, n, nq)].copy()
    want = [oracle.rad_traverse(g, X, Q[i], nts) for i in range(nq)]
    ref = DeviceTraversal(full, Q, nts)
    assert ref.run() == 0
    for r, ix in enumerate(ranks):
        ix.peer_seal()
        i = ix.info()
        assert i.sharded == 0 and i.n == n
        assert np.array_equal(ix.read_vectors(0, n), X)                       # every slot reads the right row, whoever owns it
        if r == 0:
            ix.copy_graph_from(full)                                          # device to device
        else:
            ix.load_graph(levels, adj0, upper_row, adjU, int(inf.max_level), int(inf.entry))
        with pytest.raises(RadHipError):
            ix.load_vectors(X[:10])                                           # a peer-mapped corpus is read-only
        with pytest.raises(RadHipError):
            ix.keep_rows(0, 10)
        t = DeviceTraversal(ix, Q, nts, log_pops=True)
        assert t.run() == 0
        st = t.stats()
        for q in range(nq):
            s, a, o = t.results(q)
            nodes, lv = t.pop_log(q)
            assert np.array_equal(s, want[q].slots) and np.array_equal(a, want[q].and_cnt) and np.array_equal(o, want[q].or_cnt), (r, q)
            assert np.array_equal(nodes, want[q].pop_nodes) and np.array_equal(lv, want[q].pop_levels)
            assert st.n_pops[q] == want[q].n_pops
        assert np.array_equal(t.result_hashes(), ref.result_hashes())
        # the scan and the exact top-k read the mapped range like any corpus
        a_, o_ = ix.scan(Q[:2], 0, n)
        a0, o0 = full.scan(Q[:2], 0, n)
        assert np.array_equal(a_, a0) and np.array_equal(o_, o0)
        t.close()
    ref.close()
    for ix in ranks:
        ix.close()
    full.close()


def _peer_proc(rank, world, n, key, out_path):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from oracle import rad_oracle as O
    from rad_amd.device import DeviceIndex, DeviceTraversal
    from rad_amd.rendezvous import exchange_fds
    ix = DeviceIndex(1024, 8, 16, 48)
    rps = ix.peer_create(rank, world, n)
    ix.peer_fill_synth(seed=77, mode=2)                      # this rank's rows of the closed-form corpus, on the device
    fd = ix.peer_export()
    fds = exchange_fds(rank, world, fd, key)
    for p in range(world):
        if p != rank:
            ix.peer_import(p, fds[p])
            os.close(fds[p])
    os.close(fd)
    ix.peer_seal()
    X = O.synth_rows(0, n, n, 1024, 77, 2)
    ok_rows = bool(np.array_equal(ix.read_vectors(0, n), X))
    ix.link_resident(seed=3, max_batch=512)                  # the graph, built over rows half of which live in the peer's allocation
    Q = X[[5, n // 2, n - 3, 1234]].copy()
    t = DeviceTraversal(ix, Q, 1200)
    assert t.run() == 0
    np.savez(out_path, ok_rows=ok_rows, rps=rps, hashes=t.result_hashes(), adj0=ix.read_graph()[1])
    t.close()
    ix.close()


def test_two_processes_exchange_descriptors_over_a_unix_socket(gpu, oracle, tmp_path):
    """the product path of the descriptor exchange: two rank PROCESSES on one GPU, each creates its shard, exports a dmabuf
    descriptor, the descriptors cross as SCM_RIGHTS messages (rad_amd.rendezvous.exchange_fds), each process imports the
    other's shard — and builds the graph and traverses over rows half of which live in the other process's allocation.
    Both processes return what an ordinary index returns."""
    from rad_amd.device import DeviceIndex, DeviceTraversal
    world, n = 2, 40_000
    key = f"test-{os.getpid()}"
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_peer_proc, args=(r, world, n, key, str(tmp_path / f"p{r}.npz"))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    X = oracle.synth_rows(0, n, n, 1024, 77, 2)
    full = _graph_for(oracle, X, n)
    Q = X[[5, n // 2, n - 3, 1234]].copy()
    t = DeviceTraversal(full, Q, 1200)
    assert t.run() == 0
    want = t.result_hashes()
    adj0 = full.read_graph()[1]
    for r in range(world):
        z = np.load(tmp_path / f"p{r}.npz")
        assert bool(z["ok_rows"]) and int(z["rps"]) == 32768
        assert np.array_equal(z["adj0"], adj0)
        assert np.array_equal(z["hashes"], want)
    t.close()
    full.close()
