"""BASELINE config[3] (1B x 1024-bit fingerprints sharded across GPUs) as far as ONE GPU can go:
  * the whole 1B-row corpus (128 GB) and the closed-form graph over 1B nodes (74 GB) resident in one MI355X:
    size-independent properties of the single-GPU kernel, and the product loop of the row-sharded mode
    (radhip_shard_run, world 1, real RCCL communicator) on the same index — full results equal;
  * two virtual ranks at 200M rows, each CREATED with only its half of the rows (radhip_index_synth_vectors_shard)
    and the closed-form graph over all 200M nodes, stepped in lock step on one GPU: results equal the single-GPU
    kernel's on a full 200M-row index.
No rank-side object here ever holds rows it does not own, except the explicit single-GPU references."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _trav(idx, Q, nts):
    from rad_amd.device import DeviceTraversal
    t = DeviceTraversal(idx, Q, nts, log_pops=True)
    assert t.run() == 0
    out = [t.results(i) + t.pop_log(i) for i in range(Q.shape[0])]
    st = t.stats()
    t.close()
    return out, st


def test_one_billion_rows_single_gpu_and_sharded_loop(gpu, monkeypatch):
    from rad_amd.device import DeviceIndex, DeviceShard, RcclComm
    monkeypatch.setenv("RADHIP_TRAV", "4")
    monkeypatch.delenv("RADHIP_SHARD_ENGINE", raising=False)
    n, nts = 1_000_000_000, 50_000
    idx = DeviceIndex(1024, 8, 16, 64)
    idx.synth_vectors(n, seed=20260101, mode=1)
    idx.synth_graph(seed=777)
    inf = idx.info()
    assert inf.n == n and inf.device_bytes > 190e9
    rng = np.random.default_rng(31)
    Q = np.concatenate([idx.read_vectors(int(r), 1) for r in rng.integers(0, n, 6)])
    res, st = _trav(idx, Q, nts)
    assert (st.status == 1).all() and (st.n_scored >= nts).all() and (st.n_scored < nts + 16).all()
    for i, (s, a, o, nodes, lv) in enumerate(res):
        assert np.unique(s).size == s.size and int(s.max()) < n          # scored once, valid slots
        ga, go = idx.gather(Q[i:i + 1], s, np.array([0, s.size], np.uint64))
        assert np.array_equal(ga, a) and np.array_equal(go, o)           # scores == an independent gather-Tanimoto
    assert max(int(r[0].max()) for r in res) > 500_000_000                # rows far behind the 4 GiB mark were read (64-bit row addressing)
    short, _ = _trav(idx, Q[:2], 10_000)
    for i in range(2):
        k = short[i][0].size
        for x, y in zip(short[i][:3], res[i][:3]):
            assert np.array_equal(x, y[:k])                              # stopping earlier yields a prefix
    # the product loop of the row-sharded mode on the same index: one rank that owns every row
    comm = RcclComm(0, 1, RcclComm.unique_id(), 0)
    sh = DeviceShard(idx, 0, 1, 0, n, Q, nts, log_pops=True)
    steps = sh.run(comm)
    assert steps > 100
    sst = sh.stats()
    for i in range(Q.shape[0]):
        got = sh.results(i) + sh.pop_log(i)
        for x, y in zip(got, res[i]):
            assert np.array_equal(x, y), i
        assert sst.n_pops[i] == st.n_pops[i] and sst.n_nbr[i] == st.n_nbr[i] and sst.n_scored[i] == st.n_scored[i]
    sh.close()
    idx.close()


def test_two_virtual_ranks_200m_created_with_their_halves(gpu, monkeypatch):
    from rad_amd.device import DeviceIndex, DeviceShard
    monkeypatch.setenv("RADHIP_TRAV", "4")
    monkeypatch.delenv("RADHIP_SHARD_ENGINE", raising=False)
    n, nts, nq, world = 200_000_000, 12_000, 3, 2
    full = DeviceIndex(1024, 8, 16, 64)
    full.synth_vectors(n, seed=20260101, mode=1)
    full.synth_graph(seed=777)
    rng = np.random.default_rng(8)
    Qall = np.concatenate([full.read_vectors(int(r), 1) for r in rng.integers(0, n, world * nq)])
    want, wst = _trav(full, Qall, nts)
    full_bytes = full.info().device_bytes
    full.close()
    shards, idxs = [], []
    for r in range(world):
        first, count = r * (n // world), n // world
        idx = DeviceIndex(1024, 8, 16, 64)
        idx.synth_vectors_shard(count, first, n, seed=20260101, mode=1)
        idx.synth_graph(seed=777)
        inf = idx.info()
        assert inf.sharded == 1 and inf.shard_rows == count and inf.n == n
        assert inf.device_bytes < full_bytes - (n - count) * 128 + (1 << 20)      # its rows + the adjacency, nothing else
        idxs.append(idx)
        shards.append(DeviceShard(idx, r, world, first, count, Qall, nts, log_pops=True))
    scores = [np.zeros((s.nq, s.width), np.uint32) for s in shards]
    steps = 0
    while True:
        stepped = [s.step(scores[r]) for r, s in enumerate(shards)]
        steps += 1
        if sum(live for _req, live in stepped) == 0:
            break
        req_all = np.stack([req for req, _live in stepped])
        outs = [s.evaluate(req_all) for s in shards]
        scores = [sum(outs[k][r] for k in range(world)) for r in range(world)]
    assert steps > 100
    for r in range(world):
        st = shards[r].stats()
        for q in range(nq):
            t = r * nq + q
            got = shards[r].results(q) + shards[r].pop_log(q)
            for x, y in zip(got, want[t]):
                assert np.array_equal(x, y), (r, q)
            assert st.n_pops[q] == wst.n_pops[t] and st.n_nbr[q] == wst.n_nbr[t]
