"""GPU parity (through the C ABI) of the usearch-shaped insert / search kernels against the
CPU oracle, of the device RAD traversal against golden vectors captured from the reference's
own control flow, and the end-to-end drop-in path Index -> LocalHNSWService -> RADTraverser."""
import sqlite3

import numpy as np
import pytest

from golden_util import f32_distance, golden, load_graph_npz

pytestmark = pytest.mark.gpu
NO_SLOT = 0xFFFFFFFF


def _rows(oracle, n, ndim, seed):
    return oracle.synth_rows(0, n, n, ndim, seed, 1 if ndim >= 512 else 0)


@pytest.mark.parametrize("ndim,n,M,cap0,ef,max_batch", [
    (1024, 1500, 8, 16, 32, 1),        # classical sequential insert
    (1024, 4000, 8, 16, 64, 256),      # batched
    (64, 1200, 4, 8, 20, 64),          # the reference tests' shape (tests/test_hnsw_service.py:13-28)
    (2048, 1500, 16, 32, 100, 128),
    (1024, 2500, 8, 16, 400, 512),     # README expansion_add (the top buffer in registers, eight entries per lane)
    (1024, 1500, 8, 16, 128, 64),      # ... two per lane
    (1024, 1800, 8, 16, 600, 256),     # expansion_add > 512: the top buffer in LDS (search_layer), two buffers
    (2048, 1200, 32, 64, 100, 64),     # rows of 64 slots: the neighbour selection that reloads the selected rows
])
def test_index_add_matches_oracle_builder(gpu, oracle, ndim, n, M, cap0, ef, max_batch):
    from rad_amd.index import Index
    X = _rows(oracle, n, ndim, 31)
    h = oracle.Hnsw(ndim, M, cap0, ef, seed=5)
    h.add(X, max_batch=max_batch)
    g = h.graph()
    idx = Index(ndim=ndim, dtype="b1", metric="tanimoto", connectivity=M, connectivity_base=cap0,
                expansion_add=ef, seed=5, max_batch=max_batch)
    idx.add(np.arange(n), X)
    levels, adj0, upper_row, adjU = idx.device_index().read_graph()
    assert idx.max_level == g.max_level and idx.device_index().info().entry == g.entry
    assert np.array_equal(levels, g.levels)
    assert np.array_equal(upper_row, g.upper_row)
    bad = np.nonzero((adj0 != g.adj0).any(1))[0]
    assert bad.size == 0, f"level-0 rows differ at nodes {bad[:10]}"
    assert np.array_equal(adjU, g.adjU)
    assert np.array_equal(idx.device_index().get_top_level_nodes(), g.top_level())


def test_index_add_in_two_calls_matches_oracle(gpu, oracle):
    """Batch boundaries are part of the algorithm: two add() calls restart the batch schedule
    at the call boundary, exactly as the oracle does."""
    from rad_amd.index import Index
    n, ndim = 3000, 1024
    X = _rows(oracle, n, ndim, 8)
    h = oracle.Hnsw(ndim, 8, 16, 48, seed=2)
    h.add(X[:1000], max_batch=128)
    h.add(X[1000:], max_batch=128)
    g = h.graph()
    b = Index(ndim=ndim, connectivity=8, expansion_add=48, seed=2, max_batch=128)
    b.add(np.arange(1000), X[:1000])
    b.add(np.arange(1000, n), X[1000:])
    levels, adj0, upper_row, adjU = b.device_index().read_graph()
    assert np.array_equal(levels, g.levels) and np.array_equal(adj0, g.adj0)
    assert np.array_equal(upper_row, g.upper_row) and np.array_equal(adjU, g.adjU)
    assert np.array_equal(b.device_index().read_vectors(0, n), X)
    assert len(b) == n


@pytest.mark.parametrize("ndim,n,M,ef_add,k,ef", [(1024, 5000, 8, 64, 10, 64), (1024, 5000, 8, 64, 100, 400),
                                                  (64, 1500, 4, 20, 5, 16), (2048, 3000, 16, 64, 32, 32),
                                                  (1024, 4000, 8, 64, 50, 700)])     # expansion > 512: the LDS top buffer
def test_search_matches_oracle_search(gpu, oracle, ndim, n, M, ef_add, k, ef):
    from rad_amd.index import Index
    X = _rows(oracle, n, ndim, 77)
    h = oracle.Hnsw(ndim, M, 2 * M, ef_add, seed=1)
    h.add(X, max_batch=64)
    g = h.graph()
    idx = Index(ndim=ndim, connectivity=M, expansion_add=ef_add, seed=1)
    idx.load_graph(np.arange(n) + 10, X, g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
    rng = np.random.default_rng(3)
    Q = np.concatenate([X[:6], rng.integers(0, 256, (4, X.shape[1]), dtype=np.uint8)])
    m = idx.search(Q, count=k, expansion=ef)
    tot_e = tot_p = 0
    for i in range(Q.shape[0]):
        s, a, o, ne, npop = oracle.graph_search(g, X, Q[i], k, ef)
        cnt = int(m.counts[i])
        assert cnt == s.size
        assert np.array_equal(m.slots[i, :cnt], s), f"query {i}"
        assert np.array_equal(m.keys[i, :cnt], s.astype(np.uint64) + 10)
        assert np.array_equal(m.distances[i, :cnt], f32_distance(a, o))
        tot_e += ne
        tot_p += npop
    assert (m.computed_distances, m.visited_members) == (tot_e, tot_p)
    # exact search by the scan kernel: brute-force order (distance, slot)
    ex = idx.search(Q[:3], count=k, exact=True)
    for i in range(3):
        a, o = oracle.scan(X, Q[i])
        qk = ((o.astype(np.int64) - a) << 23) // np.maximum(o.astype(np.int64), 1)
        want = np.lexsort((np.arange(n), qk))[:k]
        assert np.array_equal(ex.slots[i], want)
    # graph search finds most of the exact neighbours
    if ndim >= 512 and k == 10:
        rec = np.mean([len(set(m.slots[i, :k]) & set(ex.slots[i, :k])) / k for i in range(3)])
        assert rec >= 0.8


@pytest.mark.parametrize("tag,M", [("t64", 4), ("t1024", 8)])
def test_device_traversal_matches_reference_golden(gpu, trav_mode, tag, M):
    """trav_kernel vs the reference's own control flow (golden fixtures): expansion order,
    scored order and float32 scores, including heavy score ties on the 64-bit fixture."""
    from rad_amd.index import Index
    from rad_amd.traverser import TanimotoRADTraverser
    z = load_graph_npz(f"g1{tag}_graph.npz")
    ndim = z["fps"].shape[1] * 8
    idx = Index(ndim=ndim, connectivity=M, connectivity_base=z["adj0"].shape[1])
    idx.load_graph(None, z["fps"], z["levels"], z["adj0"], z["upper_row"], z["adjU"], int(z["max_level"]), int(z["entry"]))
    cases = golden()[f"g1{tag}"]
    for nts in sorted({c["n_to_score"] for c in cases}):
        t = TanimotoRADTraverser(idx, z["queries"], log_pops=True)
        t.traverse(n_to_score=nts)
        for c in [c for c in cases if c["n_to_score"] == nts]:
            q = c["query"]
            nodes, levels = t._trav.pop_log(q)
            assert nodes.tolist() == c["pop_nodes"], (tag, q, nts)
            assert levels.tolist() == c["pop_levels"]
            mols = t.get_molecules(q=q)
            assert [m[0] for m in mols] == c["slots"]
            assert [m[1] for m in mols] == c["scores"]
        t.shutdown()


def test_drop_in_end_to_end(gpu, oracle, trav_mode, tmp_path):
    """README.md:45-97 workflow with rad_amd substituted: Index.add on the GPU, SQLite SMILES
    join, RADTraverser with a user scoring_fn; then the same traversal with Tanimoto scoring
    entirely on the device gives the identical result."""
    from rad_amd.hnsw_service import create_local_hnsw_service
    from rad_amd.index import Index
    from rad_amd.traverser import RADTraverser, TanimotoRADTraverser, create_local_traverser
    n, ndim = 3000, 1024
    X = _rows(oracle, n, ndim, 12)
    keys = np.arange(n, dtype=np.uint64) + 500
    hnsw = Index(ndim=ndim, dtype="b1", metric="tanimoto", connectivity=8, expansion_add=64)
    hnsw.add(keys, X, log="Building HNSW")
    db = str(tmp_path / "molecules.db")
    con = sqlite3.connect(db)
    con.execute("CREATE TABLE nodes (node_key INTEGER PRIMARY KEY, smi TEXT NOT NULL)")
    con.executemany("INSERT INTO nodes VALUES (?, ?)", [(int(k), f"K{int(k)}") for k in keys])
    con.commit()
    con.close()
    query = X[42]
    qb = np.unpackbits(query).astype(np.int64)
    bits = np.unpackbits(X, axis=1).astype(np.int64)
    a = bits @ qb
    o = bits.sum(1) + qb.sum() - a

    def score_fn(smiles):
        i = int(smiles[1:]) - 500
        return float(np.float32(1.0) - np.float32(a[i]) / np.float32(o[i]))
    service = create_local_hnsw_service(hnsw, database_path=db)
    trav = RADTraverser(hnsw_service=service, scoring_fn=score_fn)
    trav.prime()
    trav.traverse(n_workers=1, n_to_score=800)
    host = trav.get_molecules()
    assert len(host) >= 800 and all(m[2] == f"K{m[0] + 500}" for m in host)
    dev = TanimotoRADTraverser(hnsw, query, smiles_of=lambda ks: [f"K{int(k)}" for k in ks])
    dev.traverse(n_to_score=800)
    assert dev.get_molecules() == host
    assert dev.get_best_molecules(5) == trav.get_best_molecules(5)
    # multi-worker mode on the same index
    t2 = create_local_traverser(hnsw, score_fn, database_path=db)
    t2.prime()
    t2.traverse(n_workers=3, n_to_score=300)
    ids = [m[0] for m in t2.get_molecules()]
    assert len(ids) >= 300 and len(set(ids)) == len(ids)
    # persistence round trip keeps the graph and the vectors
    p = str(tmp_path / "index.npz")
    hnsw.save(p)
    again = Index.restore(p)
    assert np.array_equal(again.get_neighbors(7, 0), hnsw.get_neighbors(7, 0))
    assert np.array_equal(again.search(query, 5).keys, hnsw.search(query, 5).keys)
    assert hnsw.get_node_ids_from_keys([500, 542]).tolist() == [0, 42]
