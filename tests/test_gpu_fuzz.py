"""Randomised parity of the traversal kernels against the oracle on graphs that no builder would
make: random level assignment, ragged and empty adjacency rows, links to far-away nodes, duplicate
and all-zero fingerprints (long runs of equal scores: the queue order is then decided by the
bytewise order of "{id}:{level}" alone), odd dimensions, every row width the kernels dispatch
on.  Seeds are fixed: a failure reproduces."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
NO_SLOT = 0xFFFFFFFF
# RAD_FUZZ_CASES=N widens the derived-case campaigns (default sizes keep the suite at a few seconds)
N_TRAV_CASES = int(os.environ.get("RAD_FUZZ_CASES", 40))
N_BUILD_CASES = int(os.environ.get("RAD_FUZZ_CASES", 24))


def _random_graph(oracle, rng, n, M, cap0, max_level, p_empty):
    levels = np.zeros(n, np.int8)
    # geometric-ish levels; at least one node on the top level
    for lv in range(1, max_level + 1):
        k = max(1, int(n / (3 ** lv)))
        levels[rng.choice(n, k, replace=False)] = np.maximum(levels[rng.choice(n, k, replace=False)], lv)
    levels[rng.integers(0, n)] = max_level
    upper_row = np.full(n, NO_SLOT, np.uint32)
    rows = 0
    for i in range(n):
        if levels[i] > 0:
            upper_row[i] = rows
            rows += int(levels[i])
    adjU = np.full((max(rows, 1), M), NO_SLOT, np.uint32)
    adj0 = np.full((n, cap0), NO_SLOT, np.uint32)
    members = [np.flatnonzero(levels >= lv) for lv in range(max_level + 1)]
    for i in range(n):
        for lv in range(int(levels[i]) + 1):
            cap = cap0 if lv == 0 else M
            pool = members[lv][members[lv] != i]
            if rng.random() < p_empty or pool.size == 0:
                continue
            k = int(rng.integers(1, cap + 1))
            k = min(k, pool.size)
            tg = rng.choice(pool, k, replace=False).astype(np.uint32)
            if lv == 0:
                adj0[i, :k] = tg
            else:
                adjU[upper_row[i] + lv - 1, :k] = tg
    top = np.flatnonzero(levels == max_level)
    return oracle.Graph(n, cap0, M, max_level, int(top[0]), levels, adj0, upper_row, adjU)


def _random_rows(rng, n, ndim, dup_frac):
    rb = (ndim + 7) // 8
    dens = rng.choice([0.03, 0.1, 0.5])
    bits = rng.random((n, rb * 8)) < dens
    bits[:, ndim:] = False
    X = np.packbits(bits, axis=1)
    ndup = int(n * dup_frac)
    if ndup:   # copies of a few rows: equal scores against any query
        src = rng.integers(0, n, ndup)
        dst = rng.integers(0, n, ndup)
        X[dst] = X[src]
    X[rng.integers(0, n, max(1, n // 50))] = 0   # all-zero rows
    return np.ascontiguousarray(X)


CASES = [
    # (seed, n, ndim, M, cap0, max_level, p_empty, dup_frac, kernel)
    (1, 300, 64, 4, 8, 2, 0.1, 0.5, "trav4"),
    (2, 2000, 1024, 8, 16, 3, 0.05, 0.1, "trav4"),
    (3, 2000, 1024, 8, 16, 3, 0.05, 0.1, "trav1"),
    (4, 1500, 2048, 32, 64, 2, 0.0, 0.2, "trav1"),
    (5, 1000, 200, 16, 32, 2, 0.2, 0.3, "trav1"),
    (6, 500, 32, 2, 4, 4, 0.3, 0.8, "trav4"),
    (7, 4000, 512, 8, 8, 3, 0.02, 0.0, "trav4"),
    (8, 800, 1000, 5, 10, 1, 0.1, 0.4, "trav4"),
    (9, 800, 1536, 12, 24, 2, 0.1, 0.4, "trav1"),
    (10, 3000, 1024, 16, 16, 0, 0.05, 0.2, "trav4"),    # single level: every node is primed
    (11, 100, 1024, 8, 16, 5, 0.5, 0.9, "trav4"),
    (12, 2500, 128, 3, 6, 3, 0.0, 0.6, "trav1"),
    # rows wider than 16 slots on the four-per-wavefront kernel (its WIDE form: a pop walks its row in chunks of 16)
    (13, 1500, 2048, 32, 64, 2, 0.0, 0.2, "trav4"),
    (14, 1000, 200, 16, 32, 2, 0.2, 0.3, "trav4"),
    (15, 800, 1536, 12, 24, 2, 0.1, 0.4, "trav4"),
    (16, 2000, 1024, 16, 32, 3, 0.05, 0.1, "trav4"),
    (17, 1200, 1024, 24, 48, 1, 0.0, 0.0, "trav4"),
]


@pytest.mark.parametrize("seed,n,ndim,M,cap0,max_level,p_empty,dup_frac,kernel", CASES)
def test_random_graphs_match_oracle(gpu, oracle, monkeypatch, seed, n, ndim, M, cap0, max_level, p_empty, dup_frac, kernel):
    from rad_amd.device import DeviceIndex, DeviceTraversal
    monkeypatch.setenv("RADHIP_TRAV", "1" if kernel == "trav1" else "4")
    rng = np.random.default_rng(seed)
    g = _random_graph(oracle, rng, n, M, cap0, max_level, p_empty)
    X = _random_rows(rng, n, ndim, dup_frac)
    idx = DeviceIndex(ndim, M, cap0, 32)
    idx.load_vectors(X)
    idx.load_graph(g.levels, g.adj0, g.upper_row, g.adjU, g.max_level, g.entry)
    nq = 9
    Q = X[rng.integers(0, n, nq)].copy()
    Q[0] = 0                                    # all-zero query: every distance is 1 (or 0 against zero rows)
    Q[1] = _random_rows(rng, 1, ndim, 0.0)[0]   # a query that is not in the corpus
    for nts in (max(1, n // 7), n):             # stop early / drain the queue
        t = DeviceTraversal(idx, Q, nts, log_pops=True)
        assert t.kernel == ("trav_kernel" if kernel == "trav1" else "trav4_kernel")
        assert t.run() == 0
        st = t.stats()
        for i in range(nq):
            want = oracle.rad_traverse(g, X, Q[i], nts)
            s, a, o = t.results(i)
            nodes, levels = t.pop_log(i)
            assert np.array_equal(nodes, want.pop_nodes) and np.array_equal(levels, want.pop_levels), (seed, nts, i)
            assert np.array_equal(s, want.slots) and np.array_equal(a, want.and_cnt) and np.array_equal(o, want.or_cnt), (seed, nts, i)
            assert st.n_nbr[i] == want.n_nbr and st.n_pops[i] == want.n_pops
        t.close()
    idx.close()


def _derived_case(seed):
    r = np.random.default_rng(1000 + seed)
    M = int(r.choice([2, 3, 4, 5, 6, 8, 12, 16, 24, 32]))
    cap0 = int(r.choice([M, 2 * M]))
    ndim = int(r.choice([8, 24, 64, 100, 256, 512, 1024, 1100, 2048]))
    n = int(r.integers(2, 1500))
    max_level = int(r.integers(0, 6))
    kernel = "trav4" if r.random() < 0.7 else "trav1"   # (drawn last: the cases before it stay what they were)
    return (2000 + seed, n, ndim, M, cap0, max_level, float(r.choice([0.0, 0.1, 0.4])), float(r.choice([0.0, 0.3, 0.9])), kernel)


@pytest.mark.parametrize("case", [_derived_case(s) for s in range(N_TRAV_CASES)], ids=lambda c: "-".join(str(x) for x in c))
def test_more_random_graphs(gpu, oracle, monkeypatch, case):
    test_random_graphs_match_oracle(gpu, oracle, monkeypatch, *case)


def _builder_case(seed):
    r = np.random.default_rng(5000 + seed)
    M = int(r.choice([2, 3, 4, 8, 12, 16, 32]))
    cap0 = int(r.choice([M, 2 * M]))
    return dict(seed=seed, n=int(r.integers(2, 1200)), ndim=int(r.choice([16, 64, 200, 1024, 2048])), M=M, cap0=cap0,
                ef=int(r.choice([1, 4, 20, 64, 200])), max_batch=int(r.choice([1, 3, 64, 1000])),
                dup=float(r.choice([0.0, 0.5, 0.95])))


@pytest.mark.parametrize("c", [_builder_case(s) for s in range(N_BUILD_CASES)], ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_random_builds_and_searches_match_oracle(gpu, oracle, c):
    """Index.add (GPU insert kernels) and Index.search against the oracle's usearch-shaped builder
    on rows with many exact duplicates (equal distances everywhere: candidate order is decided by
    the slot tie-break), tiny expansion_add, batches larger than the index."""
    from rad_amd.index import Index
    rng = np.random.default_rng(c["seed"])
    n, ndim, M, cap0 = c["n"], c["ndim"], c["M"], c["cap0"]
    X = _random_rows(rng, n, ndim, c["dup"])
    h = oracle.Hnsw(ndim, M, cap0, c["ef"], seed=c["seed"])
    h.add(X, max_batch=c["max_batch"])
    g = h.graph()
    idx = Index(ndim=ndim, connectivity=M, connectivity_base=cap0, expansion_add=c["ef"], seed=c["seed"],
                max_batch=c["max_batch"])
    idx.add(np.arange(n), X)
    levels, adj0, upper_row, adjU = idx.device_index().read_graph()
    assert idx.max_level == g.max_level and idx.device_index().info().entry == g.entry
    assert np.array_equal(levels, g.levels) and np.array_equal(upper_row, g.upper_row)
    bad = np.nonzero((adj0 != g.adj0).any(1))[0]
    assert bad.size == 0, f"level-0 rows differ at nodes {bad[:10]}"
    assert np.array_equal(adjU[:g.adjU.shape[0]], g.adjU)
    Q = np.concatenate([X[rng.integers(0, n, 4)], _random_rows(rng, 2, ndim, 0.0)])
    k = int(min(n, rng.choice([1, 5, 30])))
    ef = int(rng.choice([1, 8, 100]))
    m = idx.search(Q, count=k, expansion=ef)
    for i in range(Q.shape[0]):
        s, a, o, _, _ = oracle.graph_search(g, X, Q[i], k, ef)
        cnt = int(m.counts[i])
        assert cnt == s.size and np.array_equal(m.slots[i, :cnt], s), (i, cnt, s.size)
