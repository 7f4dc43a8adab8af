"""Pins the CPU oracle (oracle/rad_oracle.c) against golden vectors captured from the
reference's own control flow (tests/golden/make_golden.py): Tanimoto-scored traversals must
reproduce the reference's expansion order, scored order and scores bit-exactly."""
import numpy as np
import pytest

from golden_util import f32_distance, golden, load_graph_npz, oracle_graph


@pytest.mark.parametrize("tag", ["t64", "t1024"])
def test_oracle_traversal_matches_reference_flow(oracle, tag):
    z = load_graph_npz(f"g1{tag}_graph.npz")
    g = oracle_graph(oracle, z)
    cases = golden()[f"g1{tag}"]
    assert cases
    for c in cases:
        q = z["queries"][c["query"]]
        r = oracle.rad_traverse(g, z["fps"], q, min(c["n_to_score"], g.n))
        assert r.pop_nodes.tolist() == c["pop_nodes"], (tag, c["query"], c["n_to_score"])
        assert r.pop_levels.tolist() == c["pop_levels"]
        assert r.slots.tolist() == c["slots"]
        got = f32_distance(r.and_cnt, r.or_cnt).astype(np.float64)
        assert got.tolist() == c["scores"]          # bit-exact float32 edge values


def test_oracle_tanimoto_against_numpy_bruteforce(oracle):
    rng = np.random.default_rng(0)
    for nbytes in (1, 8, 13, 128, 256):
        X = rng.integers(0, 256, (257, nbytes), dtype=np.uint8)
        X[3] = 0
        for qi in (0, 3):
            a, o = oracle.scan(X, X[qi])
            assert np.array_equal(a, np.unpackbits(X & X[qi], axis=1).sum(1))
            assert np.array_equal(o, np.unpackbits(X | X[qi], axis=1).sum(1))
    assert oracle.distance_f32(0, 0) == 0.0
    assert oracle.distance_f32(5, 5) == 0.0
    assert oracle.distance_f32(3, 7) == float(np.float32(1) - np.float32(3) / np.float32(7))


def test_oracle_graph_accessors_match_golden_graph(oracle):
    z = load_graph_npz("g1t64_graph.npz")
    g = oracle_graph(oracle, z)
    tops = np.nonzero(z["levels"] == int(z["max_level"]))[0]
    assert g.top_level().tolist() == tops.tolist()
    for slot in (0, 5, int(z["entry"])):
        for lv in range(int(z["levels"][slot]) + 1):
            row = z["adj0"][slot] if lv == 0 else z["adjU"][z["upper_row"][slot] + lv - 1]
            assert g.neighbors(slot, lv).tolist() == [int(x) for x in row if x != 0xFFFFFFFF]
    with pytest.raises(KeyError):
        g.neighbors(0, int(z["levels"][0]) + 1)
