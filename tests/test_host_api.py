"""Host-side API parity against golden vectors captured from the reference's own modules
(tests/golden/make_golden.py).  No GPU: state backends, coordinator, worker, traverser,
HNSW service, and the C-ABI library's symbol table."""
import os
import re
import sqlite3

import numpy as np
import pytest

from golden_util import golden, load_graph_npz

NO_SLOT = 0xFFFFFFFF
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    from rad_amd import _lib
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "rad_hip.h")).read()
    declared = set(re.findall(r"\b(radhip_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(L, name), f"librad_hip.so lacks {name}"
        assert name in _lib.SIGNATURES, f"rad_amd/_lib.py does not bind {name}"
    assert L.radhip_backend_name() == b"hip:gfx950"


def test_device_entry_points_fail_loudly_without_gpu():
    """No CPU fallback: on a box without a GPU every compute call raises."""
    from rad_amd import _lib
    from rad_amd.device import DeviceIndex
    if _lib.device_count() > 0:
        pytest.skip("GPU present")
    idx = DeviceIndex(64, 4)
    idx.load_vectors(np.zeros((10, 8), np.uint8))
    with pytest.raises(_lib.RadHipError) as e:
        idx.scan(np.zeros((1, 8), np.uint8))
    assert e.value.code == _lib.E_NO_DEVICE


# ----------------------------------------------------------------- state backends
def test_priority_queue_order_matches_redis_zset():
    from rad_amd.priority_queue import InProcessPQ
    g = golden()["g2"]
    pq = InProcessPQ()
    for nid, lv, sc in g["inserts"]:
        pq.insert(nid, lv, sc)
    out = []
    while True:
        it = pq.pop()
        if it is None:
            break
        out.append([it[0], it[1], it[2]])
    assert out == g["pops"]
    assert pq.pop() is None and len(pq) == 0


def test_visited_semantics():
    from rad_amd.visited import InProcessVisited
    g = golden()["g3"]
    vs = InProcessVisited()
    assert [vs.checkAndInsert(a, b) for a, b in g["calls"]] == g["returns"]


def test_scored_set_semantics(tmp_path):
    from rad_amd.scored import InProcessScoredSet
    g = golden()["scored"]
    ss = InProcessScoredSet()
    for a, b, c in g["inserts"]:
        ss.insert(a, b, c)
    assert len(ss) == g["length"]
    assert [list(x) for x in ss.get_molecules()] == g["molecules"]
    assert [list(x) for x in ss.get_molecules(2)] == g["first2"]
    assert [list(x) for x in ss.get_best_molecules()] == g["best"]
    assert [list(x) for x in ss.get_best_molecules(2)] == g["best2"]
    assert ss.getScore(7) == g["get7"] and ss.getScore(12345) is g["get_missing"]
    assert [list(x) for x in ss] == g["iter"]
    p = tmp_path / "s.txt"
    ss.save(str(p))
    assert p.read_text().splitlines()[0] == f"{g['iter'][0][0]} {g['iter'][0][1]}"


def test_work_item_dict_shape():
    from rad_amd.coordination_service import WorkItem
    g = golden()["g6"]
    d = WorkItem(5, 2, -1.5, request_id="rid", neighbors=[1, "C"]).to_dict()
    assert sorted(d.keys()) == g["keys"]
    assert (d["node_id"], d["level"], d["score"], d["neighbors"]) == (g["node_id"], g["level"], g["score"], g["neighbors"])
    assert sorted(WorkItem.from_dict(d).to_dict().keys()) == g["roundtrip"]


# ----------------------------------------------------------------- traverser vs reference flow
def _toy_service(z):
    from rad_amd.hnsw_service import HNSWService

    class Toy(HNSWService):
        def get_neighbors(self, node_id, level):
            if level > z["levels"][node_id]:
                raise KeyError((node_id, level))
            row = z["adj0"][node_id] if level == 0 else z["adjU"][z["upper_row"][node_id] + level - 1]
            out = []
            for nb in row:
                if nb != NO_SLOT:
                    out.extend([int(nb), f"S{int(nb)}"])
            return out

        def get_top_level_nodes(self):
            out = []
            for i in np.nonzero(z["levels"] == int(z["max_level"]))[0]:
                out.extend([int(i), f"S{int(i)}"])
            return out

        def is_healthy(self):
            return True

        def shutdown(self):
            pass

        def get_service_info(self):
            return {"service_type": "Toy"}

        def get_hnsw_info(self):
            return {"max_level": int(z["max_level"])}
    return Toy()


def _hash_score(smiles):
    h = 1469598103934665603
    for ch in smiles.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return -20.0 + (h % 100000) / 5000.0


class _PopLogPQ:
    """Wraps a PriorityQueue to record the pop sequence."""

    def __init__(self, inner):
        self.inner, self.log = inner, []

    def pop(self):
        it = self.inner.pop()
        if it is not None:
            self.log.append([it[0], it[1], it[2]])
        return it

    def insert(self, *a, **k):
        return self.inner.insert(*a, **k)

    def __len__(self):
        return len(self.inner)


def test_traverser_reproduces_reference_traversal_hash_scores():
    from rad_amd.priority_queue import InProcessPQ
    from rad_amd.traverser import RADTraverser
    g = golden()
    z = load_graph_npz("g1_graph.npz")
    pq = _PopLogPQ(InProcessPQ())
    t = RADTraverser(hnsw_service=_toy_service(z), scoring_fn=_hash_score, namespace="g1", priority_queue=pq)
    t.prime()
    t.traverse(n_workers=1, n_to_score=g["g1"]["n_to_score"])
    assert pq.log == g["g1"]["pops"]
    assert [list(m) for m in t.get_molecules()] == g["g1"]["molecules"]
    assert [list(m) for m in t.get_best_molecules(10)] == g["g1"]["best10"]
    assert len(t.scored_set) >= g["g1"]["n_to_score"]
    # drain the whole graph
    pq2 = _PopLogPQ(InProcessPQ())
    t2 = RADTraverser(hnsw_service=_toy_service(z), scoring_fn=_hash_score, priority_queue=pq2)
    t2.prime()
    t2.traverse(n_workers=1, n_to_score=10 ** 9)
    ga = g["g1_all"]
    assert len(pq2.log) == ga["n_pops"] and len(t2.scored_set) == ga["n_scored"]
    assert pq2.log[:50] == ga["pops_head"] and pq2.log[-50:] == ga["pops_tail"]
    assert [list(m) for m in t2.get_molecules()[-50:]] == ga["molecules_tail"]
    stats = t2.get_traversal_stats()
    assert stats["coordination"]["scored_molecules"] == ga["n_scored"]
    assert stats["coordination"]["pending_work"] == 0


def test_empty_adjacency_row_the_stated_deviation():
    """Golden G7: the first node the G1 traversal expands above level 0 has lost its row on that level.  The
    REFERENCE fails that work item (rad/distributed_worker.py:286-288: recorded ok = False) — nothing is
    submitted, the node does not descend from it, and the item stays assigned until the coordinator's 120 s
    cleanup.  This build completes the item and descends (deviation 1): the node appears one
    level down right away instead of whenever another path reaches it.  Everything before the empty row is
    identical, and on this graph both end with the same scored set."""
    from rad_amd.priority_queue import InProcessPQ
    from rad_amd.traverser import RADTraverser
    g = golden()["g7"]
    z = load_graph_npz("g7_graph.npz")
    v, lv = g["emptied_node"], g["emptied_level"]
    assert (z["adjU"][z["upper_row"][v] + lv - 1] == NO_SLOT).all()
    assert g["failed_items"] == [[v, lv, g["pops"][g["ok"].index(False)][2]]] and g["ok"].count(False) == 1
    pq = _PopLogPQ(InProcessPQ())
    t = RADTraverser(hnsw_service=_toy_service(z), scoring_fn=_hash_score, priority_queue=pq)
    t.prime()
    t.traverse(n_workers=1, n_to_score=10 ** 9)
    at = g["ok"].index(False)
    assert pq.log[:at + 1] == g["pops"][:at + 1]                      # identical up to and including the empty row
    assert pq.log[at + 1] == [v, lv - 1, g["pops"][at][2]]            # ours descends at once, same score
    assert g["pops"][at + 1] != [v, lv - 1, g["pops"][at][2]]         # the reference does not
    assert sorted(m[0] for m in t.get_molecules()) == sorted(m[0] for m in g["molecules"])
    assert len(pq.log) == g["n_pops"]


@pytest.mark.parametrize("tag", ["t64", "t1024"])
def test_traverser_reproduces_reference_traversal_tanimoto_scores(tag):
    from rad_amd.priority_queue import InProcessPQ
    from rad_amd.traverser import RADTraverser
    z = load_graph_npz(f"g1{tag}_graph.npz")
    bits = np.unpackbits(z["fps"], axis=1).astype(np.int64)
    for c in golden()[f"g1{tag}"]:
        qb = np.unpackbits(z["queries"][c["query"]]).astype(np.int64)
        a = bits @ qb
        o = bits.sum(1) + qb.sum() - a

        def score(smiles, a=a, o=o):
            i = int(smiles[1:])
            return 0.0 if o[i] == 0 else float(np.float32(1.0) - np.float32(a[i]) / np.float32(o[i]))
        pq = _PopLogPQ(InProcessPQ())
        t = RADTraverser(hnsw_service=_toy_service(z), scoring_fn=score, priority_queue=pq)
        t.prime()
        t.traverse(n_workers=1, n_to_score=c["n_to_score"])
        assert [p[0] for p in pq.log] == c["pop_nodes"]
        assert [p[1] for p in pq.log] == c["pop_levels"]
        mols = t.get_molecules()
        assert [m[0] for m in mols] == c["slots"] and [m[1] for m in mols] == c["scores"]


def test_multi_worker_traversal_has_no_duplicates_and_terminates():
    from rad_amd.traverser import RADTraverser
    z = load_graph_npz("g1_graph.npz")
    t = RADTraverser(hnsw_service=_toy_service(z), scoring_fn=_hash_score)
    t.prime()
    t.traverse(n_workers=4, n_to_score=120)
    mols = t.get_molecules()
    assert len(mols) >= 120
    assert len({m[0] for m in mols}) == len(mols)
    assert all(isinstance(m[0], int) and isinstance(m[1], float) for m in mols)
    with pytest.raises(ValueError):
        RADTraverser(hnsw_service=_toy_service(z), scoring_fn=_hash_score).traverse(n_workers=1)
    t3 = RADTraverser(hnsw_service=_toy_service(z), scoring_fn=lambda s: (__import__("time").sleep(0.01), 1.0)[1])
    t3.prime()
    import time
    t0 = time.time()
    t3.traverse(n_workers=2, timeout=0.3)
    assert time.time() - t0 < 3.0


# ----------------------------------------------------------------- HNSW service + Index host logic
class MockHNSW:  # reference: tests/test_redis_auth.py:24-43
    max_level, connectivity, dtype, ndim, capacity, memory_usage, multi = 3, 16, "float32", 256, 1000, 1024, False

    def __len__(self):
        return 100

    def get_neighbors(self, node_id, level):
        return [1, 101, 2, 102, 3, 103]

    def get_top_level_nodes(self):
        return [0, 100, 1, 101, 2, 102]


def test_local_hnsw_service_matches_reference_round_trip(tmp_path):
    from rad_amd.hnsw_service import LocalHNSWService, create_local_hnsw_service, service_registry
    g = golden()["g5"]
    svc = LocalHNSWService(MockHNSW())
    assert svc.get_neighbors(0, 0) == g["neighbors_no_db"]
    assert svc.get_top_level_nodes() == g["top_no_db"]
    assert svc.get_hnsw_info() == g["hnsw_info"]
    assert sorted(svc.get_service_info().keys()) == g["service_info_keys"]
    assert svc.is_healthy() == g["healthy"]
    svc.shutdown()
    assert svc.is_healthy() == g["healthy_after_shutdown"]
    with pytest.raises(RuntimeError):
        svc.get_neighbors(0, 0)
    db = str(tmp_path / "m.db")
    con = sqlite3.connect(db)
    con.execute("CREATE TABLE nodes (node_key INTEGER PRIMARY KEY, smi TEXT NOT NULL)")
    con.executemany("INSERT INTO nodes VALUES (?, ?)", [(100, "C"), (101, "CC"), (103, "CCCC")])
    con.commit()
    con.close()
    svc = create_local_hnsw_service(MockHNSW(), database_path=db)
    assert svc.get_neighbors(0, 0) == g["neighbors_db"]
    assert svc.get_top_level_nodes() == g["top_db"]
    assert svc.get_neighbors_many([(0, 0), (5, 1)]) == [g["neighbors_db"], g["neighbors_db"]]
    assert service_registry.get_service() is svc and service_registry.get_service("local") is svc
    assert svc.get_service_info()["request_count"] == 3

    class Broken(MockHNSW):
        def get_neighbors(self, node_id, level):
            raise KeyError("no such node")
    with pytest.raises(RuntimeError, match="HNSW request failed"):
        LocalHNSWService(Broken()).get_neighbors(1, 1)
    service_registry.shutdown_all()
    with pytest.raises(ValueError):
        service_registry.get_service()


def test_index_serves_adjacency_without_gpu(tmp_path):
    """An Index holding only a graph (exclude_vectors) answers RAD's adjacency calls from the
    library's host mirror — as the reference's server does with view=True, exclude_vectors=True."""
    from rad_amd.hnsw_service import LocalHNSWService
    from rad_amd.index import Index
    from rad_amd._lib import RadHipError
    z = load_graph_npz("g1t64_graph.npz")
    n = z["levels"].shape[0]
    keys = np.arange(n, dtype=np.uint64) * 7 + 1000
    idx = Index(ndim=64, dtype="b1", metric="tanimoto", connectivity=4, expansion_add=20)
    idx.load_graph(keys, None, z["levels"], z["adj0"], z["upper_row"], z["adjU"], int(z["max_level"]), int(z["entry"]))
    assert len(idx) == n and idx.max_level == int(z["max_level"]) and idx.connectivity == 4
    assert idx.ndim == 64 and idx.multi is False and str(idx.dtype) == "b1" and idx.capacity == n
    row = [int(x) for x in z["adj0"][5] if x != NO_SLOT]
    flat = [int(x) for x in idx.get_neighbors(5, 0)]
    assert flat[0::2] == row and flat[1::2] == [int(keys[r]) for r in row]
    tops = np.nonzero(z["levels"] == int(z["max_level"]))[0]
    assert [int(x) for x in idx.get_top_level_nodes()][0::2] == tops.tolist()
    assert idx.get_node_ids_from_keys([1007, 1000]).tolist() == [1, 0]
    with pytest.raises(RadHipError):
        idx.get_neighbors(5, int(z["levels"][5]) + 1)     # node absent on that level
    with pytest.raises(RadHipError):
        idx.get_neighbors(n + 5, 0)
    svc = LocalHNSWService(idx)
    out = svc.get_neighbors(5, 0)
    assert out[0::2] == row and all(s == "" for s in out[1::2])
    assert svc.get_hnsw_info()["max_level"] == int(z["max_level"]) and svc.get_hnsw_info()["size"] == n
    ls = idx.levels_stats
    assert ls[0].nodes == n and ls[0].edges == int((z["adj0"] != NO_SLOT).sum())
    # duplicate / self / out-of-range targets are rejected at load time
    bad = z["adj0"].copy()
    bad[3, 1] = bad[3, 0]
    with pytest.raises(RadHipError):
        Index(ndim=64, connectivity=4).load_graph(None, None, z["levels"], bad, z["upper_row"], z["adjU"],
                                                   int(z["max_level"]), int(z["entry"]))
    # save / load round trip of a vector-less index
    p = str(tmp_path / "idx.npz")
    idx.save(p)
    idx2 = Index(path=p, view=True, exclude_vectors=True)
    assert [int(x) for x in idx2.get_neighbors(5, 0)] == flat and len(idx2) == n


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under rad_amd/ imports, loads or links it, and
    bench.py reaches it only inside its cpu_baseline leg."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "rad_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".hip", ".inc", ".h", "Makefile")):
            text = open(path, encoding="utf-8", errors="replace").read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, re.M), path
            assert "librad_oracle" not in text and not re.search(r"#\s*include[^\n]*oracle", text), path   # comments may cite it
    bench = open(os.path.join(root, "bench.py")).read()
    uses = [m.start() for m in re.finditer(r"\boracle\b.*import|import.*\boracle\b|from oracle", bench)]
    assert uses, "bench.py's cpu_baseline leg times the oracle"
    start = bench.index("def cpu_baseline")
    nxt = bench.find("\ndef ", start + 1)
    end = nxt if nxt >= 0 else bench.index("\nif __name__", start)
    assert all(start <= u < end for u in uses), "oracle referenced outside cpu_baseline()"


def test_key_map_lives_in_the_library():
    """key <-> slot map behind the C ABI (host memory, no device): unsorted keys, duplicates (lowest slot wins),
    unknown keys, partial set_keys (identity keys elsewhere), a million keys without a Python dict."""
    import ctypes as C
    from rad_amd import _lib
    from rad_amd._lib import check, ptr
    from rad_amd.index import Index
    L = _lib.lib()
    z = load_graph_npz("g1t64_graph.npz")
    n = z["levels"].shape[0]
    rng = np.random.default_rng(0)
    keys = rng.permutation(np.arange(n, dtype=np.uint64) * 3 + 50)
    keys[7] = keys[3]                                        # a duplicate: slot 3 wins
    idx = Index(ndim=64, connectivity=4, expansion_add=20)
    idx.load_graph(keys, None, z["levels"], z["adj0"], z["upper_row"], z["adjU"], int(z["max_level"]), int(z["entry"]))
    assert np.array_equal(idx.keys, keys) and np.array_equal(idx.keys_of([0, 7, n - 1]), keys[[0, 7, n - 1]])
    probe = np.array([keys[10], keys[3], keys[n - 1]], np.uint64)
    assert idx.get_node_ids_from_keys(probe).tolist() == [10, 3, n - 1]
    with pytest.raises(KeyError):
        idx.get_node_ids_from_keys([np.uint64(1)])          # not a key of this index
    out = np.empty(2, np.uint32)
    missing = C.c_uint64(0)
    q = np.array([keys[5], 2], np.uint64)
    check(L.radhip_slots_from_keys(idx._dev._h, ptr(q), 2, ptr(out), C.byref(missing)))
    assert out.tolist() == [5, NO_SLOT] and missing.value == 1
    with pytest.raises(_lib.RadHipError):
        idx.keys_of([n])                                     # slot out of range
    # keys given for a part of the slots only: the rest keep the identity key
    part = Index(ndim=64, connectivity=4, expansion_add=20)
    part.load_graph(None, None, z["levels"], z["adj0"], z["upper_row"], z["adjU"], int(z["max_level"]), int(z["entry"]))
    part._set_keys(100, np.array([9000, 9001], np.uint64))
    assert part.keys_of([99, 100, 101, 102]).tolist() == [99, 9000, 9001, 102]
    assert part.get_node_ids_from_keys([9001, 5]).tolist() == [101, 5]
    # size: a million-node graph, lookups through the sorted array
    big_n = 1_000_000
    levels = np.zeros(big_n, np.int8); levels[0] = 1
    adj0 = np.full((big_n, 8), NO_SLOT, np.uint32); adj0[:, 0] = (np.arange(big_n) + 1) % big_n
    adjU = np.full((1, 4), NO_SLOT, np.uint32)
    upper_row = np.full(big_n, NO_SLOT, np.uint32); upper_row[0] = 0
    bk = (np.arange(big_n, dtype=np.uint64)[::-1] * 11).copy()
    big = Index(ndim=64, connectivity=4, expansion_add=20)
    big.load_graph(bk, None, levels, adj0, upper_row, adjU, 1, 0)
    want = rng.integers(0, big_n, 1000)
    assert np.array_equal(big.get_node_ids_from_keys(bk[want]), want.astype(np.uint64))
    assert [int(x) for x in big.get_neighbors(5, 0)] == [6, int(bk[6])]


def test_http_front_serves_the_reference_json_shapes(tmp_path):
    """the thin HTTP front (rad_amd/hnsw_server.py) answers the routes and JSON keys the
    reference's RemoteHNSWService client reads (rad/hnsw_server.py:505-511, 538-543, 561-568, 604-613), from an
    index that holds only a graph — no GPU involved."""
    from starlette.testclient import TestClient
    from rad_amd.hnsw_server import create_app
    from rad_amd.index import Index
    z = load_graph_npz("g1t64_graph.npz")
    n = z["levels"].shape[0]
    keys = np.arange(n, dtype=np.uint64) * 7 + 1000
    idx = Index(ndim=64, dtype="b1", metric="tanimoto", connectivity=4, expansion_add=20)
    idx.load_graph(keys, None, z["levels"], z["adj0"], z["upper_row"], z["adjU"], int(z["max_level"]), int(z["entry"]))
    db = str(tmp_path / "nodes.db")
    con = sqlite3.connect(db)
    con.execute("CREATE TABLE nodes (node_key INTEGER PRIMARY KEY, smi TEXT NOT NULL)")
    con.executemany("INSERT INTO nodes VALUES (?, ?)", [(int(k), f"C{i}") for i, k in enumerate(keys) if i % 3])   # every third key missing
    con.commit(); con.close()
    c = TestClient(create_app(idx, database_path=db, api_key="s3cret"))
    hdr = {"Authorization": "Bearer s3cret"}
    assert c.get("/ping").json() == {"pong": True}
    assert c.get("/neighbors/5/0").status_code == 401                       # bearer auth, as the reference's
    r = c.get("/neighbors/5/0", headers=hdr).json()
    row = [int(x) for x in z["adj0"][5] if x != NO_SLOT]
    assert set(r) == {"node_id", "level", "neighbors", "neighbor_count", "request_id"}
    assert r["node_id"] == 5 and r["level"] == 0 and r["neighbor_count"] == len(row)
    assert r["neighbors"][0::2] == row and r["neighbors"][1::2] == [f"C{x}" if x % 3 else "" for x in row]
    t = c.get("/top-level-nodes", headers=hdr).json()
    tops = np.nonzero(z["levels"] == int(z["max_level"]))[0].tolist()
    assert set(t) == {"top_nodes", "node_count", "cached", "request_id"} and t["node_count"] == len(tops) and t["top_nodes"][0::2] == tops
    assert len(t["top_nodes"]) == 2 * t["node_count"] and all(isinstance(s, str) for s in t["top_nodes"][1::2])
    h = c.get("/health").json()
    assert h["status"] == "healthy" and h["hnsw_size"] == n and h["hnsw_max_level"] == int(z["max_level"])
    i = c.get("/info", headers=hdr).json()
    assert i["service_type"] == "RemoteHNSWService" and i["hnsw_info"]["size"] == n and i["authentication_enabled"] is True
    assert set(i["hnsw_info"]) >= {"max_level", "size", "connectivity", "dtype", "ndim", "capacity", "memory_usage", "multi"}
    # range errors are 400s with the reference's messages
    assert c.get(f"/neighbors/{n + 3}/0", headers=hdr).status_code == 400
    assert c.get(f"/neighbors/5/{int(z['max_level']) + 1}", headers=hdr).status_code == 400
    assert c.get(f"/neighbors/5/{int(z['levels'][5]) + 1}", headers=hdr).status_code == 400 or int(z["levels"][5]) == int(z["max_level"])
    m = c.post("/neighbors-many", headers=hdr, json={"pairs": [[5, 0], [6, 0]]}).json()
    assert [x["neighbors"] for x in m["results"]] == [c.get("/neighbors/5/0", headers=hdr).json()["neighbors"],
                                                      c.get("/neighbors/6/0", headers=hdr).json()["neighbors"]]


def test_last_words_survive_a_fatal_signal():
    """radhip_arm_last_words: while armed, SIGABRT (what the HIP runtime raises on a GPU fault) and SIGTERM (what a launcher sends the
    surviving ranks) write the kept line to stdout and end the process with status 0; disarmed, the signal acts as before.
    bench.py --gpus N arms it with the measured replicas line before its side legs."""
    import subprocess
    import sys
    import textwrap
    prog = textwrap.dedent('''
        import os, signal, sys
        sys.path.insert(0, %r)
        from rad_amd import _lib
        L = _lib.lib()
        how = sys.argv[1]
        assert L.radhip_arm_last_words(b'{"value": 1.5, "side": {"error": "ended by a signal"}}') == 0
        if how == "disarm":
            assert L.radhip_arm_last_words(None) == 0
        print("not the line", file=sys.stderr)
        if how == "term":
            os.kill(os.getpid(), signal.SIGTERM)
        else:
            os.abort()
    ''') % ROOT
    for how, rc_ok in (("abort", True), ("term", True), ("disarm", False)):
        r = subprocess.run([sys.executable, "-c", prog, how], capture_output=True, text=True, timeout=120)
        if rc_ok:
            assert r.returncode == 0, (how, r.returncode, r.stderr[-300:])
            assert r.stdout == '{"value": 1.5, "side": {"error": "ended by a signal"}}\n'
        else:
            assert r.returncode != 0 and r.stdout == ""
