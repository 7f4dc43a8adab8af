"""The reference's own test shapes run against rad_amd with the index built on the
GPU: the assertions are the reference's (types, lengths, uniqueness, timing bounds, SMILES
format) — tests/test_hnsw_service.py, test_integration.py, test_service_layer_smiles.py and
test_end_to_end_smiles.py of keiserlab/rad, with `usearch.index.Index` -> `rad_amd.index.Index`
and no Redis."""
import sqlite3
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TEST_SMILES = ["CCO", "CCC", "CC(C)C", "c1ccccc1", "CC(=O)O", "CCN", "CO", "CC"]


def _create_test_hnsw(n=100, dim=64, seed=0):
    """tests/test_hnsw_service.py:13-28 (unseeded there)."""
    from rad_amd.index import Index
    rng = np.random.default_rng(seed)
    packed = np.packbits(rng.integers(0, 2, size=(n, dim), dtype=np.uint8), axis=1)
    hnsw = Index(ndim=dim, dtype="b1", metric="tanimoto", connectivity=4, expansion_add=20)
    hnsw.add(np.arange(n), packed)
    return hnsw


def _create_test_database(path, n=100):
    """tests/test_integration.py:16-50"""
    con = sqlite3.connect(path)
    con.execute("CREATE TABLE nodes (node_key INTEGER PRIMARY KEY, smi TEXT NOT NULL)")
    for i in range(n):
        base = TEST_SMILES[i % len(TEST_SMILES)]
        con.execute("INSERT INTO nodes (node_key, smi) VALUES (?, ?)", (i, f"{base}.{i}" if i > 0 else base))
    con.commit()
    con.close()


def _scoring_fn(smiles):
    return float(sum(ord(c) for c in smiles) % 100)      # deterministic score in [0, 100)


def test_local_service_basics_and_concurrency(gpu):
    """test_hnsw_service.py: info keys :37-39, non-empty neighbours :45-47, 5 threads x 10 requests
    :57-113, counters :208-230, post-shutdown behaviour :177-206."""
    from rad_amd.hnsw_service import LocalHNSWService
    svc = LocalHNSWService(_create_test_hnsw())
    info = svc.get_service_info()
    assert info["service_type"] == "LocalHNSWService" and info["status"] == "running"
    nb = svc.get_neighbors(0, 0)
    assert len(nb) > 0 and len(nb) % 2 == 0
    top = svc.get_top_level_nodes()
    assert len(top) > 0 and len(top) % 2 == 0
    errors, done = [], []

    def worker(tid):
        for i in range(10):
            try:
                r = svc.get_neighbors((tid * 10 + i) % 100, 0)
                assert isinstance(r, list) and len(r) % 2 == 0
                done.append(1)
            except Exception as e:  # pragma: no cover
                errors.append(e)
    th = [threading.Thread(target=worker, args=(t,)) for t in range(5)]
    [t.start() for t in th]
    [t.join(30) for t in th]
    assert not errors and len(done) == 50
    s = svc.get_service_info()
    assert s["request_count"] >= 52 and s["error_count"] == 0
    hi = svc.get_hnsw_info()
    assert hi["size"] == 100 and hi["ndim"] == 64 and hi["connectivity"] == 4 and hi["multi"] is False
    svc.shutdown()
    assert svc.is_healthy() is False
    with pytest.raises(RuntimeError):
        svc.get_neighbors(0, 0)


def test_smiles_format_with_and_without_database(gpu, tmp_path):
    """test_service_layer_smiles.py:101-136 (int/str alternation, non-empty SMILES with a DB) and
    :150-190 (every SMILES is "" without one)."""
    from rad_amd.hnsw_service import create_local_hnsw_service
    hnsw = _create_test_hnsw()
    db = str(tmp_path / "t.db")
    _create_test_database(db)
    svc = create_local_hnsw_service(hnsw, database_path=db)
    for data in (svc.get_neighbors(0, 0), svc.get_top_level_nodes()):
        assert len(data) % 2 == 0 and len(data) > 0
        for i in range(0, len(data), 2):
            assert isinstance(data[i], int) and isinstance(data[i + 1], str) and data[i + 1] != ""
    svc.shutdown()
    plain = create_local_hnsw_service(hnsw)
    data = plain.get_neighbors(0, 0)
    assert all(isinstance(data[i], int) and data[i + 1] == "" for i in range(0, len(data), 2))
    plain.shutdown()


def test_integration_traversals(gpu, trav_mode, tmp_path):
    """test_integration.py: 1 worker, 30 molecules, < 10 s :89-120; 3 workers no duplicate keys
    :133-163; n_to_score and timeout terminations :202-247; 4 workers / 50 unique :249-277."""
    from rad_amd.hnsw_service import create_local_hnsw_service
    from rad_amd.traverser import RADTraverser
    hnsw = _create_test_hnsw()
    db = str(tmp_path / "t.db")
    _create_test_database(db)
    # single worker
    trav = RADTraverser(hnsw_service=create_local_hnsw_service(hnsw, database_path=db), scoring_fn=_scoring_fn)
    trav.prime()
    t0 = time.time()
    trav.traverse(n_workers=1, n_to_score=30)
    assert time.time() - t0 < 10
    results = list(trav.scored_set)
    assert len(results) >= 30
    for key, score in results[:5]:
        assert isinstance(key, int) and isinstance(score, (int, float)) and 0 <= score <= 100
    best = trav.get_best_molecules(5)                                   # test_end_to_end_smiles.py:168-182
    assert all(isinstance(a, int) and isinstance(b, float) and isinstance(c, str) and c != "" for a, b, c in best)
    assert len(trav.get_molecules(3)) == 3
    trav.shutdown()
    # multi worker
    for n_workers, target in ((3, 40), (4, 50)):
        t = RADTraverser(hnsw_service=create_local_hnsw_service(hnsw, database_path=db), scoring_fn=_scoring_fn)
        t.prime()
        t0 = time.time()
        t.traverse(n_workers=n_workers, n_to_score=target)
        assert time.time() - t0 < 15
        keys = [k for k, _ in t.scored_set]
        assert len(keys) >= target and len(set(keys)) == len(keys)
        stats = t.get_traversal_stats()
        assert stats["coordination"]["scored_molecules"] == len(keys)
        t.shutdown()
    # timeout termination
    slow = RADTraverser(hnsw_service=create_local_hnsw_service(hnsw, database_path=db),
                        scoring_fn=lambda s: (time.sleep(0.05), 1.0)[1])
    slow.prime()
    t0 = time.time()
    slow.traverse(n_workers=1, timeout=2)
    assert time.time() - t0 <= 3.5
    slow.shutdown()
    with pytest.raises(ValueError):
        RADTraverser(hnsw_service=create_local_hnsw_service(hnsw), scoring_fn=_scoring_fn).traverse(n_workers=1)


def test_service_registry_and_convenience_function(gpu):
    """test_hnsw_service.py:115-175: named services, default service, listing, shutdown_all; the
    convenience function registers its service as the default."""
    from rad_amd.hnsw_service import LocalHNSWService, ServiceRegistry, create_local_hnsw_service, service_registry
    registry = ServiceRegistry()
    hnsw = _create_test_hnsw()
    s1, s2 = LocalHNSWService(hnsw), LocalHNSWService(hnsw)
    try:
        registry.register_service("primary", s1, is_default=True)
        registry.register_service("secondary", s2)
        assert registry.get_service("primary") is s1 and registry.get_service("secondary") is s2
        assert registry.get_service() is s1
        listed = registry.list_services()
        assert "primary" in listed and "secondary" in listed
        assert registry.get_service("primary").get_neighbors(0, 0) is not None
        assert registry.get_service("secondary").get_neighbors(0, 0) is not None
    finally:
        registry.shutdown_all()
    assert not s1.is_healthy() and not s2.is_healthy()
    service_registry.shutdown_all()
    svc = create_local_hnsw_service(hnsw)
    try:
        assert service_registry.get_service() is svc
        assert svc.get_neighbors(0, 0) is not None
        assert svc.get_service_info()["service_type"] == "LocalHNSWService"
    finally:
        service_registry.shutdown_all()


def test_service_lifecycle(gpu, tmp_path):
    """test_integration.py:176-200: initialised and healthy after construction, not running after shutdown."""
    from rad_amd.hnsw_service import create_local_hnsw_service
    from rad_amd.traverser import RADTraverser
    hnsw = _create_test_hnsw()
    db = str(tmp_path / "t.db")
    _create_test_database(db)
    trav = RADTraverser(hnsw_service=create_local_hnsw_service(hnsw, database_path=db), scoring_fn=_scoring_fn)
    assert trav.is_initialized and trav.hnsw_service.is_healthy()
    trav.shutdown()
    assert not trav.is_running


def test_retrospective_scored_set_and_scoring_interface(gpu, tmp_path):
    """test_end_to_end_smiles.py:257-321 (scored set returns (int, float, non-empty str) triples, n
    or all) and :323-360 (the scoring function is called with non-empty SMILES strings only)."""
    from rad_amd.hnsw_service import create_local_hnsw_service
    from rad_amd.scored import InProcessScoredSet
    from rad_amd.traverser import RADTraverser
    ss = InProcessScoredSet()
    for node_id, score, smiles in [(1, 85.5, "CCO"), (2, 92.1, "c1ccccc1O"), (3, 78.3, "CC(C)O"),
                                   (4, 95.7, "c1ccc(N)cc1"), (5, 82.4, "CC(=O)O")]:
        ss.insert(node_id, score, smiles)
    mols = ss.get_molecules(3)
    assert len(mols) == 3
    for node_id, score, smiles in mols:
        assert isinstance(node_id, int) and isinstance(score, float) and isinstance(smiles, str) and smiles != ""
    assert len(ss.get_molecules()) == 5
    seen = []

    def scoring_fn(smiles):
        assert isinstance(smiles, str) and len(smiles) > 0
        assert not (set(smiles.replace(".", "")) - set("CONFClBrcnos()=[]#-+1234567890"))
        seen.append(smiles)
        return len(smiles) * 10.0
    hnsw = _create_test_hnsw()
    db = str(tmp_path / "t.db")
    _create_test_database(db)
    trav = RADTraverser(hnsw_service=create_local_hnsw_service(hnsw, database_path=db), scoring_fn=scoring_fn)
    trav.prime()
    trav.traverse(n_workers=1, n_to_score=25)
    assert len(seen) >= 25
    trav.shutdown()


def test_http_front_over_a_gpu_built_index(gpu, tmp_path):
    """SURVEY.md §8f N4 with a GPU behind it: an index BUILT on the GPU -> rad_amd.hnsw_server -> the JSON
    shapes a RemoteHNSWService client reads (rad/hnsw_server.py:505-511 /neighbors, :538-543 /top-level-nodes, :561-568
    /health, :604-613 /info), and every answer equal to what the index object itself returns."""
    from starlette.testclient import TestClient
    from rad_amd.hnsw_server import create_app
    n = 300
    hnsw = _create_test_hnsw(n=n, dim=64, seed=3)
    db = str(tmp_path / "t.db")
    _create_test_database(db, n=n)
    c = TestClient(create_app(hnsw, database_path=db, api_key="k"))
    hdr = {"Authorization": "Bearer k"}
    assert c.get("/ping").json() == {"pong": True}
    assert c.get("/neighbors/0/0").status_code == 401
    con = sqlite3.connect(db)
    smi = dict(con.execute("SELECT node_key, smi FROM nodes").fetchall())
    con.close()
    for node in (0, 7, n - 1):
        r = c.get(f"/neighbors/{node}/0", headers=hdr)
        assert r.status_code == 200
        j = r.json()
        assert set(j) == {"node_id", "level", "neighbors", "neighbor_count", "request_id"}
        flat = [int(x) for x in hnsw.get_neighbors(node, 0)]              # [slot, key, ...] straight from the index
        assert j["node_id"] == node and j["level"] == 0 and j["neighbor_count"] == len(flat) // 2 > 0
        assert j["neighbors"][0::2] == flat[0::2]
        assert j["neighbors"][1::2] == [smi[k] for k in flat[1::2]]        # keys joined to SMILES (rad/hnsw_service.py:256-283)
    t = c.get("/top-level-nodes", headers=hdr).json()
    top = [int(x) for x in hnsw.get_top_level_nodes()]
    assert set(t) == {"top_nodes", "node_count", "cached", "request_id"}
    assert t["node_count"] == len(top) // 2 and t["top_nodes"][0::2] == top[0::2]
    h = c.get("/health").json()
    assert h["status"] == "healthy" and h["hnsw_size"] == n and h["hnsw_max_level"] == hnsw.max_level
    i = c.get("/info", headers=hdr).json()
    assert i["hnsw_info"]["size"] == n and i["hnsw_info"]["ndim"] == 64 and i["hnsw_info"]["connectivity"] == 4
    assert c.get(f"/neighbors/{n}/0", headers=hdr).status_code == 400
    assert c.get(f"/neighbors/0/{hnsw.max_level + 1}", headers=hdr).status_code == 400

    # a traversal whose neighbour reads go through those routes equals one that asks the index directly
    from rad_amd.hnsw_service import HNSWService, create_local_hnsw_service
    from rad_amd.traverser import RADTraverser

    class OverHttp(HNSWService):
        def get_neighbors(self, node_id, level):
            return c.get(f"/neighbors/{node_id}/{level}", headers=hdr).json()["neighbors"]

        def get_top_level_nodes(self):
            return c.get("/top-level-nodes", headers=hdr).json()["top_nodes"]

        def get_hnsw_info(self):
            return c.get("/info", headers=hdr).json()["hnsw_info"]

        def get_service_info(self):
            return {"service_type": "OverHttp", "status": "running"}

        def is_healthy(self):
            return c.get("/health").json()["status"] == "healthy"

        def shutdown(self):
            pass

    lists = []
    for svc in (OverHttp(), create_local_hnsw_service(hnsw, database_path=db)):
        tr = RADTraverser(hnsw_service=svc, scoring_fn=_scoring_fn)
        tr.prime()
        tr.traverse(n_workers=1, n_to_score=60)
        lists.append(list(tr.scored_set))
        tr.shutdown()
    assert lists[0] == lists[1] and len(lists[0]) >= 60
