"""Drop-in check against the REAL reference code (CPU, this container only).

`rad_amd.index.Index` is handed to the reference's own `LocalHNSWService` (rad/hnsw_service.py:95-108,
which forks a server process around it) and the reference's own `RADTraverser` is driven over it —
with the in-memory `redis` stand-in of tests/golden/_fake_redis.py, as in make_golden.py.  The same
traversal is then run through rad_amd's own service + traverser: both stacks must return the same
molecules in the same order.  Needs /root/reference, so it is skipped on the GPU box (the reference
does not travel); the graph comes from a committed fixture, adjacency reads need no GPU."""
import os
import sqlite3
import sys

import numpy as np
import pytest

from golden_util import GOLDEN_DIR, load_graph_npz

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "rad")),
                                reason="the reference tree is only present in the build container")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, GOLDEN_DIR)
    import _fake_redis
    saved = sys.modules.get("redis")
    sys.modules["redis"] = _fake_redis
    sys.path.insert(0, REF)
    try:
        import rad.hnsw_service as ref_service
        import rad.traverser as ref_traverser
        import rad.distributed_worker as ref_worker
        yield dict(service=ref_service, traverser=ref_traverser, worker=ref_worker)
    finally:
        sys.path.remove(REF)
        sys.path.remove(GOLDEN_DIR)
        if saved is None:
            sys.modules.pop("redis", None)
        else:
            sys.modules["redis"] = saved


def _index_and_db(tmp_path):
    from rad_amd.index import Index
    z = load_graph_npz("g1t1024_graph.npz")
    n = z["levels"].shape[0]
    keys = np.arange(n, dtype=np.uint64) * 3 + 1000          # keys differ from slots
    idx = Index(ndim=1024, dtype="b1", metric="tanimoto", connectivity=int(z["adjU"].shape[1]),
                connectivity_base=int(z["adj0"].shape[1]))
    idx.load_graph(keys, None, z["levels"], z["adj0"], z["upper_row"], z["adjU"], int(z["max_level"]), int(z["entry"]))
    db = str(tmp_path / "mols.db")
    con = sqlite3.connect(db)
    con.execute("CREATE TABLE nodes (node_key INTEGER PRIMARY KEY, smi TEXT NOT NULL)")   # README.md:74-79
    con.executemany("INSERT INTO nodes VALUES (?, ?)", [(int(k), f"C{int(k)}") for k in keys[: n - 5]])  # last 5 missing -> ""
    con.commit()
    con.close()
    return idx, keys, db, z


def _score(smiles):
    h = 1469598103934665603
    for ch in smiles.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return -20.0 + (h % 100000) / 5000.0


def test_reference_local_service_accepts_rad_amd_index(ref, tmp_path):
    idx, keys, db, z = _index_and_db(tmp_path)
    svc = ref["service"].LocalHNSWService(idx, database_path=db)
    try:
        assert svc.is_healthy()
        info = svc.get_hnsw_info()
        assert info["max_level"] == int(z["max_level"]) and info["size"] == len(idx)
        assert info["connectivity"] == idx.connectivity and info["ndim"] == 1024
        n = len(idx)
        for slot, level in ((0, 0), (int(z["entry"]), int(z["max_level"])), (n - 1, 0), (n - 3, 0)):
            got = svc.get_neighbors(slot, level)
            row = z["adj0"][slot] if level == 0 else z["adjU"][z["upper_row"][slot] + level - 1]
            want = []
            for s in row[row != 0xFFFFFFFF]:
                want.extend([int(s), f"C{int(keys[s])}" if s < n - 5 else ""])
            assert got == want
        top = svc.get_top_level_nodes()
        assert top[0::2] == [int(i) for i in np.flatnonzero(z["levels"] == z["max_level"])]
        with pytest.raises(RuntimeError):
            svc.get_neighbors(0, int(z["levels"][0]) + 1)       # node absent on that level
    finally:
        svc.shutdown()


def _drive_reference(ref, service, n_to_score, namespace):
    trav = ref["traverser"].RADTraverser(hnsw_service=service, scoring_fn=_score, redis_host="fake", namespace=namespace)
    trav.prime()
    cs = trav.coordination_service
    cs.register_worker("w0")
    worker = ref["worker"].DistributedWorker(worker_id="w0", coordination_service=cs, scoring_fn=_score)
    pops = []
    while len(cs.scored_set) < n_to_score:
        item = cs.request_work("w0")
        if item is None:
            break
        pops.append((int(item.node_id), int(item.level)))
        assert worker._process_work_item(item)
    return pops, [(int(i), float(s), smi) for i, s, smi in trav.get_molecules()]


@pytest.mark.parametrize("n_to_score", [40, 250])
def test_reference_traverser_over_rad_amd_index_equals_rad_amd_stack(ref, tmp_path, n_to_score):
    idx, keys, db, z = _index_and_db(tmp_path)
    ref_svc = ref["service"].LocalHNSWService(idx, database_path=db)
    try:
        ref_pops, ref_mols = _drive_reference(ref, ref_svc, n_to_score, f"dropin{n_to_score}")
    finally:
        ref_svc.shutdown()
    from rad_amd.hnsw_service import create_local_hnsw_service
    from rad_amd.traverser import RADTraverser
    mine = RADTraverser(hnsw_service=create_local_hnsw_service(idx, database_path=db), scoring_fn=_score)
    try:
        mine.prime()
        mine.traverse(n_workers=1, n_to_score=n_to_score)
        got = [(int(i), float(s), smi) for i, s, smi in mine.get_molecules()]
    finally:
        mine.shutdown()
    assert len(ref_mols) >= n_to_score
    assert got == ref_mols


def test_reference_http_server_accepts_rad_amd_index(ref, tmp_path):
    """The reference's FastAPI app (rad/hnsw_server.py:85-135) constructed around a rad_amd Index:
    /neighbors, /top-level-nodes, /info and /health answer with the index's data (attribute
    surface rad/hnsw_server.py:148-161 incl. levels_stats)."""
    import rad.hnsw_server as ref_server
    from starlette.testclient import TestClient
    idx, keys, db, z = _index_and_db(tmp_path)
    app = ref_server.HNSWServerApp(idx, database_path=db)
    n = len(idx)
    with TestClient(app.app) as client:
        assert client.get("/ping").status_code == 200
        r = client.get("/neighbors/0/0")
        assert r.status_code == 200
        row = z["adj0"][0]
        want = []
        for s in row[row != 0xFFFFFFFF]:
            want.extend([int(s), f"C{int(keys[s])}" if s < n - 5 else ""])
        assert r.json()["neighbors"] == want
        top = client.get("/top-level-nodes").json()
        assert top["top_nodes"][0::2] == [int(i) for i in np.flatnonzero(z["levels"] == z["max_level"])]
        assert top["node_count"] == len(top["top_nodes"]) // 2
        info = client.get("/info").json()
        assert info["hnsw_info"]["max_level"] == int(z["max_level"]) and info["hnsw_info"]["size"] == n
        assert client.get("/health").status_code == 200
        assert client.get(f"/neighbors/{n}/0").status_code >= 400                           # no such node
        assert client.get(f"/neighbors/0/{int(z['max_level']) + 1}").status_code >= 400     # level > max_level
        assert client.get(f"/neighbors/0/{int(z['levels'][0]) + 1}").status_code >= 400 or int(z["levels"][0]) == int(z["max_level"])
