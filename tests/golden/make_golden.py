#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE
(/root/reference/rad/*.py) and driving its own control flow sequentially:

    RADTraverser.prime()                          rad/traverser.py:128-176
    loop: CoordinationService.request_work()      rad/coordination_service.py:290-347
          DistributedWorker._process_work_item()  rad/distributed_worker.py:272-333
            -> CoordinationService.submit_work_results()   :349-413

with termination `len(scored_set) >= n_to_score` checked before every request_work (the
idealised sequential semantics of SURVEY.md §3.3; the threaded reference polls it once a
second).  The `redis` client module, which this container lacks, is replaced by the
in-memory restatement in _fake_redis.py.  Run here only (needs /root/reference); the
fixtures it writes are data — inputs and expected outputs — and travel with the repo.

    python tests/golden/make_golden.py
"""
import json
import os
import sqlite3
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
import _fake_redis  # noqa: E402

sys.modules["redis"] = _fake_redis
sys.path.insert(0, REF)

from rad.coordination_service import WorkItem  # noqa: E402
from rad.distributed_worker import DistributedWorker  # noqa: E402
from rad.hnsw_service import HNSWService, LocalHNSWService  # noqa: E402
from rad.priority_queue import RedisPQ  # noqa: E402
from rad.scored import RedisScoredSet  # noqa: E402
from rad.traverser import RADTraverser  # noqa: E402
from rad.visited import RedisVisited  # noqa: E402

NO_SLOT = 0xFFFFFFFF


# ------------------------------------------------------------------ graphs
def layered_knn_graph(fps, M, cap0, seed):
    """Exact-kNN layered graph in the build's graph layout (levels, adj0, upper_row, adjU).
    numpy only — independent of both the oracle's and the product's builders."""
    rng = np.random.default_rng(seed)
    n = fps.shape[0]
    bits = np.unpackbits(fps, axis=1).astype(np.int32)
    inter = bits @ bits.T
    pop = bits.sum(1)
    union = pop[:, None] + pop[None, :] - inter
    # order by exact rational distance, ties by slot: sort on (1 - inter/union) in float64 then slot
    d = 1.0 - inter / np.maximum(union, 1)
    levels = np.minimum(np.floor(-np.log(rng.random(n)) / np.log(M)).astype(np.int64), 6)
    levels[int(rng.integers(0, n))] = levels.max() + (1 if (levels == levels.max()).sum() > 1 else 0)
    L = int(levels.max())
    adj0 = np.full((n, cap0), NO_SLOT, np.uint32)
    upper_row = np.full(n, NO_SLOT, np.uint32)
    rows = []
    for i in range(n):
        if levels[i] > 0:
            upper_row[i] = len(rows)
            rows.extend([None] * int(levels[i]))
    adjU = np.full((len(rows), M), NO_SLOT, np.uint32)
    for l in range(L + 1):
        members = np.nonzero(levels >= l)[0]
        cap = cap0 if l == 0 else M
        for i in members:
            others = members[members != i]
            if others.size == 0:
                continue
            order = np.lexsort((others, d[i, others]))
            nb = others[order[:cap]]
            if l == 0:
                adj0[i, :nb.size] = nb
            else:
                adjU[upper_row[i] + l - 1, :nb.size] = nb
    entry = int(np.nonzero(levels == L)[0][0])
    return dict(levels=levels.astype(np.int8), adj0=adj0, upper_row=upper_row, adjU=adjU,
                max_level=L, entry=entry)


def neighbors(g, slot, level):
    if level > g["levels"][slot]:
        raise KeyError((slot, level))
    row = g["adj0"][slot] if level == 0 else g["adjU"][g["upper_row"][slot] + level - 1]
    return [int(x) for x in row if x != NO_SLOT]


class ToyService(HNSWService):
    """HNSWService over an in-memory layered graph; smiles of node i is "S{i}"."""

    def __init__(self, g):
        self.g = g

    def get_neighbors(self, node_id, level):
        out = []
        for nb in neighbors(self.g, node_id, level):
            out.extend([nb, f"S{nb}"])
        return out

    def get_top_level_nodes(self):
        out = []
        for i in np.nonzero(self.g["levels"] == self.g["max_level"])[0]:
            out.extend([int(i), f"S{int(i)}"])
        return out

    def is_healthy(self):
        return True

    def shutdown(self):
        pass

    def get_service_info(self):
        return {"service_type": "ToyService"}

    def get_hnsw_info(self):
        return {"max_level": int(self.g["max_level"])}


def run_reference(g, scoring_fn, n_to_score, namespace):
    trav = RADTraverser(hnsw_service=ToyService(g), scoring_fn=scoring_fn, redis_host="fake",
                        namespace=namespace)
    trav.prime()
    cs = trav.coordination_service
    cs.register_worker("w0")
    worker = DistributedWorker(worker_id="w0", coordination_service=cs, scoring_fn=scoring_fn)
    pops = []
    while len(cs.scored_set) < n_to_score:
        item = cs.request_work("w0")
        if item is None:
            break
        pops.append([int(item.node_id), int(item.level), float(item.score)])
        ok = worker._process_work_item(item)
        assert ok, "reference failed a work item (empty neighbour row?)"
    mols = trav.get_molecules()
    best = trav.get_best_molecules(10)
    return dict(pops=pops, molecules=[[int(i), float(s), smi] for i, s, smi in mols],
                best10=[[int(i), float(s), smi] for i, s, smi in best])


def run_reference_tolerant(g, scoring_fn, n_to_score, namespace, max_items=100000):
    """run_reference without the assertion: records whether the reference's worker completed each item.
    A work item whose neighbour list is empty is FAILED by the reference (rad/distributed_worker.py:286-288):
    nothing is submitted, so the node is neither expanded nor descended; the item stays assigned until the
    coordinator's 120 s cleanup re-queues it (rad/coordination_service.py:554-580), which a sequential drive
    never reaches."""
    trav = RADTraverser(hnsw_service=ToyService(g), scoring_fn=scoring_fn, redis_host="fake", namespace=namespace)
    trav.prime()
    cs = trav.coordination_service
    cs.register_worker("w0")
    worker = DistributedWorker(worker_id="w0", coordination_service=cs, scoring_fn=scoring_fn)
    pops, oks = [], []
    while len(cs.scored_set) < n_to_score and len(pops) < max_items:
        item = cs.request_work("w0")
        if item is None:
            break
        pops.append([int(item.node_id), int(item.level), float(item.score)])
        oks.append(bool(worker._process_work_item(item)))
    mols = trav.get_molecules()
    return dict(pops=pops, ok=oks, molecules=[[int(i), float(s), smi] for i, s, smi in mols])


def hash_score(smiles):
    """Deterministic pseudo docking score from the smiles string (lower = better)."""
    h = 1469598103934665603
    for ch in smiles.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return -20.0 + (h % 100000) / 5000.0


def save_graph_npz(path, fps, g, extra):
    np.savez_compressed(path, fps=fps, levels=g["levels"], adj0=g["adj0"], upper_row=g["upper_row"],
                        adjU=g["adjU"], max_level=np.int32(g["max_level"]), entry=np.uint32(g["entry"]),
                        **extra)


def main():
    out = {}
    # ---------------- G1: hash-scored toy graph (A5-A10 control flow) -------------
    rng = np.random.default_rng(101)
    fps = rng.integers(0, 256, (400, 8), dtype=np.uint8)
    g = layered_knn_graph(fps, M=4, cap0=8, seed=7)
    res = run_reference(g, hash_score, 150, "g1")
    res_all = run_reference(g, hash_score, 10 ** 9, "g1all")  # queue drains
    save_graph_npz(os.path.join(HERE, "g1_graph.npz"), fps, g, {})
    out["g1"] = dict(n_to_score=150, **res)
    out["g1_all"] = dict(n_to_score=10 ** 9, n_pops=len(res_all["pops"]), n_scored=len(res_all["molecules"]),
                         pops_head=res_all["pops"][:50], pops_tail=res_all["pops"][-50:],
                         molecules_tail=res_all["molecules"][-50:])

    # ---------------- G7: an EMPTY adjacency row (the stated deviation) --------------
    # the first node the G1 traversal expands on an upper level loses its row on that level
    first_upper = next(p for p in res_all["pops"] if p[1] >= 1)
    v, lv = int(first_upper[0]), int(first_upper[1])
    g7 = {k: (np.array(val, copy=True) if isinstance(val, np.ndarray) else val) for k, val in g.items()}
    g7["adjU"][int(g7["upper_row"][v]) + lv - 1, :] = NO_SLOT
    r7 = run_reference_tolerant(g7, hash_score, 10 ** 9, "g7")
    save_graph_npz(os.path.join(HERE, "g7_graph.npz"), fps, g7, {})
    out["g7"] = dict(emptied_node=v, emptied_level=lv, n_pops=len(r7["pops"]), n_scored=len(r7["molecules"]),
                     pops=r7["pops"], ok=r7["ok"], failed_items=[r7["pops"][i] for i, k in enumerate(r7["ok"]) if not k],
                     molecules=r7["molecules"])

    # ---------------- G1t: Tanimoto-scored traversals (A1 + A5-A10) ----------------
    for tag, ndim, n, M, cap0, nts_list in (("t64", 64, 500, 4, 8, (60, 300, 10 ** 9)),
                                            ("t1024", 1024, 700, 8, 16, (100, 400))):
        rng = np.random.default_rng(ndim)
        if ndim == 64:
            fps = np.packbits(rng.integers(0, 2, (n, ndim), dtype=np.uint8), axis=1)
        else:
            # clustered sparse rows (ECFP-like): cluster base with flipped bits
            base = (rng.random((35, ndim)) < 0.07)
            rows = base[rng.integers(0, 35, n)]
            rows = (rows & (rng.random((n, ndim)) > 0.15)) | (rng.random((n, ndim)) < 0.015)
            fps = np.packbits(rows.astype(np.uint8), axis=1)
        g = layered_knn_graph(fps, M=M, cap0=cap0, seed=ndim + 1)
        queries = np.concatenate([fps[[3, n // 2]], np.packbits(
            (rng.random((2, ndim)) < (0.5 if ndim == 64 else 0.07)).astype(np.uint8), axis=1)])
        bits = np.unpackbits(fps, axis=1).astype(np.int64)
        cases = []
        for qi, q in enumerate(queries):
            qb = np.unpackbits(q).astype(np.int64)
            a = bits @ qb
            o = bits.sum(1) + qb.sum() - a

            def tanimoto_score(smiles, a=a, o=o):
                i = int(smiles[1:])
                if o[i] == 0:
                    return 0.0
                return float(np.float32(1.0) - np.float32(a[i]) / np.float32(o[i]))
            for nts in nts_list:
                r = run_reference(g, tanimoto_score, nts, f"{tag}_{qi}_{nts}")
                cases.append(dict(query=qi, n_to_score=nts,
                                  pop_nodes=[p[0] for p in r["pops"]], pop_levels=[p[1] for p in r["pops"]],
                                  slots=[m[0] for m in r["molecules"]], scores=[m[1] for m in r["molecules"]]))
        save_graph_npz(os.path.join(HERE, f"g1{tag}_graph.npz"), fps, g, dict(queries=queries))
        out[f"g1{tag}"] = cases

    # ---------------- G2: queue order (rad/priority_queue.py) ---------------------
    r = _fake_redis.StrictRedis()
    pq = RedisPQ(redis_client=r, queue_name="pq")
    inserts = [(10, 0, 1.5), (9, 0, 1.5), (1, 0, 1.5), (19, 0, 1.5), (2, 1, 1.5), (2, 0, 1.5), (2, 10, 1.5),
               (100, 0, 1.5), (99, 0, 1.5), (5, 0, -3.25), (6, 0, 2.0), (7, 2, 0.1), (7, 1, 0.1),
               (123456789, 0, 1.5), (12345678, 3, 1.5), (5, 0, 9.0)]  # last one overwrites (5,0)
    for nid, lv, sc in inserts:
        pq.insert(nid, lv, sc)
    popped = []
    while True:
        it = pq.pop()
        if it is None:
            break
        popped.append([it[0], it[1], it[2]])
    out["g2"] = dict(inserts=[list(x) for x in inserts], pops=popped)

    # ---------------- G3: visited semantics (rad/visited.py) -----------------------
    vs = RedisVisited(redis_client=_fake_redis.StrictRedis(), visited_name="v")
    seq = [(5, 0), (5, 1), (5, 0), (50, 0), (5, 1), (0, 0), (0, 0)]
    out["g3"] = dict(calls=[list(x) for x in seq], returns=[bool(vs.checkAndInsert(a, b)) for a, b in seq])

    # ---------------- scored set (rad/scored.py) ------------------------------------
    ss = RedisScoredSet(redis_client=_fake_redis.StrictRedis(), scored_name="s")
    ins = [(7, 1.25, "CCO"), (3, -4.5, "c1ccccc1"), (7, 99.0, "XX"), (11, 0.0, ""), (2, -4.5, "N")]
    for a, b, c in ins:
        ss.insert(a, b, c)
    out["scored"] = dict(inserts=[list(x) for x in ins], length=len(ss),
                         molecules=[list(x) for x in ss.get_molecules()],
                         first2=[list(x) for x in ss.get_molecules(2)],
                         best=[list(x) for x in ss.get_best_molecules()],
                         best2=[list(x) for x in ss.get_best_molecules(2)],
                         get7=ss.getScore(7), get_missing=ss.getScore(12345),
                         iter=[list(x) for x in ss])

    # ---------------- G4/G5: LocalHNSWService round trip (rad/hnsw_service.py) -------
    class MockHNSW:  # tests/test_redis_auth.py:24-43
        max_level, connectivity, dtype, ndim, capacity, memory_usage, multi = 3, 16, "float32", 256, 1000, 1024, False

        def __len__(self):
            return 100

        def get_neighbors(self, node_id, level):
            return [1, 101, 2, 102, 3, 103]

        def get_top_level_nodes(self):
            return [0, 100, 1, 101, 2, 102]
    svc = LocalHNSWService(MockHNSW())
    g5 = dict(neighbors_no_db=svc.get_neighbors(0, 0), top_no_db=svc.get_top_level_nodes(),
              hnsw_info=svc.get_hnsw_info(), service_info_keys=sorted(svc.get_service_info().keys()),
              healthy=svc.is_healthy())
    svc.shutdown()
    g5["healthy_after_shutdown"] = svc.is_healthy()
    with tempfile.TemporaryDirectory() as td:
        db = os.path.join(td, "m.db")
        con = sqlite3.connect(db)
        con.execute("CREATE TABLE nodes (node_key INTEGER PRIMARY KEY, smi TEXT NOT NULL)")
        con.executemany("INSERT INTO nodes VALUES (?, ?)", [(100, "C"), (101, "CC"), (103, "CCCC")])
        con.commit()
        con.close()
        svc = LocalHNSWService(MockHNSW(), database_path=db)
        g5["neighbors_db"] = svc.get_neighbors(0, 0)
        g5["top_db"] = svc.get_top_level_nodes()
        svc.shutdown()
    out["g5"] = g5

    # ---------------- G6: WorkItem dict shape ------------------------------------------
    wi = WorkItem(5, 2, -1.5, request_id="rid", neighbors=[1, "C"])
    d = wi.to_dict()
    out["g6"] = dict(keys=sorted(d.keys()), node_id=d["node_id"], level=d["level"], score=d["score"],
                     neighbors=d["neighbors"], roundtrip=sorted(WorkItem.from_dict(d).to_dict().keys()))

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f)
    print("wrote golden.json:", {k: (len(v) if hasattr(v, "__len__") else v) for k, v in out.items()})


if __name__ == "__main__":
    main()
