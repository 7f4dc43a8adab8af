"""RAD traverser (reference: rad/traverser.py).

`RADTraverser` keeps the reference's constructor, `prime()` (rad/traverser.py:128-176),
`traverse(n_workers, timeout, n_to_score)` (:178-245), result accessors (:273-344), `shutdown()`
and the three factory functions (:387-420).  Traversal state is in-process (rad_amd
.priority_queue / .visited / .scored) — the `redis_*` arguments are accepted and ignored, no
redis-server is spawned (rad/redis_server.py is an external-daemon launcher, out of scope).

With `n_workers == 1` the expansion loop runs inline and deterministically: the termination
conditions are checked before every pop, so exactly the idealised sequential semantics of the
reference are produced (the threaded reference polls them once a second and overshoots;
its tests assert only `>=`).  With `n_workers > 1`, worker threads race for the queue as in the
reference.

`TanimotoRADTraverser` runs the same traversal entirely on the GPU for the case
scoring_fn = Tanimoto distance to a query fingerprint (rad_amd/csrc/traverse.hip), many
independent queries at once.
"""
from __future__ import annotations

import logging
import time
from typing import Any, Callable, Dict, List, Optional

import numpy as np

from .coordination_service import CoordinationService, create_coordination_service
from .distributed_worker import DistributedWorker, WorkerPool, create_worker_pool
from .hnsw_service import HNSWService, create_local_hnsw_service

logger = logging.getLogger(__name__)


class RADTraverser:
    def __init__(self, hnsw_service: HNSWService, scoring_fn: Callable, deployment_mode: str = "local",
                 redis_host: Optional[str] = None, redis_port: int = 6379,
                 redis_password: Optional[str] = None, namespace: Optional[str] = None, **kwargs):
        self.hnsw_service = hnsw_service
        self.scoring_fn = scoring_fn
        self.deployment_mode = deployment_mode
        self.namespace = namespace or f"rad_session_{int(time.time())}"
        self.coordination_service: Optional[CoordinationService] = None
        self.redis_client = None
        self.redis_server = None
        self.workers: List[DistributedWorker] = []
        self.worker_pool: Optional[WorkerPool] = None
        self.is_initialized = False
        self.is_running = False
        if not self.hnsw_service.is_healthy():
            raise RuntimeError("Provided HNSW service is not healthy")
        cs_kwargs = {k: v for k, v in kwargs.items()
                     if k in ("worker_timeout", "heartbeat_interval", "priority_queue", "visited_set", "scored_set")}
        self.coordination_service = create_coordination_service(None, self.hnsw_service,
                                                               namespace=self.namespace, **cs_kwargs)
        self.is_initialized = True

    # -- prime -----------------------------------------------------------------
    def prime(self, **kwargs):
        """Score the top-level nodes and seed visited/queue on level max(0, max_level - 1)
        (rad/traverser.py:141-170; the start level is one below the index's max_level)."""
        if not self.is_initialized:
            raise RuntimeError("Services not initialized")
        cs = self.coordination_service
        top = self.hnsw_service.get_top_level_nodes()
        start_level = max(0, self.hnsw_service.get_hnsw_info().get("max_level", 1) - 1)
        for i in range(0, len(top), 2):
            node_id, smiles = top[i], top[i + 1]
            score = self.scoring_fn(smiles, **kwargs)
            cs.scored_set.insert(node_id=node_id, score=score, smiles=smiles)
            cs.visited_set.checkAndInsert(node_id=node_id, level=start_level)
            cs.priority_queue.insert(node_id=node_id, level=start_level, score=score)

    # -- traverse ----------------------------------------------------------------
    def traverse(self, n_workers: int, timeout: Optional[float] = None,
                 n_to_score: Optional[int] = None, **kwargs):
        if not self.is_initialized:
            raise RuntimeError("Services not initialized")
        if timeout is None and n_to_score is None:
            raise ValueError("Must provide either timeout or n_to_score")
        cond: Dict[str, Any] = {}
        if timeout is not None:
            cond["timeout"] = timeout
        if n_to_score is not None:
            cond["n_to_score"] = n_to_score
        cs = self.coordination_service
        try:
            cs.start(cond)
            self.is_running = True
            if n_workers == 1:
                worker = DistributedWorker(worker_id=f"{self.namespace}_worker_0", coordination_service=cs,
                                           scoring_fn=self.scoring_fn, **kwargs)
                cs.register_worker(worker.worker_id, worker.worker_type, worker.capabilities)
                worker.started_at = time.time()
                self.workers.append(worker)
                while True:
                    done, reason = cs.check_termination()
                    if done:
                        break
                    item = cs.request_work(worker.worker_id)
                    if item is None:
                        break
                    if worker._process_work_item(item):
                        worker.work_completed += 1
                    else:
                        worker.work_failed += 1
            else:
                self.worker_pool = create_worker_pool(
                    n_workers, worker_id_prefix=f"{self.namespace}_worker", coordination_service=cs,
                    scoring_fn=self.scoring_fn, **kwargs)
                if not self.worker_pool.start_all():
                    raise RuntimeError("Failed to start worker pool")
                self._monitor_traversal()
                self.worker_pool.stop_all()
            self.is_running = False
            _, reason = cs.check_termination()
            cs.shutdown(reason or "Traversal complete")
        except Exception:
            self.shutdown()
            raise

    def _monitor_traversal(self, poll: float = 0.005):
        cs = self.coordination_service
        while self.is_running:
            done, _ = cs.check_termination()
            if done:
                break
            time.sleep(poll)

    # -- accessors -----------------------------------------------------------------
    @property
    def scored_set(self):
        return self.coordination_service.scored_set

    @property
    def priority_queue(self):
        return self.coordination_service.priority_queue

    @property
    def visited_set(self):
        return self.coordination_service.visited_set

    def get_traversal_stats(self) -> Dict[str, Any]:
        stats = {"deployment_mode": self.deployment_mode, "namespace": self.namespace,
                 "is_initialized": self.is_initialized, "is_running": self.is_running}
        if self.coordination_service:
            stats["coordination"] = self.coordination_service.get_coordination_stats()
        if self.hnsw_service:
            stats["hnsw_service"] = self.hnsw_service.get_service_info()
        if self.worker_pool:
            stats["worker_pool"] = self.worker_pool.get_pool_stats()
        elif self.workers:
            stats["workers"] = [w.get_worker_stats() for w in self.workers]
        return stats

    def get_molecules(self, n: int = None):
        return self.coordination_service.scored_set.get_molecules(n) if self.coordination_service else []

    def get_best_molecules(self, n: int = None):
        return self.coordination_service.scored_set.get_best_molecules(n) if self.coordination_service else []

    def shutdown(self, **kwargs):
        self.is_running = False
        if self.worker_pool:
            self.worker_pool.stop_all()
            self.worker_pool = None
        for w in self.workers:
            w.stop()
        self.workers.clear()
        if self.coordination_service and not self.coordination_service.should_terminate:
            self.coordination_service.shutdown("Traverser shutdown")
        if self.hnsw_service:
            self.hnsw_service.shutdown()


def create_local_traverser(hnsw, scoring_fn, **kwargs) -> RADTraverser:
    svc_kw = {k: kwargs[k] for k in ("database_path", "max_queue_size", "response_timeout",
                                     "health_check_interval") if k in kwargs}
    return RADTraverser(hnsw_service=create_local_hnsw_service(hnsw, **svc_kw), scoring_fn=scoring_fn,
                        deployment_mode="local", **kwargs)


def create_distributed_traverser(hnsw, scoring_fn, redis_host: str = None, redis_port: int = 6379,
                                 redis_password: Optional[str] = None, **kwargs) -> RADTraverser:
    svc_kw = {k: kwargs[k] for k in ("database_path",) if k in kwargs}
    return RADTraverser(hnsw_service=create_local_hnsw_service(hnsw, **svc_kw), scoring_fn=scoring_fn,
                        deployment_mode="distributed", redis_host=redis_host, redis_port=redis_port,
                        redis_password=redis_password, **kwargs)


def create_remote_traverser(hnsw_service_url: str, scoring_fn, **kwargs) -> RADTraverser:
    from .hnsw_service import create_remote_hnsw_service
    return RADTraverser(hnsw_service=create_remote_hnsw_service(hnsw_service_url, **kwargs),
                        scoring_fn=scoring_fn, deployment_mode="distributed", **kwargs)


class TanimotoRADTraverser:
    """RAD traversals scored by Tanimoto distance to query fingerprints, on the GPU.

    Equivalent to one RADTraverser per query with
    ``scoring_fn = lambda smiles: float32(1) - float32(|q & fp|) / float32(|q | fp|)``, run with the
    idealised sequential semantics; every traversal is one wavefront of trav_kernel and keeps
    its queue / visited / scored state in HBM.  `get_molecules(q)` returns
    (node_id, score, smiles) in traversal order like rad/scored.py:63-85; smiles come from the
    optional `smiles_of(keys) -> list[str]` callable (e.g. a LocalHNSWService SQLite join).
    """

    def __init__(self, index, queries: np.ndarray, smiles_of: Optional[Callable] = None, log_pops: bool = False):
        from .device import DeviceTraversal  # noqa: F401  (fails loudly without the HIP library)
        self.index = index
        self._dev = index.device_index() if hasattr(index, "device_index") else index
        self.queries = np.ascontiguousarray(queries, np.uint8)
        if self.queries.ndim == 1:
            self.queries = self.queries.reshape(1, -1)
        self.smiles_of = smiles_of
        self.log_pops = log_pops
        self._trav = None
        self._n_to_score = None

    def prime(self, **kwargs):
        """Priming (scoring the top-level nodes) happens on the device at the start of
        traverse(); kept for API symmetry with RADTraverser."""
        return None

    def traverse(self, n_workers: int = 1, timeout: Optional[float] = None,
                 n_to_score: Optional[int] = None, round_pops: int = 0, **kwargs):
        from .device import DeviceTraversal
        if n_to_score is None:
            raise ValueError("TanimotoRADTraverser needs n_to_score")
        # a run to completion keeps the heavy state per resident row of the kernel (a batch of any size then needs the rows'
        # 40 GB + 0.8 MB per query at n_to_score = 100k); rounds / timeouts park traversals, which needs state per traversal
        to_completion = timeout is None and not round_pops
        if self._trav is None or self._n_to_score != n_to_score or (self._trav.slots != 0 and not to_completion):
            if self._trav is not None:
                self._trav.close()
            self._trav = DeviceTraversal(self._dev, self.queries, n_to_score, log_pops=self.log_pops, slots=to_completion)
            self._n_to_score = n_to_score
        t0 = time.time()
        if timeout is None and not round_pops:
            self._trav.run(0)
        else:
            step = round_pops or 4096
            while self._trav.run(step) > 0:
                if timeout is not None and time.time() - t0 >= timeout:
                    break

    def results(self, q: int = 0):
        """(slots, and_counts, or_counts) of query q in traversal order."""
        return self._trav.results(q)

    def get_molecules(self, n: int = None, q: int = 0):
        from .device import distance_f32
        s, a, o = self._trav.results(q)
        if n is not None:
            s, a, o = s[:n], a[:n], o[:n]
        d = distance_f32(a, o)
        smiles = self._smiles(s)
        return [(int(i), float(x), smi) for i, x, smi in zip(s, d, smiles)]

    def get_best_molecules(self, n: int = None, q: int = 0):
        mols = sorted(self.get_molecules(q=q), key=lambda x: x[1])
        return mols if n is None else mols[:n]

    def _smiles(self, slots):
        if self.smiles_of is None:
            return [""] * len(slots)
        keys = self.index.keys_of(slots) if hasattr(self.index, "keys_of") else slots
        return list(self.smiles_of(keys))

    def get_traversal_stats(self) -> Dict[str, Any]:
        st = self._trav.stats()
        ms, launches = self._trav.kernel_time()
        return {"n_scored": st.n_scored.tolist(), "n_pops": st.n_pops.tolist(), "status": st.status.tolist(),
                "kernel_ms": ms, "launches": launches, "state_bytes": self._trav.state_bytes()}

    def shutdown(self, **kwargs):
        if self._trav is not None:
            self._trav.close()
            self._trav = None

    close = shutdown
