"""Work hand-out and result merge of a RAD traversal (reference: rad/coordination_service.py).

Same public surface as the reference's CoordinationService for the hot path —
`request_work` (rad/coordination_service.py:290-347) and `submit_work_results` (:349-413) —
over injectable in-process state backends instead of hard-wired Redis ones (:158-169), which
removes ~4 Redis round trips per neighbour.  Worker registration / heartbeats are kept as plain
bookkeeping; dead-worker reassignment threads are out of scope.

Stated deviation: a node whose adjacency row is empty still descends one level (the
reference's worker fails such an item and the node is re-queued after 120 s, forever:
rad/distributed_worker.py:286-288, rad/coordination_service.py:559-574).
"""
from __future__ import annotations

import json
import logging
import threading
import time
import uuid
from typing import Any, Dict, List, Optional, Tuple

from .priority_queue import InProcessPQ, PriorityQueue
from .scored import InProcessScoredSet, ScoredSet
from .visited import InProcessVisited, VisitedSet

logger = logging.getLogger(__name__)


class WorkItem:
    """One (node, level) to expand, with its pre-fetched neighbours
    (reference: rad/coordination_service.py:34-74)."""

    def __init__(self, node_id: int, level: int, score: float,
                 request_id: Optional[str] = None, neighbors: Optional[List] = None):
        self.node_id = node_id
        self.level = level
        self.score = score
        self.request_id = request_id or str(uuid.uuid4())
        self.neighbors = neighbors
        self.created_at = time.time()
        self.assigned_at = None
        self.assigned_to = None

    def to_dict(self) -> Dict[str, Any]:
        return {"node_id": self.node_id, "level": self.level, "score": self.score,
                "request_id": self.request_id, "neighbors": self.neighbors,
                "created_at": self.created_at, "assigned_at": self.assigned_at,
                "assigned_to": self.assigned_to}

    @classmethod
    def from_dict(cls, data: Dict[str, Any]) -> "WorkItem":
        item = cls(data["node_id"], data["level"], data["score"], request_id=data["request_id"],
                   neighbors=data.get("neighbors"))
        item.created_at = data["created_at"]
        item.assigned_at = data.get("assigned_at")
        item.assigned_to = data.get("assigned_to")
        return item


class WorkerInfo:
    def __init__(self, worker_id: str, worker_type: str = "default", capabilities: Optional[Dict] = None):
        self.worker_id = worker_id
        self.worker_type = worker_type
        self.capabilities = capabilities or {}
        self.registered_at = time.time()
        self.last_heartbeat = time.time()
        self.assigned_work = set()
        self.completed_work = 0
        self.error_count = 0
        self.status = "active"

    def to_dict(self) -> Dict[str, Any]:
        return {"worker_id": self.worker_id, "worker_type": self.worker_type,
                "capabilities": self.capabilities, "registered_at": self.registered_at,
                "last_heartbeat": self.last_heartbeat, "assigned_work": list(self.assigned_work),
                "completed_work": self.completed_work, "error_count": self.error_count,
                "status": self.status}


class CoordinationService:
    def __init__(self, redis_client=None, hnsw_service=None, namespace: str = "rad_coordination",
                 worker_timeout: float = 60.0, heartbeat_interval: float = 10.0,
                 priority_queue: Optional[PriorityQueue] = None,
                 visited_set: Optional[VisitedSet] = None,
                 scored_set: Optional[ScoredSet] = None, **kwargs):
        """`redis_client` is accepted (and ignored) so reference call sites keep working."""
        if hnsw_service is None:
            raise ValueError("CoordinationService requires an HNSW service")
        self.redis = redis_client
        self.hnsw_service = hnsw_service
        self.namespace = namespace
        self.worker_timeout = worker_timeout
        self.heartbeat_interval = heartbeat_interval
        self.coordination_id = str(uuid.uuid4())
        self.started_at = time.time()
        self.is_running = False
        self.total_neighbor_queries = 0
        self.total_neighbor_time = 0.0
        self.workers: Dict[str, WorkerInfo] = {}
        self.worker_lock = threading.Lock()
        self.termination_conditions: Dict[str, Any] = {}
        self.should_terminate = False
        self.termination_reason = None
        self.priority_queue = InProcessPQ(queue_name=f"{namespace}:priority_queue") if priority_queue is None else priority_queue
        self.visited_set = InProcessVisited(visited_name=f"{namespace}:visited") if visited_set is None else visited_set
        self.scored_set = InProcessScoredSet(scored_name=f"{namespace}:scored") if scored_set is None else scored_set
        self._assignments: Dict[str, WorkItem] = {}
        # One merge at a time: the reference gets per-call atomicity from Redis Lua scripts;
        # in-process the whole result merge is one critical section.
        self._merge_lock = threading.Lock()
        self._inflight = 0   # pops whose assignment is not registered yet (guards termination)
        self.final_stats = None

    # -- lifecycle ----------------------------------------------------------
    def start(self, termination_conditions: Dict[str, Any]) -> None:
        if self.is_running:
            raise RuntimeError("Coordination service is already running")
        self.termination_conditions = dict(termination_conditions)
        self.should_terminate = False
        self.termination_reason = None
        self.started_at = time.time()
        self.is_running = True

    def shutdown(self, reason: str = "Manual shutdown") -> None:
        self.should_terminate = True
        self.termination_reason = reason
        self.is_running = False
        self.final_stats = json.dumps(self.get_coordination_stats(), default=str)

    # -- workers --------------------------------------------------------------
    def register_worker(self, worker_id: str, worker_type: str = "default",
                        capabilities: Optional[Dict] = None) -> bool:
        with self.worker_lock:
            if worker_id in self.workers:
                logger.warning("Worker %s already registered", worker_id)
                return False
            self.workers[worker_id] = WorkerInfo(worker_id, worker_type, capabilities)
            return True

    def worker_heartbeat(self, worker_id: str) -> bool:
        with self.worker_lock:
            w = self.workers.get(worker_id)
            if w is None:
                return False
            w.last_heartbeat = time.time()
            w.status = "active"
            return True

    # -- hot path ---------------------------------------------------------------
    def request_work(self, worker_id: str) -> Optional[WorkItem]:
        """Pop the best (node, level) and pre-fetch its neighbours
        (rad/coordination_service.py:290-347)."""
        if self.should_terminate:
            return None
        if worker_id not in self.workers:
            logger.warning("Work request from unregistered worker: %s", worker_id)
            return None
        with self.worker_lock:
            self._inflight += 1
        work = self.priority_queue.pop()
        if work is None:
            with self.worker_lock:
                self._inflight -= 1
            return None
        node_id, level, score = work
        try:
            t0 = time.time()
            neighbors = self.hnsw_service.get_neighbors(node_id, level)
            self.total_neighbor_queries += 1
            self.total_neighbor_time += time.time() - t0
        except Exception as e:  # put the work back, as the reference does (:324-328)
            logger.error("Failed to get neighbors for node %s at level %s: %s", node_id, level, e)
            self.priority_queue.insert(node_id, level, score)
            with self.worker_lock:
                self._inflight -= 1
            return None
        item = WorkItem(node_id, level, score, neighbors=neighbors)
        item.assigned_at = time.time()
        item.assigned_to = worker_id
        with self.worker_lock:
            self.workers[worker_id].assigned_work.add(item.request_id)
            self._assignments[item.request_id] = item
            self._inflight -= 1
        return item

    def submit_work_results(self, worker_id: str, work_item: WorkItem, neighbors: List,
                            new_scores: Dict[int, tuple]) -> bool:
        """Merge one expansion (rad/coordination_service.py:349-413): per neighbour the
        visited gate on (id, level), scored insert keyed by id (first write wins), queue insert
        on the SAME level; then the expanded node descends one level with its own score."""
        if worker_id not in self.workers:
            logger.warning("Results from unregistered worker: %s", worker_id)
            return False
        try:
            with self._merge_lock:
                level = work_item.level
                for i in range(0, len(neighbors), 2):
                    nid, smiles = neighbors[i], neighbors[i + 1]
                    if self.visited_set.checkAndInsert(nid, level):
                        continue
                    if nid in new_scores:
                        score, smi = new_scores[nid]
                        self.scored_set.insert(nid, score, smi)
                    else:
                        score = self.scored_set.getScore(nid)
                        if score is None:
                            logger.warning("No score provided for neighbor %s", nid)
                            continue
                    self.priority_queue.insert(nid, level, score)
                if level > 0 and not self.visited_set.checkAndInsert(work_item.node_id, level - 1):
                    self.priority_queue.insert(work_item.node_id, level - 1, work_item.score)
            with self.worker_lock:
                w = self.workers[worker_id]
                w.assigned_work.discard(work_item.request_id)
                w.completed_work += 1
                self._assignments.pop(work_item.request_id, None)
            return True
        except Exception as e:
            logger.error("Error processing results from worker %s: %s", worker_id, e)
            with self.worker_lock:
                self.workers[worker_id].error_count += 1
            return False

    # -- termination / stats ------------------------------------------------------
    def check_termination(self) -> Tuple[bool, Optional[str]]:
        """rad/coordination_service.py:415-457, without the pop-and-put-back probe (it races
        with workers in the reference; here the queue length is read directly)."""
        if self.should_terminate:
            return True, self.termination_reason
        c = self.termination_conditions
        if "timeout" in c:
            runtime = time.time() - self.started_at
            if runtime >= c["timeout"]:
                return True, f"Timeout reached ({runtime:.1f}s >= {c['timeout']}s)"
        if "n_to_score" in c:
            k = len(self.scored_set)
            if k >= c["n_to_score"]:
                return True, f"Target molecules scored ({k} >= {c['n_to_score']})"
        if self._pending() == 0:
            with self.worker_lock:
                active = sum(len(w.assigned_work) for w in self.workers.values()) + self._inflight
            if active == 0 and self._pending() == 0:
                return True, "No more work available and no active assignments"
        return False, None

    def _pending(self) -> int:
        try:
            return len(self.priority_queue)
        except TypeError:
            return -1

    def get_coordination_stats(self) -> Dict[str, Any]:
        with self.worker_lock:
            ws = list(self.workers.values())
        return {
            "coordination_id": self.coordination_id,
            "runtime_seconds": time.time() - self.started_at,
            "is_running": self.is_running,
            "should_terminate": self.should_terminate,
            "termination_reason": self.termination_reason,
            "scored_molecules": len(self.scored_set),
            "pending_work": self._pending(),
            "workers": {"total_workers": len(ws),
                        "active_workers": sum(1 for w in ws if w.status == "active"),
                        "total_completed_work": sum(w.completed_work for w in ws),
                        "total_errors": sum(w.error_count for w in ws)},
            "hnsw_proxy": {"total_neighbor_queries": self.total_neighbor_queries,
                           "total_neighbor_time": self.total_neighbor_time,
                           "avg_neighbor_time": self.total_neighbor_time / max(self.total_neighbor_queries, 1)},
            "termination_conditions": self.termination_conditions,
        }


def create_coordination_service(redis_client=None, hnsw_service=None, **kwargs) -> CoordinationService:
    return CoordinationService(redis_client, hnsw_service, **kwargs)
