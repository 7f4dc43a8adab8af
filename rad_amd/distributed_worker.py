"""Workers of a RAD traversal (reference: rad/distributed_worker.py).

`DistributedWorker._process_work_item` restates rad/distributed_worker.py:272-333: score (with
the user's scoring_fn) every pre-fetched neighbour that has no score yet, skip neighbours whose
scoring raises, then hand the results to CoordinationService.submit_work_results.  Workers are
threads of this process, as in the reference (rad/distributed_worker.py:172-177); the 1 s idle
sleep of the reference's loop (:253) is replaced by a short back-off.  Heartbeat / signal
plumbing is out of scope.
"""
from __future__ import annotations

import logging
import threading
import time
import uuid
from typing import Any, Callable, Dict, List, Optional

from .coordination_service import CoordinationService, WorkItem

logger = logging.getLogger(__name__)


class DistributedWorker:
    def __init__(self, worker_id: Optional[str] = None,
                 coordination_service: Optional[CoordinationService] = None,
                 scoring_fn: Optional[Callable] = None, worker_type: str = "default",
                 capabilities: Optional[Dict] = None, heartbeat_interval: float = 10.0,
                 work_timeout: float = 30.0, max_retries: int = 3, idle_sleep: float = 0.002, **kwargs):
        self.worker_id = worker_id or f"worker_{uuid.uuid4().hex[:8]}"
        self.worker_type = worker_type
        self.capabilities = capabilities or {}
        self.heartbeat_interval = heartbeat_interval
        self.work_timeout = work_timeout
        self.max_retries = max_retries
        self.idle_sleep = idle_sleep
        self.coordination_service = coordination_service
        self.scoring_fn = scoring_fn
        self.is_running = False
        self.should_stop = False
        self.started_at = None
        self.last_work_at = None
        self.work_completed = 0
        self.work_failed = 0
        self.total_score_time = 0.0
        self.errors: List[Dict[str, Any]] = []
        self.work_thread = None
        self.worker_lock = threading.Lock()

    def connect_services(self, coordination_service: Optional[CoordinationService] = None, **kwargs) -> bool:
        if coordination_service:
            self.coordination_service = coordination_service
        return self.coordination_service is not None

    def start(self, register_worker: bool = True) -> bool:
        if self.is_running:
            return False
        if self.coordination_service is None or self.scoring_fn is None:
            logger.error("%s cannot start without coordination service and scoring_fn", self.worker_id)
            return False
        if register_worker:
            self.coordination_service.register_worker(self.worker_id, self.worker_type, self.capabilities)
        self.is_running = True
        self.should_stop = False
        self.started_at = time.time()
        self.work_thread = threading.Thread(target=self._work_loop, daemon=True, name=f"Work-{self.worker_id}")
        self.work_thread.start()
        return True

    def stop(self, timeout: float = 10.0) -> None:
        self.should_stop = True
        self.is_running = False
        if self.work_thread and self.work_thread.is_alive() and self.work_thread is not threading.current_thread():
            self.work_thread.join(timeout=timeout)

    def get_worker_stats(self) -> Dict[str, Any]:
        runtime = time.time() - self.started_at if self.started_at else 0
        with self.worker_lock:
            return {"worker_id": self.worker_id, "worker_type": self.worker_type,
                    "is_running": self.is_running, "runtime_seconds": runtime,
                    "work_completed": self.work_completed, "work_failed": self.work_failed,
                    "total_score_time": self.total_score_time,
                    "avg_score_time": self.total_score_time / max(self.work_completed, 1),
                    "error_count": len(self.errors), "last_work_at": self.last_work_at}

    def _work_loop(self):
        cs = self.coordination_service
        while self.is_running and not self.should_stop:
            try:
                done, _ = cs.check_termination()
                if done:
                    break
                item = cs.request_work(self.worker_id)
                if item is None:
                    time.sleep(self.idle_sleep)
                    continue
                ok = self._process_work_item(item)
                with self.worker_lock:
                    if ok:
                        self.work_completed += 1
                        self.last_work_at = time.time()
                    else:
                        self.work_failed += 1
            except Exception as e:  # pragma: no cover - defensive
                self._record_error(f"Work loop error: {e}")
                time.sleep(self.idle_sleep)

    def _process_work_item(self, work_item: WorkItem) -> bool:
        try:
            neighbors = work_item.neighbors or []   # empty row: still descends (stated deviation)
            t0 = time.time()
            new_scores = {}
            scored = self.coordination_service.scored_set
            for i in range(0, len(neighbors), 2):
                nid, smiles = neighbors[i], neighbors[i + 1]
                try:
                    if scored.getScore(nid) is None:
                        new_scores[nid] = (self.scoring_fn(smiles), smiles)
                except Exception as e:
                    logger.warning("scoring_fn failed for node %s (%r): %s — neighbour skipped", nid, smiles, e)
                    continue
            dt = time.time() - t0
            ok = self.coordination_service.submit_work_results(self.worker_id, work_item, neighbors, new_scores)
            if ok:
                with self.worker_lock:
                    self.total_score_time += dt
            return ok
        except Exception as e:
            self._record_error(f"Work processing error: {e}")
            return False

    def _record_error(self, msg: str):
        with self.worker_lock:
            self.errors.append({"timestamp": time.time(), "message": msg})
            del self.errors[:-100]


class WorkerPool:
    def __init__(self, n_workers: int, worker_config: Dict[str, Any]):
        self.n_workers = n_workers
        self.worker_config = dict(worker_config)
        self.workers: List[DistributedWorker] = []
        self.is_running = False

    def start_all(self) -> bool:
        prefix = self.worker_config.pop("worker_id_prefix", "worker")
        for i in range(self.n_workers):
            w = DistributedWorker(worker_id=f"{prefix}_{i}", **self.worker_config)
            if not w.start():
                self.stop_all()
                return False
            self.workers.append(w)
        self.is_running = True
        return True

    def stop_all(self) -> None:
        for w in self.workers:
            w.should_stop = True
        for w in self.workers:
            w.stop()
        self.workers.clear()
        self.is_running = False

    def get_pool_stats(self) -> Dict[str, Any]:
        ws = [w.get_worker_stats() for w in self.workers]
        return {"n_workers": self.n_workers, "is_running": self.is_running,
                "active_workers": sum(1 for s in ws if s["is_running"]),
                "total_work_completed": sum(s["work_completed"] for s in ws),
                "total_work_failed": sum(s["work_failed"] for s in ws),
                "total_errors": sum(s["error_count"] for s in ws), "worker_stats": ws}


def create_worker_pool(n_workers: int, **worker_config) -> WorkerPool:
    return WorkerPool(n_workers, worker_config)
