"""HNSW service boundary (reference: rad/hnsw_service.py).

`HNSWService` is the reference's ABC (rad/hnsw_service.py:26-78).  `LocalHNSWService` serves
an index object (rad_amd.Index, or any object with the duck-typed usearch surface of
SURVEY.md §8 B1) to the traverser.  The reference forks a server process and pays one
multiprocessing.Queue round trip with a fresh uuid4 per call (rad/hnsw_service.py:129-134,
311-359); here the adjacency read is a direct, thread-safe call into the index (librad_hip's
host mirror of the graph), and the SQLite key -> SMILES join of
`_transform_to_smiles_format` (rad/hnsw_service.py:256-283) runs on a per-thread connection.
Return format, SMILES "" fill, error behaviour (RuntimeError("HNSW request failed: ..."),
RuntimeError once shut down) and the info dictionaries follow the reference.

`RemoteHNSWService` (HTTP transport, rad/hnsw_service.py:455-758) is out of scope.
"""
from __future__ import annotations

import logging
import os
import sqlite3
import threading
import time
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional, Sequence, Tuple

logger = logging.getLogger(__name__)


class HNSWService(ABC):
    @abstractmethod
    def get_neighbors(self, node_id: int, level: int) -> List[int]:
        """[neighbor_id, smiles, neighbor_id, smiles, ...] of a node on one level."""

    @abstractmethod
    def get_top_level_nodes(self) -> List[int]:
        """[node_id, smiles, ...] of the nodes on the top level."""

    @abstractmethod
    def is_healthy(self) -> bool:
        pass

    @abstractmethod
    def shutdown(self) -> None:
        pass

    @abstractmethod
    def get_service_info(self) -> Dict[str, Any]:
        pass

    @abstractmethod
    def get_hnsw_info(self) -> Dict[str, Any]:
        pass


class LocalHNSWService(HNSWService):
    def __init__(self, hnsw, database_path: Optional[str] = None, max_queue_size: int = 1000,
                 response_timeout: float = 30.0, health_check_interval: float = 5.0, **kwargs):
        self.hnsw = hnsw
        self.database_path = database_path
        self.max_queue_size = max_queue_size
        self.response_timeout = response_timeout
        self.health_check_interval = health_check_interval
        self.is_running = True
        self.start_time = time.time()
        self.request_count = 0
        self.error_count = 0
        self._count_lock = threading.Lock()
        self._tls = threading.local()
        self._db_ok = False
        if database_path:
            try:
                con = self._db()
                n = con.execute("SELECT COUNT(*) FROM nodes").fetchone()[0]
                logger.info("Database connected with %s nodes", n)
                self._db_ok = True
            except Exception as e:  # same tolerance as rad/hnsw_service.py:165-167
                logger.error("Database initialization failed: %s", e)
                self._db_ok = False

    # -- SMILES join -------------------------------------------------------
    def _db(self):
        con = getattr(self._tls, "con", None)
        if con is None:
            con = sqlite3.connect(self.database_path)
            self._tls.con = con
        return con

    def _get_smiles_batch(self, node_keys: Sequence[int]) -> Dict[int, str]:
        if not self._db_ok or not node_keys:
            return {}
        out: Dict[int, str] = {}
        try:
            con = self._db()
            keys = [int(k) for k in node_keys]
            for i in range(0, len(keys), 900):  # SQLite's default variable limit is 999
                part = keys[i:i + 900]
                q = f"SELECT node_key, smi FROM nodes WHERE node_key IN ({','.join('?' * len(part))})"
                out.update({int(k): s for k, s in con.execute(q, part)})
            missing = set(keys) - set(out)
            if missing:
                logger.warning("Missing SMILES for node keys: %s", missing)
        except Exception as e:
            logger.error("Error fetching SMILES: %s", e)
            return {}
        return out

    def _transform_to_smiles_format(self, hnsw_data: Sequence[int]) -> List:
        """[node_id, node_key, ...] -> [node_id, smiles, ...]; missing SMILES -> ""."""
        if len(hnsw_data) == 0:
            return []
        ids = [int(hnsw_data[i]) for i in range(0, len(hnsw_data), 2)]
        keys = [int(hnsw_data[i + 1]) for i in range(0, len(hnsw_data), 2)]
        smiles = self._get_smiles_batch(keys)
        out: List = []
        for i, k in zip(ids, keys):
            out.extend([i, smiles.get(k, "")])
        return out

    # -- requests -------------------------------------------------------------
    def _request(self, fn, *args):
        if not self.is_running:
            raise RuntimeError("HNSW service is not running")
        with self._count_lock:
            self.request_count += 1
        try:
            return fn(*args)
        except Exception as e:
            with self._count_lock:
                self.error_count += 1
            raise RuntimeError(f"HNSW request failed: {e}") from e

    def get_neighbors(self, node_id: int, level: int) -> List:
        return self._request(lambda: self._transform_to_smiles_format(
            [int(x) for x in self.hnsw.get_neighbors(node_id, level)]))

    def get_top_level_nodes(self) -> List:
        return self._request(lambda: self._transform_to_smiles_format(
            [int(x) for x in self.hnsw.get_top_level_nodes()]))

    def get_neighbors_many(self, pairs: Sequence[Tuple[int, int]]) -> List[List]:
        """Batched get_neighbors: one SMILES query for all rows."""
        def run():
            rows = [[int(x) for x in self.hnsw.get_neighbors(n, lv)] for n, lv in pairs]
            smiles = self._get_smiles_batch([r[i + 1] for r in rows for i in range(0, len(r), 2)])
            out = []
            for r in rows:
                o: List = []
                for i in range(0, len(r), 2):
                    o.extend([r[i], smiles.get(r[i + 1], "")])
                out.append(o)
            return out
        return self._request(run)

    def is_healthy(self) -> bool:
        return bool(self.is_running)

    def get_service_info(self) -> Dict[str, Any]:
        return {"service_type": "LocalHNSWService",
                "status": "running" if self.is_running else "stopped",
                "process_id": os.getpid(), "process_alive": bool(self.is_running),
                "uptime_seconds": time.time() - self.start_time,
                "request_count": self.request_count, "error_count": self.error_count,
                "error_rate": self.error_count / max(self.request_count, 1),
                "pending_requests": 0, "queue_sizes": {"request_queue": 0, "response_queue": 0}}

    def get_hnsw_info(self) -> Dict[str, Any]:
        try:
            h = self.hnsw
            return {"max_level": h.max_level, "size": len(h), "connectivity": h.connectivity,
                    "dtype": str(h.dtype), "ndim": h.ndim, "capacity": h.capacity,
                    "memory_usage": h.memory_usage, "multi": h.multi}
        except Exception as e:
            logger.error("Error getting HNSW info: %s", e)
            return {"max_level": 0, "size": -1, "connectivity": -1, "dtype": "unknown", "ndim": -1,
                    "capacity": -1, "memory_usage": -1, "multi": False, "error": str(e)}

    def shutdown(self) -> None:
        self.is_running = False


InProcessHNSWService = LocalHNSWService


class ServiceRegistry:
    """reference: rad/hnsw_service.py:761-812"""

    def __init__(self):
        self.services: Dict[str, HNSWService] = {}
        self.default_service: Optional[str] = None

    def register_service(self, name: str, service: HNSWService, is_default: bool = False) -> None:
        self.services[name] = service
        if is_default or self.default_service is None:
            self.default_service = name

    def get_service(self, name: Optional[str] = None) -> HNSWService:
        key = name or self.default_service
        if key not in self.services:
            raise ValueError(f"HNSW service not found: {key}")
        return self.services[key]

    def list_services(self) -> Dict[str, Dict[str, Any]]:
        return {n: s.get_service_info() for n, s in self.services.items()}

    def shutdown_all(self) -> None:
        for n, s in self.services.items():
            try:
                s.shutdown()
            except Exception as e:
                logger.error("Error shutting down service %s: %s", n, e)
        self.services.clear()
        self.default_service = None


service_registry = ServiceRegistry()


def create_local_hnsw_service(hnsw, **kwargs) -> LocalHNSWService:
    service = LocalHNSWService(hnsw, **kwargs)
    service_registry.register_service("local", service, is_default=True)
    return service


def create_remote_hnsw_service(base_url: str, *args, **kwargs):
    raise NotImplementedError(
        "RemoteHNSWService (HTTP transport of rad/hnsw_service.py:455-758) is outside the scope of "
        "rad_amd: serve a rad_amd.Index through the reference's own FastAPI server instead")
