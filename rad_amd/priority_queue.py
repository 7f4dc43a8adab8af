"""Priority queue of RAD's traversal state (reference: rad/priority_queue.py).

`PriorityQueue` is the reference's ABC (rad/priority_queue.py:4-11).  `InProcessPQ` is an
in-process implementation with the exact ordering of the reference's Redis ZSET
(rad/priority_queue.py:22-42): ascending score (double), ties broken by the bytewise order of
the member string "{node_id}:{level}" (so "10:0" < "9:0"), and re-inserting an existing
(node_id, level) overwrites its score (ZADD).  It removes the Redis round trip per pop/insert
(SURVEY.md §8f N1); it is host bookkeeping for user-supplied scores — the Tanimoto-scored
traversal keeps its queue on the GPU (rad_amd/csrc/traverse.hip).
"""
from __future__ import annotations

import heapq
import threading
from abc import ABC, abstractmethod
from typing import Optional, Tuple


class PriorityQueue(ABC):
    @abstractmethod
    def pop(self) -> Tuple[int, int, float]:
        pass

    @abstractmethod
    def insert(self, node_id: int, level: int, score: float):
        pass


class InProcessPQ(PriorityQueue):
    def __init__(self, queue_name: str = "pq", **kwargs):
        self.queue_name = queue_name
        self._heap = []          # (score, member_bytes, version)
        self._live = {}          # member_bytes -> (score, version)
        self._version = 0
        self._lock = threading.Lock()

    def pop(self) -> Optional[Tuple[int, int, float]]:
        with self._lock:
            while self._heap:
                score, member, version = heapq.heappop(self._heap)
                cur = self._live.get(member)
                if cur is None or cur[1] != version:
                    continue  # stale entry of an overwritten / removed member
                del self._live[member]
                node_id, level = map(int, member.decode("ascii").split(":"))
                return node_id, level, float(score)
            return None

    def insert(self, node_id, level, score):
        member = f"{int(node_id)}:{int(level)}".encode("ascii")
        score = float(score)
        with self._lock:
            self._version += 1
            self._live[member] = (score, self._version)
            heapq.heappush(self._heap, (score, member, self._version))

    def __len__(self):
        with self._lock:
            return len(self._live)
