"""Visited set of RAD's traversal state (reference: rad/visited.py).

`VisitedSet` is the reference's ABC (rad/visited.py:3-6); `InProcessVisited` is an atomic
test-and-set on the key (node_id, level) that returns True when the key was ALREADY present
(rad/visited.py:17-29) — the same node on two levels is two keys.
"""
from __future__ import annotations

import threading
from abc import ABC, abstractmethod


class VisitedSet(ABC):
    @abstractmethod
    def checkAndInsert(self, node_id: int, level: int) -> bool:
        pass


class InProcessVisited(VisitedSet):
    def __init__(self, visited_name: str = "visited", **kwargs):
        self.visited_name = visited_name
        self._set = set()
        self._lock = threading.Lock()

    def checkAndInsert(self, node_id, level) -> bool:
        key = (int(node_id), int(level))
        with self._lock:
            if key in self._set:
                return True
            self._set.add(key)
            return False

    def __len__(self):
        return len(self._set)
