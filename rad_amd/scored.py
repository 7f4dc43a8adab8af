"""Scored set of RAD's traversal state (reference: rad/scored.py).

`ScoredSet` is the reference's ABC (rad/scored.py:3-24); `InProcessScoredSet` keeps insertion
order, first write wins (rad/scored.py:37-47), scores round-trip through their decimal string
as they do through Redis (rad/scored.py:49-61 stores str(score), returns float(...)).
"""
from __future__ import annotations

import threading
from abc import ABC, abstractmethod


class ScoredSet(ABC):
    @abstractmethod
    def getScore(self, node_id: int) -> float:
        pass

    @abstractmethod
    def insert(self, node_id: int, score: float, smiles: str = ""):
        pass

    @abstractmethod
    def get_molecules(self, n: int = None):
        """(node_id, score, smiles) tuples, oldest first; n limits the count."""
        pass

    @abstractmethod
    def get_best_molecules(self, n: int = None):
        """The same tuples ordered by ascending score (lower docking score = better)."""
        pass

    @abstractmethod
    def __len__(self):
        pass


class InProcessScoredSet(ScoredSet):
    def __init__(self, scored_name: str = "scored", **kwargs):
        self.scored_name = scored_name
        self._order = []
        self._score = {}
        self._smiles = {}
        self._lock = threading.Lock()

    def getScore(self, node_id):
        return self._score.get(int(node_id))

    def insert(self, node_id, score, smiles=""):
        node_id = int(node_id)
        with self._lock:
            if node_id not in self._score:
                self._score[node_id] = float(str(score))
                self._smiles[node_id] = str(smiles)
                self._order.append(node_id)

    def get_molecules(self, n=None):
        ids = self._order[:] if n is None else self._order[:n]
        return [(i, self._score[i], self._smiles.get(i) or "") for i in ids]

    def get_best_molecules(self, n=None):
        ordered = sorted(self.get_molecules(), key=lambda x: x[1])  # stable: ties keep insertion order
        return ordered if n is None else ordered[:n]

    def save(self, path):
        with open(path, "w") as f:
            for key, score in self:
                f.write(f"{key} {score}\n")

    def __iter__(self):
        for key in list(self._order):
            yield (key, self._score[key])

    def __len__(self):
        return len(self._order)
