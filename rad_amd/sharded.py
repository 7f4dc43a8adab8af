"""Corpus-sharded RAD traversal: one shard (rows + shard-local layered graph) per GPU, one
process per GPU, a per-round RCCL all-gather of frontier candidate scores and scored counts.

Semantics ("federated best-first", this build's definition — the reference has no multi-GPU
path).  A query's traversal runs on every shard against that shard's graph.  A global budget
`n_to_score` is split over the shards round by round:

    round:    every shard advances each of its traversals until it has scored its current
              target (or its queue is empty), at most `round_pops` expansions if given
    exchange: all-gather {scored count, best frontier key} of every traversal (2 x u64 each)
    every rank computes the same new targets with `allocate_targets`:
              remaining = n_to_score - sum(scored); shards whose queue is empty (or whose
              local capacity is used up) get nothing; the others split `remaining` evenly,
              the remainder going to the shards with the best (smallest) frontier key first,
              ties by rank
    stop:     remaining <= 0, or no shard can take more, or a round made no progress

With one shard it degenerates to the single-GPU traversal (target = n_to_score).  The result
of a query is the union of the shards' scored lists (slots are shard-local; global slot =
shard row offset + slot).
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

KEY_EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)


def allocate_targets(scored: np.ndarray, frontier: np.ndarray, n_to_score: int,
                     local_cap: int) -> Tuple[np.ndarray, np.ndarray]:
    """scored, frontier: [world, nq].  Returns (targets [world, nq] u64, done [nq] bool).
    Pure integer arithmetic on the all-gathered values: identical on every rank."""
    scored = np.asarray(scored).astype(np.int64)
    frontier = np.asarray(frontier, dtype=np.uint64)
    world, nq = scored.shape
    remaining = n_to_score - scored.sum(0)
    if (remaining <= 0).all():      # every budget is spent (the common last round): nothing to split
        return scored.astype(np.uint64), np.ones(nq, bool)
    live = (frontier != KEY_EMPTY) & (scored < local_cap)
    n_live = live.sum(0)
    done = (remaining <= 0) | (n_live == 0)
    # position of every shard in (frontier key, rank) order among the LIVE shards
    sort_key = np.where(live, frontier, KEY_EMPTY)
    order = np.argsort(sort_key, axis=0, kind="stable")
    pos = np.empty_like(order)
    np.put_along_axis(pos, order, np.repeat(np.arange(world)[:, None], nq, axis=1), axis=0)
    rem = np.maximum(remaining, 0)
    nl = np.maximum(n_live, 1)
    share = rem // nl + (pos < (rem % nl)[None, :]).astype(np.int64)
    share = np.where(live & ~done[None, :], share, 0)
    targets = np.minimum(scored + share, local_cap)
    return targets.astype(np.uint64), done


class ShardedTraversal:
    """Drives one shard's traversal object through the federated rounds.

    `local` needs nq, set_targets(u64[nq]), run(max_pops) and frontier() -> (keys, scored):
    rad_amd.device.DeviceTraversal on a GPU.  `allgather(u64[k]) -> u64[world, k]` is the
    exchange step: rad_amd.device.RcclComm.allgather_u64 on GPUs (gloo in the CPU tests).
    `local_cap` is the n_to_score the local traversal state was sized for.
    """

    def __init__(self, local, allgather: Callable[[np.ndarray], np.ndarray], rank: int, world: int,
                 n_to_score: int, local_cap: int, round_pops: int = 0):
        self.local = local
        self.allgather = allgather
        self.rank, self.world = int(rank), int(world)
        self.n_to_score = int(n_to_score)
        self.local_cap = int(local_cap)
        self.round_pops = int(round_pops)
        self.rounds = 0
        self.exchanged_bytes = 0
        self.scored_all = self.frontier_all = self.done = None

    def run(self, max_rounds: int = 10000):
        nq = self.local.nq
        first = min(-(-self.n_to_score // self.world), self.local_cap)   # even split, rounded up
        self.local.set_targets(np.full(nq, first, np.uint64))
        prev_total = -1
        while True:
            self.local.run(self.round_pops)
            keys, scored = self.local.frontier()
            mine = np.concatenate([np.asarray(scored, np.uint64), np.asarray(keys, np.uint64)])
            allv = np.asarray(self.allgather(mine), np.uint64).reshape(self.world, 2 * nq)
            self.exchanged_bytes += allv.nbytes
            self.rounds += 1
            sc, fr = allv[:, :nq], allv[:, nq:]
            targets, done = allocate_targets(sc, fr, self.n_to_score, self.local_cap)
            total = int(sc.astype(np.int64).sum())
            self.scored_all, self.frontier_all, self.done = sc, fr, done
            if bool(done.all()) or self.rounds >= max_rounds:
                return sc, fr
            # "no progress anywhere" is a global fact (total is all-gathered): every rank stops together
            if total == prev_total and not self.round_pops:
                return sc, fr
            prev_total = total
            self.local.set_targets(targets[self.rank])
