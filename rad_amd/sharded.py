"""Multi-GPU traversal: one process per GPU.

`RowShardedTraversal` — the mode BASELINE.json's north_star and SURVEY.md §8(e) describe.  The
fingerprint corpus is partitioned by contiguous slot range, the layered graph is ONE graph over all
rows, the traversals of a batch are partitioned over the ranks.  Per frontier step:

    step      every rank advances each of ITS traversals to the point where a fingerprint would be
              read: apply the scores of the last step's candidates (scored insert + queue insert,
              rad/coordination_service.py:379-389), pop / expand (visited test-and-set,
              rad/visited.py:17-29) until a neighbour is not in the scored set
              (rad/distributed_worker.py:296-305); those slots are the frontier candidates
    exchange  all-gather of the candidates of all ranks (nq x W slots each, RADHIP_NO_SLOT padded)
    evaluate  every rank scores the candidates whose rows it owns against the asking traversal's query
    exchange  the packed (and | or << 16) scores return to the asking rank: a reduce-scatter of
              disjoint contributions (a candidate has one owner; everybody else adds 0)

The control flow per traversal is the single-GPU one (strict best-first), so scored lists and pop logs
are bit-identical to a single-GPU traversal of the same corpus and graph for any number of ranks:
tests/test_sharded.py (world 2, gloo, the oracle's stepper as the local engine) and
tests/test_gpu_sharded.py (two ranks' worth of kernels on one GPU).  On GPUs the whole loop runs in
the library (radhip_shard_run: kernels and RCCL collectives on one stream, device buffers);
`RowShardedTraversal` is the same loop with the exchange injected, for hosts that own the exchange.

`FederatedTraversal` (the round-1 mode, kept as a labelled alternative): shard-LOCAL graphs and a
global budget split round by round — cheaper to exchange (16 B per traversal and round) but NOT the
traversal of one global graph; its results cannot be compared with a single-GPU run.
"""
from __future__ import annotations

from typing import Callable, Tuple

import numpy as np

KEY_EMPTY = np.uint64(0xFFFFFFFFFFFFFFFF)
NO_SLOT = np.uint32(0xFFFFFFFF)


class RowShardedTraversal:
    """Drives one rank of the row-sharded traversal with an injected exchange.

    `local` provides  nq, width,
        step(scores_in u32[nq, W]) -> (requests u32[nq, W], live)      advance the local traversals
        evaluate(requests_all u32[world, nq, W]) -> u32[world, nq, W]  score the candidates this rank owns
    (rad_amd.device.DeviceShard on a GPU; the oracle's stepper in the CPU tests).
    `allgather(u32[...]) -> u32[world, ...]`, `reduce_scatter(u32[world, ...]) -> u32[...]` (sum over
    ranks of this rank's block) are the exchange: gloo in the CPU tests, rad_amd.rendezvous.TcpGroup when
    several ranks rehearse on one GPU."""

    def __init__(self, local, allgather: Callable, reduce_scatter: Callable, rank: int, world: int):
        self.local, self.allgather, self.reduce_scatter = local, allgather, reduce_scatter
        self.rank, self.world = int(rank), int(world)
        self.steps = 0
        self.exchanged_bytes = 0

    def run(self, max_steps: int = 0) -> int:
        nq, W = self.local.nq, self.local.width
        scores = np.zeros((nq, W), np.uint32)
        while True:
            req, live = self.local.step(scores)
            # the live count travels behind the candidates: every rank sees the same total and stops at the same step
            send = np.concatenate([np.ascontiguousarray(req, np.uint32).reshape(-1), np.array([live], np.uint32)])
            allv = np.asarray(self.allgather(send), np.uint32).reshape(self.world, nq * W + 1)
            self.steps += 1
            self.exchanged_bytes += allv.nbytes
            if int(allv[:, -1].astype(np.int64).sum()) == 0 or (max_steps and self.steps >= max_steps):
                return self.steps
            out = np.asarray(self.local.evaluate(allv[:, :-1].reshape(self.world, nq, W)), np.uint32)
            scores = np.asarray(self.reduce_scatter(out.reshape(self.world, nq, W)), np.uint32).reshape(nq, W)
            self.exchanged_bytes += out.nbytes


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


class FederatedTraversal:
    """LABELLED ALTERNATIVE (not the north-star partitioning): drives one shard's traversal object through
    the federated rounds over shard-local graphs.

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


ShardedTraversal = FederatedTraversal   # round-1 name
