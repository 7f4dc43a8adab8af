"""Test-only helpers for the sharded traversal: an oracle-backed stand-in for the local
traversal object (same interface as rad_amd.device.DeviceTraversal) and a sequential
reference of the federated rounds."""
import numpy as np

KEY_EMPTY = 0xFFFFFFFFFFFFFFFF


class OracleLocalTraversal:
    """set_targets / run / frontier over the CPU oracle.  The oracle cannot resume, so every
    run() restarts from scratch with the current target: identical by determinism."""

    def __init__(self, O, graph, X, Q, local_cap):
        from rad_amd import _lib
        self.O, self.g, self.X, self.Q = O, graph, X, Q
        self.nq = Q.shape[0]
        self.cap = min(int(local_cap), graph.n)
        self.targets = np.full(self.nq, self.cap, np.uint64)
        self._key = _lib.lib().radhip_rad_key
        self.res = [None] * self.nq

    def set_targets(self, t):
        self.targets = np.minimum(np.asarray(t, np.uint64), np.uint64(self.cap))

    def run(self, max_pops=0):
        assert max_pops == 0
        for i in range(self.nq):
            self.res[i] = self.O.rad_traverse(self.g, self.X, self.Q[i], int(self.targets[i]), log_pops=False)
        return 0

    def frontier(self):
        keys = np.empty(self.nq, np.uint64)
        scored = np.empty(self.nq, np.uint64)
        for i, r in enumerate(self.res):
            scored[i] = r.slots.shape[0]
            keys[i] = KEY_EMPTY if r.frontier is None else self._key(*r.frontier)
        return keys, scored

    def results(self, i):
        r = self.res[i]
        return r.slots, r.and_cnt, r.or_cnt


def make_shards(O, world, n_per, ndim, M, cap0, seed):
    """Shard r = rows [r*n_per, (r+1)*n_per) of one logical corpus + its own synthetic graph."""
    shards = []
    for r in range(world):
        X = O.synth_rows(r * n_per, n_per, world * n_per, ndim, seed, 1)
        g = O.synth_graph(n_per, M, cap0, seed + 100 + r)
        shards.append((X, g))
    return shards
