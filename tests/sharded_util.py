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


class OracleRowShard:
    """Test-only `local` engine of rad_amd.sharded.RowShardedTraversal over the CPU oracle: the oracle's
    stepper (traversal cut at the fingerprint read) for this rank's traversals, plain popcounts for the
    candidates whose rows this rank owns.  Same interface as rad_amd.device.DeviceShard."""

    def __init__(self, O, graph, X_shard, row_first, queries_all, rank, world, n_to_score):
        self.O, self.X, self.first = O, X_shard, int(row_first)
        self.rank, self.world = rank, world
        self.Qall = np.ascontiguousarray(queries_all, np.uint8)
        self.nq = self.Qall.shape[0] // world
        self.width = max(graph.cap0, graph.capU)
        self.steppers = [O.Stepper(graph, n_to_score) for _ in range(self.nq)]
        self.pending = [np.empty(0, np.uint32)] * self.nq

    def step(self, scores_in):
        scores_in = np.asarray(scores_in, np.uint32).reshape(self.nq, self.width)
        req = np.full((self.nq, self.width), 0xFFFFFFFF, np.uint32)
        live = 0
        for q, st in enumerate(self.steppers):
            k = self.pending[q].shape[0]
            r = st.step(scores_in[q, :k] & 0xFFFF, scores_in[q, :k] >> 16)
            self.pending[q] = r
            req[q, :r.shape[0]] = r
            live += 1 if st.status == 0 else 0
        return req, live

    def evaluate(self, requests_all):
        requests_all = np.asarray(requests_all, np.uint32).reshape(self.world, self.nq, self.width)
        out = np.zeros_like(requests_all)
        n_rows = self.X.shape[0]
        for r in range(self.world):
            for q in range(self.nq):
                row = requests_all[r, q]
                mine = (row != 0xFFFFFFFF) & (row >= self.first) & (row < self.first + n_rows)
                if mine.any():
                    a, o = self.O.gather(self.X, self.Qall[r * self.nq + q], row[mine] - np.uint32(self.first))
                    out[r, q, mine] = a | (o << 16)
        return out

    def results(self, q):
        r = self.steppers[q].result()
        return r.slots, r.and_cnt, r.or_cnt, r.pop_nodes, r.pop_levels
