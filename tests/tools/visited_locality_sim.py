"""Offline study (CPU, oracle graph): could a traversal answer some of its visited-set probes without a memory request?
Two exact sources of knowledge are simulated on a host-built HNSW graph with a best-first level-0 traversal:
  (1) the discoverer: when u was first reached through p, every neighbour of p (and p) is visited by the time u is popped;
  (2) a small direct-mapped cache of recently probed ids (what an LDS-resident filter could hold).
Result (100000 rows, connectivity 8, expansion_add 128, 20000 scored; profiles/r04/README.md section 6): (1) answers 1.2 of 10.8
probes per pop, (2) 0.08 / 0.19 / 0.27 / 0.56 at 64 / 256 / 1024 / 4096 entries - neither pays for its own bookkeeping.
usage: python tests/tools/visited_locality_sim.py [rows] [expansion_add] [corpus mode] [n_to_score]"""
import sys, time, heapq, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from oracle import rad_oracle as O
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
M = 8; ef = int(sys.argv[2]) if len(sys.argv) > 2 else 128
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
rows = O.synth_rows(0, N, N, 1024, 3, mode)
t = time.time(); h = O.Hnsw(1024, M, 0, ef, 7); h.add(rows, 64); g = h.graph(); print("built", time.time() - t, "s", g.cap0, g.max_level)
adj = g.adj0; INV = 0xFFFFFFFF
nbrs = [set(int(x) for x in r if x != INV and x < N) for r in adj]
deg = np.array([len(s) for s in nbrs]); print("mean valid level-0 degree", deg.mean())
# static: for edge p->u, share of N(u) inside N(p) + {p}
rng = np.random.default_rng(0); tot = 0; cov = 0
for p in rng.integers(0, N, 20000):
    P = nbrs[p] | {int(p)}
    for u in nbrs[p]:
        tot += len(nbrs[u]); cov += len(nbrs[u] & P)
print("static share of N(u) covered by N(p)+p over edges p->u: %.3f" % (cov / tot))
bits = np.unpackbits(rows, axis=1)
def run(qi, n_to_score):
    q = rows[qi]
    pc = np.bitwise_count
    def dist(v):
        a = int(pc(rows[v] & q).sum()); o = int(pc(rows[v] | q).sum()); return 1.0 - a / o if o else 0.0
    visited = {}; heap = []
    e = g.entry; visited[e] = -1; heapq.heappush(heap, (dist(e), e))
    pops = valid = new = known = known2 = 0
    while heap and len(visited) < n_to_score:
        d, u = heapq.heappop(heap); p = visited[u]; pops += 1
        K = (nbrs[p] | {p}) if p >= 0 else set()
        gp = visited[p] if p >= 0 else -1
        K2 = K | ((nbrs[gp] | {gp}) if gp >= 0 else set())
        for v in nbrs[u]:
            valid += 1
            if v in K: known += 1
            if v in K2: known2 += 1
            if v not in visited:
                assert v not in K
                visited[v] = u; new += 1; heapq.heappush(heap, (dist(v), v))
    return pops, valid, new, known, known2
T = np.zeros(5)
for qi in rng.integers(0, N, 8): T += run(int(qi), int(sys.argv[4]) if len(sys.argv) > 4 else 20000)
print("per pop: valid %.2f new %.2f already-visited %.2f known-by-discoverer %.2f  +grand-discoverer %.2f" % (T[1]/T[0], T[2]/T[0], (T[1]-T[2])/T[0], T[3]/T[0], T[4]/T[0]))
def run2(qi, n_to_score, C):
    q = rows[qi]; pc = np.bitwise_count
    def dist(v):
        a = int(pc(rows[v] & q).sum()); o = int(pc(rows[v] | q).sum()); return 1.0 - a / o if o else 0.0
    visited = set(); heap = []; cache = [-1] * C
    e = g.entry; visited.add(e); heapq.heappush(heap, (dist(e), e))
    pops = valid = new = hit = 0
    while heap and len(visited) < n_to_score:
        d, u = heapq.heappop(heap); pops += 1
        for v in nbrs[u]:
            valid += 1
            s = (v * 2654435761 >> 7) % C
            if cache[s] == v: hit += 1; continue
            if v not in visited:
                visited.add(v); new += 1; heapq.heappush(heap, (dist(v), v))
            cache[s] = v
    return pops, valid, new, hit
for C in (64, 256, 1024, 4096):
    T = np.zeros(4)
    for qi in rng.integers(0, N, 4): T += run2(int(qi), 20000, C)
    print("cache %5d entries: per pop valid %.2f new %.2f, probes answered by the cache %.2f" % (C, T[1]/T[0], T[2]/T[0], T[3]/T[0]))
