#!/usr/bin/env python3
"""How many of an expansion's probes ask for a node that was scored only a few expansions ago?  (A per-traversal window of
the last W scored nodes kept on chip would answer those without a memory request.)
    python scripts/recent_hits.py [n_rows = 20M] [n_traversals = 6]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from rad_amd.device import DeviceIndex, DeviceTraversal
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 6
os.environ["RADHIP_TRAV"] = "4"
idx = DeviceIndex(1024, 8, 16, 64)
idx.synth_vectors(n, seed=20260101, mode=2)
idx.link_resident(seed=777, max_batch=16384)
levels, adj0, upper_row, adjU = idx.read_graph()
Q = idx.read_vectors(12345, nt)
t = DeviceTraversal(idx, Q, 100_000, log_pops=True)
t.run()
tot = {w: 0 for w in (16, 32, 64, 128, 256)}
seen_all = probes = 0
for q in range(nt):
    slots, _a, _o = t.results(q)
    nodes, lv = t.pop_log(q)
    when = {int(s): i for i, s in enumerate(slots)}          # position in the scored list
    # replay: the scored count at each pop = number of scored nodes before it; reconstruct by walking pops and rows
    pos = 0
    order = {}
    scored_so_far = 0
    # a node's scored index tells when it was scored; the pop that scored it is the first pop whose row contains it with index >= running count
    running = 0
    for node, l in zip(nodes.tolist(), lv.tolist()):
        row = adj0[node] if l == 0 else adjU[upper_row[node] + l - 1]
        row = row[row != 0xFFFFFFFF]
        if l != 0:
            continue                                         # (level-0 expansions only: 87 % of all)
        newc = 0
        for v in row.tolist():
            i = when.get(v)
            probes += 1
            if i is not None and i < running:                # already scored when this expansion runs
                seen_all += 1
                for w in tot:
                    if i >= running - w:
                        tot[w] += 1
            elif i is not None:
                newc += 1
        running += newc
    # (upper-level expansions also score nodes: `running` is re-synchronised with the scored list as far as level 0 goes)
print(f"{nt} traversals: {probes} level-0 probes, {seen_all} of them for a node already scored ({seen_all / probes:.2f})")
for w, c in tot.items():
    print(f"  scored within the last {w:3d} scored nodes: {c} = {c / probes:.3f} of all probes, {c / max(seen_all, 1):.3f} of the already-scored ones")
