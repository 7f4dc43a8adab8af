import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from golden_util import golden, load_graph_npz
from rad_amd.index import Index
from rad_amd.traverser import TanimotoRADTraverser
tag, M = sys.argv[1], int(sys.argv[2])
z = load_graph_npz(f"g1{tag}_graph.npz")
idx = Index(ndim=z["fps"].shape[1] * 8, connectivity=M, connectivity_base=z["adj0"].shape[1])
idx.load_graph(None, z["fps"], z["levels"], z["adj0"], z["upper_row"], z["adjU"], int(z["max_level"]), int(z["entry"]))
for c in golden()[f"g1{tag}"]:
    t = TanimotoRADTraverser(idx, z["queries"][c["query"]:c["query"] + 1], log_pops=True)
    t.traverse(n_to_score=c["n_to_score"])
    nodes, levels = t._trav.pop_log(0)
    want = c["pop_nodes"]
    got = nodes.tolist()
    k = next((i for i in range(min(len(got), len(want))) if got[i] != want[i] or levels[i] != c["pop_levels"][i]), None)
    st = t._trav.stats()
    print(f"query {c['query']} nts {c['n_to_score']}: pops got {len(got)} want {len(want)} first diff at {k} repivots {st.n_repivot} flushes {st.n_flush} status {st.status}")
    if k is not None:
        print("  got ", list(zip(got[max(0,k-3):k+4], levels[max(0,k-3):k+4].tolist())))
        print("  want", list(zip(want[max(0,k-3):k+4], c["pop_levels"][max(0,k-3):k+4])))
        break
    t.shutdown()
