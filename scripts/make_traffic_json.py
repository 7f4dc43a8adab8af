#!/usr/bin/env python3
"""profiles/traffic_latest.json from one PMC session (scripts/profile_r04.sh): HBM bytes per launch of the traversal
kernel from the memory-side counters, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950
(FETCH_SIZE x 2; cross-checked against TCC_EA0_RDREQ x 128 B), requests and instructions per expansion.
    python scripts/make_traffic_json.py <session dir>   (run on the box: it asks the library for its build id)"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rad_amd import _lib

d = sys.argv[1]
ctr, ms, calls = {}, {}, {}
for f in glob.glob(os.path.join(d, "pmc_*.csv")):
    for r in csv.DictReader(open(f)):
        if "trav4_kernel" in r["kernel"] or "trav_kernel" in r["kernel"]:
            ctr[r["counter"]] = float(r["sum_over_dispatches"])
            ms[r["counter"]] = float(r["total_ms"]); calls[r["counter"]] = int(r["calls"])
            kname = r["kernel"]
bj = None
for f in sorted(glob.glob(os.path.join(d, "bench_pmc_*.json"))):
    try:
        bj = json.loads(open(f).read().strip().splitlines()[-1]); break
    except Exception:
        pass
if bj is None or "FETCH_SIZE" not in ctr:
    raise SystemExit("no PMC data found")
n_launch = calls["FETCH_SIZE"]
rd = ctr["FETCH_SIZE"] * 1024 * 2 / n_launch
wr = ctr["WRITE_SIZE"] * 1024 / calls["WRITE_SIZE"]
pops = bj["value"] * bj["ms_per_step"] * 1e-3
evals = bj["evals_per_s"] * bj["ms_per_step"] * 1e-3
alg = bj["roofline"]["algorithmic_bytes_per_launch"]
out = {
    "build_id": _lib.build_id(), "traverse_build_id": _lib.traverse_build_id(), "kernel": kname, "table": bj["roofline"]["table"], "graph": "built", "corpus_mode": bj["config"]["corpus_mode"],
    "n": bj["config"]["rows"], "nq": bj["config"]["nq_per_gpu"], "n_to_score": bj["config"]["n_to_score"], "expansion_add": bj["config"].get("expansion_add", 64),
    "method": "rocprofv3 -f csv --kernel-trace --pmc <group> in separate passes over `python3 bench.py --no-cpu-baseline --no-reference-corpus "
              "--steps 1 --warmup 0` (scripts/profile_r04.sh), summed per kernel on the box; FETCH_SIZE x 2 per MI355X_MICROARCH.md "
              "(= TCC_EA0_RDREQ x 128 B when every read is a 128-B request); WRITE_SIZE x 1024",
    "fetch_size_kb": ctr["FETCH_SIZE"], "write_size_kb": ctr["WRITE_SIZE"],
    "read_requests_128B": ctr.get("TCC_EA0_RDREQ"), "read_requests_32B": ctr.get("TCC_EA0_RDREQ_32B"),
    "write_requests": {"total": ctr.get("TCC_EA0_WRREQ"), "64B": ctr.get("TCC_EA0_WRREQ_64B")},
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "launch_ms_under_pmc": ms["FETCH_SIZE"] / n_launch, "hbm_real_gbs": (rd + wr) / (ms["FETCH_SIZE"] / n_launch * 1e-3) / 1e9,
    "expansions_per_launch": pops, "evaluations_per_launch": evals, "algorithmic_bytes_per_launch": alg,
    "traffic_over_algorithmic": (rd + wr) / alg,
    "requests_per_expansion": {"reads": (ctr.get("TCC_EA0_RDREQ") or 0) / pops, "writes": (ctr.get("TCC_EA0_WRREQ") or 0) / pops},
    "sq_per_expansion": {k: ctr[k] / pops for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR") if k in ctr},
    "sq_cycles": {k: ctr[k] for k in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY") if k in ctr},
}
if "SQ_WAVE_CYCLES" in ctr:
    out["sq_cycles"]["wait_any_over_wave_cycles"] = ctr.get("SQ_WAIT_ANY", 0) / ctr["SQ_WAVE_CYCLES"]
    out["sq_cycles"]["active_inst_over_wave_cycles"] = ctr.get("SQ_ACTIVE_INST_ANY", 0) / ctr["SQ_WAVE_CYCLES"]
print(json.dumps(out, indent=1))
