#!/bin/bash
# Memory-side requests per pop of the WIDE-row workloads ("... or a counter-backed statement of its request bound"):
# BASELINE configs[4] (2048-bit, connectivity 32: rows of 64 slots, trav_kernel, closed-form graph over 20M rows, 6144 traversals)
# and the reference notebook's shape (1024-bit, connectivity 16: rows of 32 slots, trav4_kernel's WIDE form, graph built with
# expansion_add 400 over 20M rows, 65536 traversals).  One launch alone on the device per pass; counters in their own passes.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_wide_r04
mkdir -p $OUT
COMMON="--no-cpu-baseline --no-kernel-legs --no-config-legs --secondary-expansion-add 0 --no-overlap --steps 1 --warmup 0 --rows 20000000"
run() {   # tag, bench flags
    local tag=$1; shift
    for ctr in "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
        local c=$(echo $ctr | tr ' ' '_' | cut -c1-24)
        timeout -k 10 400 rocprofv3 --kernel-trace -f csv --pmc $ctr -d $OUT/pmc_${tag}_$c -o p -- python3 bench.py $COMMON "$@" > $OUT/bench_${tag}_$c.json 2>> $OUT/session.log || { echo "pass $tag $c failed" | tee -a $OUT/session.log; return 1; }
        python3 scripts/pmc_summarize.py $OUT/pmc_${tag}_$c $OUT/pmc_${tag}_$c.csv > /dev/null 2>> $OUT/session.log
        grep -E "trav4_kernel|trav_kernel|kernel,calls" $OUT/pmc_${tag}_$c.csv | tee -a $OUT/session.log
        rm -rf $OUT/pmc_${tag}_$c
    done
}
run c4 --ndim 2048 --connectivity 32 --nq 6144 --graph synthetic --corpus-mode 1 --table auto || exit 1
run nb --ndim 1024 --connectivity 16 --expansion-add 400 --nq 65536 --table local --graph-cache /tmp/radhip_nb || exit 1
run nbb --ndim 1024 --connectivity 16 --expansion-add 400 --nq 65536 --table auto --graph-cache /tmp/radhip_nb || exit 1
python3 - <<'PY' | tee -a $OUT/session.log
import csv, glob, json, os
OUT = "gpurun_out/prof_wide_r04"
for tag, what in (("c4", "configs[4]: 2048-bit, connectivity 32, closed-form graph, 6144 traversals (trav_kernel)"),
                  ("nb", "notebook shape: 1024-bit, connectivity 16, expansion_add 400, 65536 traversals (trav4_kernel WIDE, local table)"),
                  ("nbb", "notebook shape, slot-hashed bucket table")):
    ctr = {}
    for f in glob.glob(f"{OUT}/pmc_{tag}_*.csv"):
        for r in csv.DictReader(open(f)):
            if "trav" in r["kernel"]:
                ctr[r["counter"]] = float(r["sum_over_dispatches"]); ms = float(r["total_ms"]) / int(r["calls"])
    bj = json.loads(open(glob.glob(f"{OUT}/bench_{tag}_TCC*.json")[0]).read().strip().splitlines()[-1])
    pops = bj["value"] * bj["ms_per_step"] * 1e-3
    rd, wr, w64 = ctr["TCC_EA0_RDREQ"] / pops, ctr["TCC_EA0_WRREQ"] / pops, ctr["TCC_EA0_WRREQ_64B"] / pops
    t = rd / 44.0 + (wr - w64) / 21.9 + w64 / 49.5
    print(f"{what}\n   {bj['evals_per_expansion']:.2f} evaluations per pop; per pop {rd:.1f} line reads, {wr - w64:.1f} writes <= 32 B, {w64:.1f} writes of 64 B "
          f"-> {t:.3f} ns by the request-cost model = {1 / t:.3f} G pops/s; measured (one launch alone, under the profiler) {pops / (ms * 1e-3) / 1e9:.3f} G pops/s, "
          f"frac {bj['roofline']['frac']:.3f}; instructions per pop: VALU {ctr.get('SQ_INSTS_VALU', 0) / pops:.0f} SALU {ctr.get('SQ_INSTS_SALU', 0) / pops:.0f}")
PY
echo done | tee -a $OUT/session.log
