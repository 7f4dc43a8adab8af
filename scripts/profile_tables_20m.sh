#!/bin/bash
# requests and instructions per expansion of the three table forms of trav4_kernel on one workload (20M rows, expansion_add 400,
# one launch of 65536 traversals alone on the device): where the grouped table stands against the bucket tables
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_tables
mkdir -p $OUT
B="bench.py --rows 20000000 --no-cpu-baseline --no-kernel-legs --no-config-legs --secondary-expansion-add 0 --steps 1 --warmup 0 --no-overlap --graph-cache /tmp/radhip_g20"
for t in local auto group; do
  for ctr in "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    tag=${t}_$(echo $ctr | tr ' ' '_' | cut -c1-20)
    timeout -k 10 300 rocprofv3 --kernel-trace -f csv --pmc $ctr -d $OUT/pmc_$tag -o p -- python3 $B --table $t > $OUT/bench_$tag.json 2>> $OUT/session.log || { echo "pmc $tag failed"; continue; }
    python3 scripts/pmc_summarize.py $OUT/pmc_$tag $OUT/pmc_$tag.csv > /dev/null 2>> $OUT/session.log
    rm -rf $OUT/pmc_$tag
  done
  python3 - <<PY
import csv, glob, json
ctr = {}
for f in glob.glob("$OUT/pmc_${t}_*.csv"):
    for r in csv.DictReader(open(f)):
        if "trav4_kernel" in r["kernel"]:
            ctr[r["counter"]] = float(r["sum_over_dispatches"]); ms = float(r["total_ms"]) / int(r["calls"]); k = r["kernel"]
bj = json.loads(open(glob.glob("$OUT/bench_${t}_TCC*.json")[0]).read().strip().splitlines()[-1])
pops = bj["value"] * bj["ms_per_step"] * 1e-3
print(f"table $t ({k}): {pops / (ms * 1e-3) / 1e9:.3f} G expansions/s alone under the profiler; per expansion {ctr['TCC_EA0_RDREQ'] / pops:.2f} reads + {ctr['TCC_EA0_WRREQ'] / pops:.2f} writes ({ctr['TCC_EA0_WRREQ_64B'] / pops:.2f} of 64 B), "
      f"VALU {ctr['SQ_INSTS_VALU'] / pops:.0f} SALU {ctr['SQ_INSTS_SALU'] / pops:.0f} LDS {ctr['SQ_INSTS_LDS'] / pops:.1f} VMEM {ctr['SQ_INSTS_VMEM_RD'] / pops:.2f} + {ctr['SQ_INSTS_VMEM_WR'] / pops:.2f}; {bj['evals_per_expansion']:.2f} evaluations per expansion")
PY
done
