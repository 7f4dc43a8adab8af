#!/bin/bash
# rocprofv3 of the K1 / K2 / top-k legs alone (closed-form graph, no traversal legs): kernel durations, then instruction counters
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_topk
mkdir -p $OUT
B="bench.py --graph synthetic --corpus-mode 1 --steps 1 --warmup 0 --no-config-legs --secondary-expansion-add 0 --no-cpu-baseline --no-overlap --nq 8192"
for rows in ${ROWS:-0 1}; do
  export RADHIP_TOPK_ROWS=$rows
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats$rows -o k -- python3 $B > $OUT/bench_rows$rows.json 2>> $OUT/session.log || { echo "stats run failed"; exit 1; }
  find $OUT/stats$rows -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_rows$rows.csv \;
  rm -rf $OUT/stats$rows
  grep -E "topk|scan_kernel|gather_kernel" $OUT/kernel_stats_rows$rows.csv | cut -c1-60,200-400 | tee -a $OUT/session.log
  timeout -k 10 300 rocprofv3 --kernel-trace -f csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/pmc$rows -o p -- python3 $B > /dev/null 2>> $OUT/session.log || { echo "pmc run failed"; exit 1; }
  python3 scripts/pmc_summarize.py $OUT/pmc$rows $OUT/pmc_rows$rows.csv > /dev/null 2>> $OUT/session.log
  grep -E "topk_(scan|rows)_kernel|scan_kernel" $OUT/pmc_rows$rows.csv | tee -a $OUT/session.log
  rm -rf $OUT/pmc$rows
done
echo done
