#!/bin/bash
# average memory / LDS latencies seen by the wavefronts (SQ_INST_LEVEL_* / instructions issued) of the build and traversal kernels
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_latency
mkdir -p $OUT
for what in build trav; do
  if [ $what = build ]; then CMD="scripts/build_only.py 10000000 400"; else CMD="bench.py --rows 20000000 --no-cpu-baseline --no-kernel-legs --no-config-legs --secondary-expansion-add 0 --steps 1 --warmup 0 --no-overlap --expansion-add 64"; fi
  for ctr in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS_ATOMIC"; do
    tag=${what}_$(echo $ctr | tr ' ' '_' | cut -c1-24)
    timeout -k 10 300 rocprofv3 --kernel-trace -f csv --pmc $ctr -d $OUT/pmc_$tag -o p -- python3 $CMD > /dev/null 2>> $OUT/session.log || { echo "pmc $tag failed"; continue; }
    python3 scripts/pmc_summarize.py $OUT/pmc_$tag $OUT/pmc_$tag.csv > /dev/null 2>> $OUT/session.log
    grep -E "build_insert_kernel|trav4_kernel" $OUT/pmc_$tag.csv
    rm -rf $OUT/pmc_$tag
  done
done
