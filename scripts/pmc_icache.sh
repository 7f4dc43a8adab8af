#!/bin/bash
# instruction fetch of trav4_kernel: rows that keep their traversal (static) against rows that take traversals from the
# counter, narrow (connectivity 8) and WIDE (16) form:   gpurun -- bash scripts/pmc_icache.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_icache
mkdir -p $OUT
for M in 8 16; do for st in 1 0; do
    tag=M${M}_static$st
    RADHIP_TRAV_STATIC=$st timeout -k 10 400 rocprofv3 --kernel-trace -f csv --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQC_ICACHE_REQ SQC_ICACHE_MISSES -d $OUT/p_$tag -o p -- python3 scripts/one_batch.py $M 32768 > $OUT/run_$tag.log 2>&1 || { echo "$tag failed"; tail -3 $OUT/run_$tag.log; continue; }
    python3 scripts/pmc_summarize.py $OUT/p_$tag $OUT/pmc_$tag.csv > /dev/null
    echo "== $tag"; grep "^M=" $OUT/run_$tag.log; grep -E "trav4_kernel" $OUT/pmc_$tag.csv
    rm -rf $OUT/p_$tag
done; done
