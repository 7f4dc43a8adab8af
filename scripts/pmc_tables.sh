#!/bin/bash
# memory-side requests per expansion of the bucket table against the grouped table, same graph, same queries
# (20M hierarchical rows, 32768 traversals): gpurun -- bash scripts/pmc_tables.sh
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmc_tables
mkdir -p $OUT
for ctr in "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
    timeout -k 10 400 rocprofv3 --kernel-trace -f csv --pmc $ctr -d $OUT/pmc_$tag -o p -- python3 scripts/ab_table.py 20000000 2 32768 100000 bucket,group 1 > $OUT/ab_$tag.log 2>&1 || { echo "pmc run $tag failed"; tail -5 $OUT/ab_$tag.log; exit 1; }
    python3 scripts/pmc_summarize.py $OUT/pmc_$tag $OUT/pmc_$tag.csv > /dev/null
    grep -E "trav4_kernel|kernel,calls" $OUT/pmc_$tag.csv
    grep "^rep" $OUT/ab_$tag.log
    rm -rf $OUT/pmc_$tag
done
