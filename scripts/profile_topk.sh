#!/bin/bash
# PMC of the exact top-k scan: is it the VALU that caps it below the plain scan's 5.5 TB/s?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_topk
mkdir -p $OUT
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE"; do
    tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
    timeout -k 10 300 rocprofv3 --kernel-trace -f csv --pmc $ctr -d $OUT/pmc_$tag -o p -- python3 scripts/topk_bench.py 100000000 > $OUT/topk_$tag.log 2>> $OUT/session.log
    python3 scripts/pmc_summarize.py $OUT/pmc_$tag $OUT/pmc_$tag.csv > /dev/null 2>> $OUT/session.log
    grep -E "topk_scan|scan_kernel|kernel,calls" $OUT/pmc_$tag.csv
    rm -rf $OUT/pmc_$tag
done
cat $OUT/topk_FETCH_SIZE.log
