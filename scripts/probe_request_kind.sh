#!/bin/bash
# does the memory type of the allocation change what a 16-B probe / a 4-B entry store costs?  (fine-grained and uncached
# allocations against plain hipMalloc; same kernels as probe_request_size.sh)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/probe_kind
mkdir -p $OUT scripts/bin
hipcc --offload-arch=gfx950 -O3 -o scripts/bin/probe_request_size scripts/probe_request_size.hip || exit 1
GIB=${1:-32}
: > $OUT/session.log
for kind in 3 1; do
    echo "== plain run (rates), allocation kind $kind" | tee -a $OUT/session.log
    timeout -k 10 200 scripts/bin/probe_request_size $GIB 16 $kind | tee -a $OUT/session.log || exit 1
    for ctr in "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "FETCH_SIZE" "WRITE_SIZE"; do
        tag=k${kind}_$(echo $ctr | tr ' ' '_' | cut -c1-40)
        echo "== pmc $ctr, allocation kind $kind" | tee -a $OUT/session.log
        timeout -k 10 300 rocprofv3 --kernel-trace -f csv --pmc $ctr -d $OUT/pmc_$tag -o p -- scripts/bin/probe_request_size $GIB 16 $kind >> $OUT/session.log 2>&1 || { echo "pmc run $tag failed" | tee -a $OUT/session.log; continue; }
        python3 scripts/pmc_summarize.py $OUT/pmc_$tag $OUT/pmc_$tag.csv > /dev/null 2>> $OUT/session.log
        cat $OUT/pmc_$tag.csv | tee -a $OUT/session.log
        rm -rf $OUT/pmc_$tag
    done
done
echo done | tee -a $OUT/session.log
