#!/bin/bash
# bytes per memory-side request for each access shape of the traversal kernel (VERDICT r03 #7b): run on the GPU box through gpurun.
# Counters in their own passes, --kernel-trace only, the program itself behind `--`; summed per kernel by scripts/pmc_summarize.py.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/probe_req
mkdir -p $OUT scripts/bin
[ -x scripts/bin/probe_request_size ] || hipcc --offload-arch=gfx950 -O3 -o scripts/bin/probe_request_size scripts/probe_request_size.hip || exit 1
GIB=${1:-32}
echo "== plain run (rates)" | tee $OUT/session.log
timeout -k 10 120 scripts/bin/probe_request_size $GIB 16 | tee -a $OUT/session.log || exit 1
timeout -k 10 120 scripts/bin/probe_request_size $GIB 4 | tee -a $OUT/session.log || exit 1
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "TCC_HIT TCC_MISS TCC_REQ"; do
    tag=$(echo $ctr | tr ' ' '_' | cut -c1-48)
    echo "== pmc $ctr" | tee -a $OUT/session.log
    timeout -k 10 200 rocprofv3 --kernel-trace -f csv --pmc $ctr -d $OUT/pmc_$tag -o p -- scripts/bin/probe_request_size $GIB 16 >> $OUT/session.log 2>&1 || { echo "pmc run $tag failed" | tee -a $OUT/session.log; continue; }
    python3 scripts/pmc_summarize.py $OUT/pmc_$tag $OUT/pmc_$tag.csv > /dev/null 2>> $OUT/session.log
    cat $OUT/pmc_$tag.csv | tee -a $OUT/session.log
    rm -rf $OUT/pmc_$tag
done
echo done | tee -a $OUT/session.log
