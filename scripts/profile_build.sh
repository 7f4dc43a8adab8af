#!/bin/bash
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_build
mkdir -p $OUT
N=${1:-10000000}
timeout -k 10 200 python3 scripts/build_only.py $N 400 | tee $OUT/plain.log || exit 1
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-30)
  timeout -k 10 300 rocprofv3 --kernel-trace -f csv --pmc $ctr -d $OUT/pmc_$tag -o p -- python3 scripts/build_only.py $N 400 > /dev/null 2>> $OUT/session.log || { echo "pmc $tag failed"; exit 1; }
  python3 scripts/pmc_summarize.py $OUT/pmc_$tag $OUT/pmc_$tag.csv > /dev/null 2>> $OUT/session.log
  grep -E "build_insert|build_reverse|kernel,calls" $OUT/pmc_$tag.csv
  rm -rf $OUT/pmc_$tag
done
