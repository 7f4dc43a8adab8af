#!/bin/bash
# PMC of BASELINE config[4] (trav_kernel: 20M x 2048-bit, connectivity 32, 6144 traversals): instructions per expansion
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_c4
mkdir -p $OUT
B="bench.py --ndim 2048 --connectivity 32 --rows 20000000 --nq 6144 --graph synthetic --corpus-mode 1 --no-cpu-baseline --no-reference-corpus --steps 1 --warmup 0"
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "TCC_EA0_RDREQ TCC_EA0_WRREQ TCC_EA0_WRREQ_64B"; do
    tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
    timeout -k 10 300 rocprofv3 --kernel-trace -f csv --pmc $ctr -d $OUT/pmc_$tag -o p -- python3 $B > $OUT/bench_$tag.json 2>> $OUT/session.log || echo "pass $tag failed"
    python3 scripts/pmc_summarize.py $OUT/pmc_$tag $OUT/pmc_$tag.csv > /dev/null 2>> $OUT/session.log
    grep -E "trav_kernel|kernel,calls" $OUT/pmc_$tag.csv
    rm -rf $OUT/pmc_$tag
done
python3 -c "
import json; d=json.loads(open('$OUT/bench_SQ_INSTS_VALU_SQ_INSTS_SALU_SQ_INSTS_LDS_S.json').readline()); r=d['roofline']
print('value', d['value'], 'evals/exp', d['evals_per_expansion'], 'alg bytes', r['algorithmic_bytes_per_launch'], 'launch ms', r['avg_launch_ms'], 'frac', r['frac'])"
