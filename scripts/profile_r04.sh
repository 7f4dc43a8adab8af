#!/bin/bash
# round-4 profiling session on the GPU box (run through gpurun): rocprofv3 kernel stats of the default bench's headline leg, then PMC
# passes (counters in their own runs, --kernel-trace only; the program itself behind `--`) summed per kernel by
# scripts/pmc_summarize.py, then scripts/make_traffic_json.py -> gpurun_out/prof_r04/traffic.json (copied to
# profiles/traffic_latest.json, keyed to the traversal kernels' build id).  The headline graph (expansion_add 400) takes 160 s to
# build: the first run saves it (--graph-cache, /tmp on the box), the others load it — the kernels that are profiled are the same.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_r04
mkdir -p $OUT
B="bench.py --no-cpu-baseline --no-kernel-legs --no-config-legs --secondary-expansion-add 0 --graph-cache /tmp/radhip_graph"
echo "== kernel trace + stats" | tee -a $OUT/session.log
timeout -k 10 700 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- python3 $B --steps 6 --warmup 6 > $OUT/bench_under_rocprof.json 2>> $OUT/session.log || echo "stats run failed" | tee -a $OUT/session.log
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_kernel_stats.csv \;
head -8 $OUT/bench_kernel_stats.csv | tee -a $OUT/session.log
rm -rf $OUT/stats
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
    tag=$(echo $ctr | tr ' ' '_' | cut -c1-48)
    echo "== pmc $ctr" | tee -a $OUT/session.log
    # one launch of the batch, alone on the device (no overlap: the counters are per launch)
    timeout -k 10 500 rocprofv3 --kernel-trace -f csv --pmc $ctr -d $OUT/pmc_$tag -o p -- python3 $B --steps 1 --warmup 0 --no-overlap > $OUT/bench_pmc_$tag.json 2>> $OUT/session.log || { echo "pmc run $tag failed" | tee -a $OUT/session.log; }
    python3 scripts/pmc_summarize.py $OUT/pmc_$tag $OUT/pmc_$tag.csv > /dev/null 2>> $OUT/session.log
    grep -E "trav4_kernel|kernel,calls" $OUT/pmc_$tag.csv | tee -a $OUT/session.log
    rm -rf $OUT/pmc_$tag
done
python3 scripts/make_traffic_json.py $OUT > $OUT/traffic.json 2>> $OUT/session.log && cat $OUT/traffic.json | tee -a $OUT/session.log
echo done | tee -a $OUT/session.log
