#!/bin/bash
# rocprofv3 kernel statistics of a WHOLE default bench run (every leg but the CPU baseline): one file with the average duration of
# every kernel of the path (traversal, scan, top-k, gather, build)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_full
mkdir -p $OUT
timeout -k 10 800 rocprofv3 --kernel-trace --stats -f csv -d $OUT/stats -o bench -- python3 bench.py --no-cpu-baseline --steps 6 --warmup 6 > $OUT/bench_full_under_rocprof.json 2> $OUT/session.log || { echo "run failed"; tail -5 $OUT/session.log; exit 1; }
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/bench_full_kernel_stats.csv \;
rm -rf $OUT/stats
head -30 $OUT/bench_full_kernel_stats.csv | cut -c1-70,160-330
