#!/bin/bash
# world-1 product loop of the row-sharded mode on one GPU (20M rows): step engines and traversals per rank
#   gpurun -- bash scripts/sharded_engines.sh
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03_sharded_engines
mkdir -p $OUT
for cfg in "row 32768 2" "row 32768 0" "row 16384 2" "row 49152 2"; do
    set -- $cfg
    echo "== engine $1, $2 traversals, speculation $3" | tee -a $OUT/session.log
    RADHIP_SHARD_ENGINE=$1 RADHIP_SHARD_SPEC=$3 timeout -k 10 280 python3 bench.py --mode sharded --rows 20000000 --sharded-nq $2 --steps 1 --warmup 0 --no-cpu-baseline \
        > $OUT/sharded_world1_$1_$2_s$3.json 2>> $OUT/session.log || { echo "failed" | tee -a $OUT/session.log; exit 1; }
    python3 -c "
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = d['sharded']
print({k: s[k] for k in ('value', 'ms_per_step', 'traversals_per_gpu_per_step', 'frontier_steps_per_step', 'engine', 'parity_vs_single_gpu', 'state_bytes_per_rank')}, s['speculation']['depth'], s.get('phase_us'))
" $OUT/sharded_world1_$1_$2_s$3.json | tee -a $OUT/session.log
done
