#!/bin/bash
# world-1 product loop of the row-sharded mode on one GPU (20M rows): traversals per batch, slots, speculation
#   gpurun -- bash scripts/sharded_engines.sh
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r03_sharded_slots
mkdir -p $OUT
for cfg in "32768 0 2" "65536 16384 2" "65536 32768 2" "65536 16384 0" "131072 32768 2" "131072 16384 2"; do
    set -- $cfg
    echo "== $1 traversals per batch, $2 slots (0 = one per traversal), speculation $3" | tee -a $OUT/session.log
    RADHIP_SHARD_SPEC=$3 timeout -k 10 280 python3 bench.py --mode sharded --rows 20000000 --sharded-nq $1 --sharded-slots $2 --steps 1 --warmup 0 --no-cpu-baseline \
        > $OUT/sharded_world1_$1_$2_s$3.json 2>> $OUT/session.log || { echo "failed" | tee -a $OUT/session.log; tail -3 $OUT/session.log; continue; }
    python3 -c "
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); s = d['sharded']
print({k: s[k] for k in ('value', 'ms_per_step', 'traversals_per_gpu_per_step', 'slots_per_gpu', 'frontier_steps_per_step', 'engine', 'parity_vs_single_gpu', 'state_bytes_per_rank')}, s['speculation']['depth'])
" $OUT/sharded_world1_$1_$2_s$3.json | tee -a $OUT/session.log
done
