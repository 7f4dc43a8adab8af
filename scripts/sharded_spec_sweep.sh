cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "2 4" "2 6" "2 8" "2 16" "1 8"; do
    set -- $cfg
    RADHIP_SHARD_SPEC=$1 RADHIP_SHARD_INNER=$2 timeout -k 10 280 python3 bench.py --mode sharded --rows 20000000 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['sharded']
print('spec $1 inner $2:', round(s['value']/1e6,1), 'M', s['frontier_steps_per_step'], 'steps', round(s['ms_per_step']*1e3/s['frontier_steps_per_step'],1), 'us/step', s['parity_vs_single_gpu'], 'wasted', round(s['speculation']['wasted_fraction_of_all_evaluations'],3))"
done
