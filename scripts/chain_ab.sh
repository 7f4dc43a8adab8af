#!/bin/bash
# --chain 20 against --chain 10 on ONE box (the graph is built once and cached in /tmp of the box)
cd $GRAFT_REPO_ROOT || exit 1
B="bench.py --no-cpu-baseline --no-kernel-legs --no-config-legs --secondary-expansion-add 0 --graph-cache /tmp/radhip_graph --steps 20 --warmup 5"
for c in 20 10 20 10; do
  timeout -k 10 400 python3 $B --chain $c > gpurun_out/chain_ab_$c.json 2>> gpurun_out/chain_ab.err || exit 1
  python3 -c "
import json
j=json.loads(open('gpurun_out/chain_ab_$c.json').read().strip().splitlines()[-1]); r=j['roofline']
print('chain $c: %.3f G expansions/s, %.1f ms per step, kernel %.1f ms per step, frac %.4f' % (j['value']/1e9, j['ms_per_step'], r['avg_launch_ms']*r['launches']/j['steps'], r['frac']))
" | tee -a gpurun_out/chain_ab.log
done
