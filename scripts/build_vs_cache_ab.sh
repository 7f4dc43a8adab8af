#!/bin/bash
# does a graph BUILT in the benching process leave the device in a slower state than one loaded from a cache file? (one box)
cd $GRAFT_REPO_ROOT || exit 1
B="bench.py --no-cpu-baseline --no-kernel-legs --no-config-legs --secondary-expansion-add 0 --steps 20 --warmup 5"
i=0
for mode in build cache-save cache-load build cache-load; do
  i=$((i+1))
  case $mode in build) F="";; *) F="--graph-cache /tmp/radhip_graph_ab";; esac
  timeout -k 10 400 python3 $B $F > gpurun_out/bvc_$i.json 2>> gpurun_out/bvc.err || exit 1
  python3 -c "
import json
j=json.loads(open('gpurun_out/bvc_$i.json').read().strip().splitlines()[-1]); r=j['roofline']
print('run $i ($mode): %.3f G expansions/s, %.1f ms per step, kernel %.1f ms per step' % (j['value']/1e9, j['ms_per_step'], r['avg_launch_ms']*r['launches']/j['steps']))
" | tee -a gpurun_out/bvc.log
done
