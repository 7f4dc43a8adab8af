#!/bin/bash
# the K1 / K2 / top-k legs of the bench line, once per form of the 1024-bit kernels: a row across eight lanes (RADHIP_TOPK_ROWS=0
# RADHIP_SCAN_ROWS=0) and a row per lane (the default)
cd $GRAFT_REPO_ROOT || exit 1
for rows in 0 1; do
  RADHIP_TOPK_ROWS=$rows RADHIP_SCAN_ROWS=$rows timeout -k 10 250 python3 bench.py --graph synthetic --corpus-mode 1 --steps 1 --warmup 0 --no-config-legs --secondary-expansion-add 0 --no-cpu-baseline --no-overlap --nq 8192 > gpurun_out/kl_rows$rows.json 2> gpurun_out/kl_rows$rows.err || exit 1
  python3 -c "
import json,sys
j=json.loads(open('gpurun_out/kl_rows$rows.json').read().strip().splitlines()[-1])
print('rows per lane = $rows', {k:(round(v['ms'],3), round(v['GB/s'])) for k,v in j['kernels'].items()})
"
done
