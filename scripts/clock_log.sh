#!/bin/bash
# the device's clocks, temperature and power once a second beside a bench run (what differs between a 1.61 G and a 1.74 G box?)
cd $GRAFT_REPO_ROOT || exit 1
OUT=gpurun_out/clock_log
mkdir -p $OUT
( while true; do echo "t=$(date +%s.%N)"; rocm-smi --showclocks --showtemp --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Temperature|Power" ; sleep 1; done ) > $OUT/smi.log 2>&1 &
SM=$!
timeout -k 10 420 python3 bench.py --no-cpu-baseline --no-kernel-legs --no-config-legs --secondary-expansion-add 0 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
rc=$?
kill $SM
grep -E "timed steps|warm-up|built|graph" $OUT/bench.err | tail -5
echo "t_end=$(date +%s.%N)"
exit $rc
