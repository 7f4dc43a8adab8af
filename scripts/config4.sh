# BASELINE config[4] (2048-bit rows, connectivity 32 -> trav_kernel, one traversal per wavefront) with the libraries named:
#   bash scripts/config4.sh "<build dirs>" [nq]
LIBS=${1:-"_build"}; NQ=${2:-6144}
for l in $LIBS; do
  RADHIP_LIB=$PWD/rad_amd/$l/librad_hip.so timeout -k 10 280 python bench.py --ndim 2048 --connectivity 32 --rows 20000000 --nq $NQ --graph synthetic --corpus-mode 1 \
      --no-cpu-baseline --no-reference-corpus --steps 12 --warmup 3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print('$l', 'nq', $NQ, 'value %.1f M exp/s' % (d['value'] / 1e6), 'evals/exp %.1f' % d['evals_per_expansion'], 'launch %.1f ms' % r['avg_launch_ms'], 'frac %.3f' % r['frac'], 'parity', d.get('parity_sample'))" || exit 1
done
