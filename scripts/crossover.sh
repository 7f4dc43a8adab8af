# trav_kernel (one traversal per wavefront) vs trav4_kernel (four) by batch size: bash scripts/crossover.sh [mode]
for nq in 1024 2048 4096 6144 8192 12288; do for k in 1 4; do
  RADHIP_TRAV=$k RADHIP_LIB=$PWD/rad_amd/_build/librad_hip.so timeout -k 10 200 python scripts/ab_bench.py 20000000 ${1:-2} $nq 100000 2>&1 | grep "rep 3" | sed "s/.*rep 3/trav$k nq $nq:/" || exit 1
done; done
