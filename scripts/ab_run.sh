# A/B of traversal-kernel builds on one box: bash scripts/ab_run.sh "<build dirs>" [n] [nq] [nts]
LIBS=${1:-"_build_old _build"}; N=${2:-20000000}; NQ=${3:-30720}; NTS=${4:-100000}
for m in 2 1; do for l in $LIBS; do RADHIP_LIB=$PWD/rad_amd/$l/librad_hip.so timeout -k 10 200 python scripts/ab_bench.py $N $m $NQ $NTS 2>&1 | grep -E "prof|rep 3" || exit 1; done; done
