#!/bin/bash
# scripts/kernel_ab.sh "<build dirs>" "<tables>" [rows] [nq] [batches]: scripts/kernel_ab.py for every build x table, one process each
cd "$GRAFT_REPO_ROOT" || exit 1
for b in $1; do for t in $2; do
  if [ "$t" = bucket ]; then unset RADHIP_TABLE; else export RADHIP_TABLE=$t; fi
  RADHIP_LIB=$PWD/rad_amd/$b/librad_hip.so timeout -k 10 300 python3 scripts/kernel_ab.py ${3:-20000000} ${4:-65536} ${5:-7} 2>&1 | tail -2 || exit 1
done; done
