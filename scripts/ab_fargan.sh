#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for v in "$@"; do
  echo "== variant [$v]"
  PROMONET_HIP_LIB=$ROOT/promonet_amd/lib/libpromonet_hip$v.so timeout 300 python $ROOT/scripts/bench_fargan.py 2>&1 | grep -E "^(fp32|mixed|f16) (1|32|64|256) " | cut -c1-150
done
