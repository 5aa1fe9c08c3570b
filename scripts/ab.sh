#!/bin/bash
# On the GPU box: interleaved A/B of library variants, 3 rounds each.
# usage: scripts/ab.sh "" _suffix1 _suffix2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for v in "$@"; do
    PROMONET_HIP_LIB=$ROOT/promonet_amd/lib/libpromonet_hip$v.so python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernels']
print('variant[$v] round $round: %.2f ms  c128_k11 %.3f c128_k7 %.3f c256_k11 %.3f blk64_k11 %.3f blk32_k11 %.3f blk32_k3 %.3f blk32_k7 %.3f blk64_k3 %.3f blk64_k7 %.3f blk128_k3 %.3f' % (r['ms_per_step'], k.get('pair_c128_k11',{}).get('ms_per_step',0), k.get('pair_c128_k7',{}).get('ms_per_step',0), k.get('pair_c256_k11',{}).get('ms_per_step',0), k.get('block_c64_k11',{}).get('ms_per_step',0), k.get('block_c32_k11',{}).get('ms_per_step',0), k.get('block_c32_k3',{}).get('ms_per_step',0), k.get('block_c32_k7',{}).get('ms_per_step',0), k.get('block_c64_k3',{}).get('ms_per_step',0), k.get('block_c64_k7',{}).get('ms_per_step',0), k.get('block_c128_k3',{}).get('ms_per_step',0)))"
  done
done
