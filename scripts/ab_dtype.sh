#!/bin/bash
# On the GPU box: interleaved A/B of library variants at a given --dtype, 3 rounds each.
# usage: DTYPE=checkpoint [AB_FILTER=substr] scripts/ab_dtype.sh "" _suffix1 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for v in "$@"; do
    PROMONET_HIP_LIB=$ROOT/promonet_amd/lib/libpromonet_hip$v.so python $ROOT/bench.py --dtype ${DTYPE:-checkpoint} --steps 6 --warmup 2 --sustain 0 --no-cpu-baseline --no-traffic --no-secondary 2>/dev/null | AB_FILTER=$AB_FILTER python -c "
import json,sys,os; r=json.loads(sys.stdin.read()); k=r['kernels']; f=os.environ.get('AB_FILTER','')
print('variant[$v] round $round: %.2f ms | ' % r['ms_per_step'] + ' '.join('%s %.3f' % (n.replace('block_','b').replace('pair_','p'), v['ms_per_step']) for n, v in sorted(k.items()) if (f in n if f else v['ms_per_step'] > 0.6)))"
  done
done
