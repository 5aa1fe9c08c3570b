#!/bin/bash
# On the GPU box: interleaved A/B of library variants on the three preprocess
# transforms (C ABI, batch 32 x 10 s). usage: scripts/ab_preprocess.sh "" _base ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2 3; do
  for v in "$@"; do
    PROMONET_HIP_LIB=$ROOT/promonet_amd/lib/libpromonet_hip$v.so python $ROOT/scripts/bench_preprocess.py 2>/dev/null | tail -1 | python -c "
import json,sys; r=json.loads(sys.stdin.read())
print('variant[$v] round $round: ' + ' '.join('%s g%d %.1f us' % (n, g, r['%s_abi_group%d' % (n, g)]['ms'] * 1e3) for g in (16, 32) for n in ('spectrogram', 'log_mel', 'loudness_8_bands')))"
  done
done
