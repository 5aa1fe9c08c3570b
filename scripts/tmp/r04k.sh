mkdir -p gpurun_out/r04l; O=gpurun_out/r04l
L=$GRAFT_REPO_ROOT/promonet_amd/lib
(timeout 300 python -m pytest tests -m gpu -x -q -k "stft or mel or loud or spectro or preprocess or fft" > $O/pytest_preprocess.log 2>&1; echo "rc $?" >> $O/pytest_preprocess.log); tail -2 $O/pytest_preprocess.log
for r in 1 2 3; do for v in "" _noxcd _base; do
  PROMONET_HIP_LIB=$L/libpromonet_hip$v.so timeout 200 python scripts/bench_preprocess.py 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('variant[$v] round $r:', ' '.join('%s %.1f us' % (k.replace('_abi_group',' g'), v['ms']*1e3) for k,v in r.items() if 'group' in k))" | tee -a $O/ab_fft_persistent.txt
done; done
