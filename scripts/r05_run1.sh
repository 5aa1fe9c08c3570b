#!/bin/bash
# round 5, GPU call 1: the new persistent-walk preprocess parity tests, the
# tightened FARGAN gates, the worker-config test, fp32 x3-skew A/B (config 2)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05a; mkdir -p $OUT; cd $ROOT
python - > $OUT/device_props.txt 2>&1 <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print(p)
for name in ('uuid', 'pci_bus_id', 'pci_device_id', 'pci_domain_id', 'gcnArchName', 'multi_processor_count'):
    print(name, getattr(p, name, 'MISSING'))
PY
PM_RECORD_ERRORS=1 timeout 1500 python -m pytest tests/test_gpu_preprocess_full.py tests/test_gpu_fargan.py -x -q -s > $OUT/pytest_new.log 2>&1; echo "pytest new rc $?" | tee -a $OUT/pytest_new.log
tail -3 $OUT/pytest_new.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "files_to_files or preprocess or spectrogram or loudness or mel" > $OUT/pytest_files.log 2>&1; echo "pytest files rc $?" | tee -a $OUT/pytest_files.log
tail -3 $OUT/pytest_files.log
for round in 1 2 3; do
  for v in "" _nox3f32; do
    PROMONET_HIP_LIB=$ROOT/promonet_amd/lib/libpromonet_hip$v.so timeout 300 python bench.py --dtype fp32 --batch 8 --seconds 5 --steps 6 --warmup 2 --sustain 0 --no-cpu-baseline --no-traffic --no-secondary 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernels']
print('fp32 config2 variant[$v] round $round: %.2f ms | ' % r['ms_per_step'] + ' '.join('%s %.3f' % (n, v['ms_per_step']) for n, v in sorted(k.items()) if 'c32' in n or 'c64' in n))" | tee -a $OUT/ab_x3skew_f32.txt
  done
done
