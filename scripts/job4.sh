cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 900 python -m pytest tests/test_gpu_edit.py tests/test_gpu_fargan.py -m gpu -q -s > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc $?"; tail -5 gpurun_out/r02d/pytest.log
for round in 1 2; do
for v in "" _fg2 _fg6; do
  PROMONET_HIP_LIB=$GRAFT_REPO_ROOT/promonet_amd/lib/libpromonet_hip$v.so timeout 200 python bench.py --model fargan --dtype fp32 --steps 4 --warmup 1 --sustain 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('fargan variant[$v] round $round: %.2f ms  %.2f us/step' % (r['ms_per_step'], r['roofline']['latency_model']['us_per_step']))"
done
done
for b in 64 128; do
  timeout 200 python bench.py --model fargan --dtype fp32 --batch $b --steps 3 --warmup 1 --sustain 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('fargan batch $b: %.2f ms  rtf %.0f' % (r['ms_per_step'], r['rtf']))"
done
timeout 200 python bench.py --model fargan --dtype f16 --steps 3 --warmup 1 --sustain 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('fargan f16 weights batch 32: %.2f ms' % (r['ms_per_step']))"
