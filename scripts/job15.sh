cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "" _fg6 _fg8; do
  PROMONET_HIP_LIB=$GRAFT_REPO_ROOT/promonet_amd/lib/libpromonet_hip$v.so timeout 200 python bench.py --model fargan --steps 4 --warmup 1 --sustain 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('fargan variant[$v] round $round: %.2f ms  %.2f us/step' % (r['ms_per_step'], r['roofline']['latency_model']['us_per_step']))"
done
done
