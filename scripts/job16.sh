cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02q
timeout 120 python scripts/rccl_sanity.py 2>&1 | tail -3
# the driver's N > 1 command shape with one rank per (the only) GPU
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 3 --warmup 1 --sustain 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
( time python bench.py > gpurun_out/r02q/bench_default.json 2> gpurun_out/r02q/bench_default.err ) 2>&1 | grep real
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r02q/bench_default.json').read().strip().splitlines()[-1])
print(r['ms_per_step'], r['dtype'], r['roofline']['frac'], r['roofline']['traffic'], r['whole_path']['measured_hbm_gbs'], r['cpu_baseline']['samples_per_s_by_threads'])
PY
