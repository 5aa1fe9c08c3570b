cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02l
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_model.py -m gpu -q -x -k "block_iteration or whole or golden or ragged or small_config or variants or windows" > gpurun_out/r02l/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/r02l/pytest.log
bash scripts/ab.sh "" _b64 2>&1 | tee gpurun_out/r02l/ab_b128.log
export TMPDIR=/tmp; ROOT=$GRAFT_REPO_ROOT; cd /tmp
timeout 300 rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d $ROOT/gpurun_out/r02l/pmc_lds -o lds -- python $ROOT/bench.py --no-cpu-baseline --sustain 0 --steps 1 --warmup 0 > $ROOT/gpurun_out/r02l/pmc.log 2>&1
cd $ROOT
python - <<'PY'
import csv, re, glob
from collections import defaultdict
s=defaultdict(lambda: defaultdict(float))
for f in glob.glob('gpurun_out/r02l/pmc_lds/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:70]
        s[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in s.items():
    if v.get('SQ_LDS_IDX_ACTIVE',0)>1e6: print('%-72s conflict/active %.3f' % (k, v['SQ_LDS_BANK_CONFLICT']/v['SQ_LDS_IDX_ACTIVE']))
PY
find gpurun_out/r02l -name "*.csv" -size +1M -delete
