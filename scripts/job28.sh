#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ac
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02ac/pytest.log 2>&1; echo "pytest rc $?"; tail -6 gpurun_out/r02ac/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
bash scripts/gpu_round.sh r02ac bench
bash scripts/profile_gpu.sh r02_final > gpurun_out/r02ac/profile.log 2>&1; tail -3 gpurun_out/r02ac/profile.log
