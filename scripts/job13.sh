cd $GRAFT_REPO_ROOT
AB_FILTER=block_c64_k3 bash scripts/ab.sh "" _stag3 _stag6 2>&1 | tee gpurun_out/ab_stagger.log
