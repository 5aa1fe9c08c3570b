cd $GRAFT_REPO_ROOT
AB_FILTER=pair_c128 bash scripts/ab.sh "" _nodma _dma8 2>&1 | tee gpurun_out/r02e/ab3.log
