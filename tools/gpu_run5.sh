cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pairs.py tests/test_gpu_sparse.py -x -q > gpurun_out/r03_tests_fold.log 2>&1 || { tail -40 gpurun_out/r03_tests_fold.log; exit 1; }
tail -3 gpurun_out/r03_tests_fold.log
python tools/ab_options.py "net_fold=0" "net_fold=1" "net_fold=1,pairs_waves=2048" "net_fold=1,pairs_waves=3072" "net_fold=1,dw_blocks=640" "net_fold=1,dw_blocks=1024" > gpurun_out/r03_ab_fold.log 2>&1; cat gpurun_out/r03_ab_fold.log
