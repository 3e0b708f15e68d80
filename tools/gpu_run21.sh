cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dense.py tests/test_dense_golden.py -x -q > gpurun_out/r03_tests_dense.log 2>&1 || { tail -40 gpurun_out/r03_tests_dense.log; exit 1; }
tail -2 gpurun_out/r03_tests_dense.log
python tools/run_dense_cfg2.py 128 5 1 2>&1 | grep "per fwd"
URN_GRAPH=1 python tools/run_dense_cfg2.py 128 5 1 2>&1 | grep "per fwd"
