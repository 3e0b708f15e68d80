cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sparse.py tests/test_gpu_distributed.py -x -q > gpurun_out/r03_tests_head.log 2>&1 || { tail -40 gpurun_out/r03_tests_head.log; exit 1; }
tail -3 gpurun_out/r03_tests_head.log
python tools/ab_head.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_ab_head.log
python __graft_entry__.py smoke 2>&1 | tail -2
