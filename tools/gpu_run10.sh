cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_distributed.py tests/test_gpu_sparse.py -x -q -k "distributed or trainval or two_rank" > gpurun_out/r03_tests_overlap.log 2>&1 || { tail -40 gpurun_out/r03_tests_overlap.log; exit 1; }
tail -3 gpurun_out/r03_tests_overlap.log
timeout -k 10 300 python tools/nccl_one_rank.py 2>&1 | grep "ms per step" | tee gpurun_out/r03_nccl_one_rank_v2.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
