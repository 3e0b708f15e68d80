cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pairs.py -x -q > gpurun_out/r03_pairs_tests.log 2>&1 || { tail -30 gpurun_out/r03_pairs_tests.log; exit 1; }
tail -3 gpurun_out/r03_pairs_tests.log
python tools/bench_pairs.py v3 > gpurun_out/r03_bench_pairs_v3.log 2>&1; cat gpurun_out/r03_bench_pairs_v3.log
python tools/ab_options.py "pairs_v3=0" "pairs_v3=382" "pairs_v3=6" "pairs_v3=30" > gpurun_out/r03_ab_v3.log 2>&1; cat gpurun_out/r03_ab_v3.log
