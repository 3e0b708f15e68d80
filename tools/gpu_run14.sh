cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest_v2.log 2>&1 || { tail -40 gpurun_out/r03_gputest_v2.log; exit 1; }
tail -2 gpurun_out/r03_gputest_v2.log
python tools/many_executors.py 2>&1 | grep "executor " | tee gpurun_out/r03_many_executors_v2.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
