cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -2
python tools/run_cfg5.py 512 50000 16 5 0 2>&1 | grep "ms/step"
python tools/run_cfg5.py 512 50000 16 5 0 2>&1 | grep "ms/step"
