cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
