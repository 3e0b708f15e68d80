cd $GRAFT_REPO_ROOT
python tools/run_cfg5.py 512 50000 16 5 0 2>&1 | grep "ms/step"
URN_OPTIONS=net_side2=1 python tools/run_cfg5.py 512 50000 16 5 0 2>&1 | grep "ms/step"
python tools/run_cfg5.py 512 50000 16 5 0 2>&1 | grep "ms/step"
URN_OPTIONS=net_side2=1 python tools/run_cfg5.py 512 50000 16 5 0 2>&1 | grep "ms/step"
URN_OPTIONS=net_side2=1 python tools/run_cfg5.py 768 200000 32 7 2 2>&1 | grep "ms/step"
