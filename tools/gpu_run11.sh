cd $GRAFT_REPO_ROOT
python tools/run_cfg5.py 768 200000 32 7 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_cfg5_v1.log
python tools/run_dense_cfg2.py 128 5 1 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_dense_v1.log
