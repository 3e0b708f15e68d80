cd $GRAFT_REPO_ROOT
python tools/host_path.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_host_path_v1.log
python tools/host_lead.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_host_lead_v1.log
