cd $GRAFT_REPO_ROOT
python tools/ab_options.py "pairs_waves=2560" "pairs_waves=2048" "pairs_waves=3072" "dw_blocks=640" "dw_blocks=1024" "pairs_wgs=384" "pairs_wgs=768" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_ab_tune1.log
