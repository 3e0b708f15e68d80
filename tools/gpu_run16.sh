cd $GRAFT_REPO_ROOT
python tools/ab_options.py "dw_blocks=768" "dw_blocks=512" "dw_blocks=640" "dw_blocks=1024" "pairs_waves=3072" "pairs_waves=3584" "dw_group=2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_ab_tune2.log
