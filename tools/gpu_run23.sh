cd $GRAFT_REPO_ROOT
CFG=5 PREC=fp16 python tools/ab_options.py "tiles_nbr=64" "tiles_nbr=128" "tiles_nbr=64/64/128/128/128/128/128" "tiles_nbr=64/64/64/128/128/128/128" "pairs_wgs16=128" "pairs_wgs16=512" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03_ab_cfg5_tiles.log
