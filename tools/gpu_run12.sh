cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -- python3 $GRAFT_REPO_ROOT/tools/run_cfg5.py 768 200000 32 7 2 > /tmp/prof5.log 2>&1)
python tools/prof_summary.py /tmp/prof5 gpurun_out/r03_sparse_cfg5_fp16_kernel_stats_v1.csv 13 > /dev/null
head -40 gpurun_out/r03_sparse_cfg5_fp16_kernel_stats_v1.csv | cut -c1-140
