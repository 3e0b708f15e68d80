cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
URN_GRAPH=1 python tools/run_dense_cfg2.py 128 5 1 2>&1 | grep "per fwd\|graph replay"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profd -- python3 $GRAFT_REPO_ROOT/tools/run_dense_cfg2.py 128 5 1 > /tmp/profd.log 2>&1)
python tools/prof_summary.py /tmp/profd gpurun_out/r03_dense_cfg2_kernel_stats_v1.csv 5 > /dev/null
head -30 gpurun_out/r03_dense_cfg2_kernel_stats_v1.csv | cut -c1-150
