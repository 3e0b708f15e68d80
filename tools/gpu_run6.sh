cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/ab_options.py "net_fold=0" "net_fold=1" "net_fold=0" "net_fold=1" > gpurun_out/r03_ab_fold2.log 2>&1; cat gpurun_out/r03_ab_fold2.log
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/prof2.log 2>&1)
python tools/prof_summary.py /tmp/prof2 gpurun_out/r03_sparse_cfg3_kernel_stats_v2_fold.csv 26 > /dev/null
head -45 gpurun_out/r03_sparse_cfg3_kernel_stats_v2_fold.csv | cut -c1-150
