cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r03_bench_v1.json.log 2>&1; tail -1 gpurun_out/r03_bench_v1.json.log | cut -c1-400
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/prof1.log 2>&1)
python tools/prof_summary.py /tmp/prof1 gpurun_out/r03_sparse_cfg3_kernel_stats_v1.csv 26 > /dev/null
python tools/one_step_list.py /tmp/prof1 > gpurun_out/r03_step_list_v1.txt 2>&1
python tools/trace_gaps.py /tmp/prof1 > gpurun_out/r03_gaps_v1.txt 2>&1; cat gpurun_out/r03_gaps_v1.txt
