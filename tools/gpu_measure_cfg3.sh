# Round measurement of the headline config: bench line, kernel-trace summary of the same command, two --pmc passes (traffic)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-v3}
python bench.py > gpurun_out/r03_bench_$TAG.json.log 2>gpurun_out/r03_bench_$TAG.err; tail -1 gpurun_out/r03_bench_$TAG.json.log | cut -c1-1500
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profk -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /tmp/profk.log 2>&1)
python tools/prof_summary.py /tmp/profk gpurun_out/r03_sparse_cfg3_kernel_stats_$TAG.csv 26 > /dev/null
python tools/one_step_list.py /tmp/profk > gpurun_out/r03_step_list_$TAG.txt 2>&1
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/pmcf.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/pmcw.log 2>&1)
python tools/pmc_summary.py /tmp/pmcf /tmp/pmcw gpurun_out/r03_pmc_traffic_$TAG.json | tail -4
head -12 gpurun_out/r03_sparse_cfg3_kernel_stats_$TAG.csv | cut -c1-160
