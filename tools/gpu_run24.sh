cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests/test_gpu_pairs.py -x -q 2>&1 | tail -1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-260
python tools/bench_pairs.py auto 2>&1 | grep "nbr\|chd" | cut -c1-120
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmcf -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/pmcf.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmcw -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/pmcw.log 2>&1)
python tools/pmc_summary.py /tmp/pmcf /tmp/pmcw gpurun_out/r03_pmc_traffic_v4.json | tail -3
