set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_gputest_v1.log 2>&1 || { tail -40 gpurun_out/r03_gputest_v1.log; exit 1; }
tail -5 gpurun_out/r03_gputest_v1.log
timeout -k 10 300 python tools/nccl_one_rank.py > gpurun_out/r03_nccl_one_rank.log 2>&1 || true
cat gpurun_out/r03_nccl_one_rank.log | tail -8
export TMPDIR=/tmp
cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/qtrace -- python3 $GRAFT_REPO_ROOT/tools/many_executors.py > $GRAFT_REPO_ROOT/gpurun_out/r03_many_executors.log 2>&1 || true
cd $GRAFT_REPO_ROOT
tail -8 gpurun_out/r03_many_executors.log
python tools/queue_table.py /tmp/qtrace 25 > gpurun_out/r03_queue_table.txt 2>&1 || true
cat gpurun_out/r03_queue_table.txt | tail -60
