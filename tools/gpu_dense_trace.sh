# kernel trace of the dense cfg2 step (bf16 operands), one step listed launch by launch
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-v1}
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profd -- python3 $GRAFT_REPO_ROOT/tools/run_dense_cfg2.py 128 5 1 > /tmp/profd.log 2>&1)
tail -3 /tmp/profd.log
python tools/prof_summary.py /tmp/profd gpurun_out/r03_dense_cfg2_kernel_stats_$TAG.csv 5 > /dev/null
python tools/step_list_by.py /tmp/profd k_dce_fwd > gpurun_out/r03_dense_step_list_$TAG.txt 2>&1
tail -45 gpurun_out/r03_dense_step_list_$TAG.txt
