# kernel-trace summary of the cfg5-shaped step (768^3, 200k voxels, uf 32, uns 7, fp16 operands)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-v2}
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -- python3 $GRAFT_REPO_ROOT/tools/run_cfg5.py 768 200000 32 7 2 > /tmp/prof5.log 2>&1)
grep "ms/step" /tmp/prof5.log
python tools/prof_summary.py /tmp/prof5 gpurun_out/r03_sparse_cfg5_fp16_kernel_stats_$TAG.csv 13 > /dev/null
head -30 gpurun_out/r03_sparse_cfg5_fp16_kernel_stats_$TAG.csv | cut -c1-150
