cd $GRAFT_REPO_ROOT
export URN_DIST_BACKEND=gloo
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03_bench_2rank_gloo.log 2>&1
tail -3 gpurun_out/r03_bench_2rank_gloo.log | cut -c1-700
