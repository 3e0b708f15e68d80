#!/usr/bin/env python
"""bench.py -- active-voxels/sec of one sparse U-ResNet training step (forward + loss +
backward + gradient all-reduce + Adam step) on N MI355X GPUs of one node.

Workload at N=1: BASELINE.json configs[2]: uresnet_sparse -dd 3 -ss 512, one event of
50,000 active voxels, -nc 5 -uf 16 -uns 5, fp32.  Weak scaling: every rank processes its
own event(s) (rank r uses generator seeds r*E .. r*E+E-1), so whole-job voxels grow with N.
Inputs are synthetic (SURVEY App. D generator) and resident in HBM before timing starts.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline_hbm -- the integer phase (hash / unique / rulebook kernels) against the HBM roofline, same method;
  roofline     -- dominant kernel (gather-conv MFMA kernel, forward + input-gradient launches):
                  algorithmic FLOPs (2*R*Cin*Cout per launch, R = rules of that launch) divided by
                  the kernel's launch durations measured with HIP events on the launch stream;
  cpu_baseline -- an SCN-style fp32 CPU restatement (oracle/cpu_port.py + oracle/cpu_fast.c, a port: the reference's own
                  sparse path cannot run without sparseconvnet): the gather convolutions in C with OpenMP on the host
                  cores (thread count = fastest of a sweep up to all of them), forward + backward of the same event,
                  2 warm-ups + median of 5.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
SPATIAL, VOXELS, FILTERS, STRIDES, NCLASS = 512, 50000, 16, 5, 5


def conv_launch_flops(model, geo):
    """Algorithmic FLOPs of every gather-conv MFMA launch (forward + dX) of one step."""
    from uresnet_pytorch_amd import scn
    total, launches = 0.0, 0

    def rules(mod, level_in):
        if isinstance(mod, scn.SubmanifoldConvolution):
            return geo.rules[level_in]
        if isinstance(mod, scn.NetworkInNetwork):
            return geo.n[level_in]
        return None

    # walk the module tree with the level each module runs at
    def walk(mod, level):
        nonlocal total, launches
        if isinstance(mod, (scn.SubmanifoldConvolution, scn.NetworkInNetwork)):
            cin, cout = mod.nIn, mod.nOut
            if cin % 16 == 0 and cout % 16 == 0:
                total += 2 * (2.0 * rules(mod, level) * cin * cout); launches += 2   # fwd + dX
            return level
        if isinstance(mod, scn.Convolution):
            total += 2 * (2.0 * geo.n[level] * mod.nIn * mod.nOut); launches += 2
            return level + 1
        if isinstance(mod, scn.Deconvolution):
            total += 2 * (2.0 * geo.n[level - 1] * mod.nIn * mod.nOut); launches += 2
            return level - 1
        if isinstance(mod, scn.ConcatTable):
            out = level
            for c in mod.children():
                out = walk(c, level)
            return out
        if isinstance(mod, torch.nn.Sequential):
            for c in mod.children():
                level = walk(c, level)
            return level
        return level

    walk(model.sparseModel, 0)
    return total, launches


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--events-per-gpu', type=int, default=1)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from types import SimpleNamespace
    from uresnet_pytorch_amd import lib as urn_lib
    from uresnet_pytorch_amd import parallel
    from uresnet_pytorch_amd.iotools.synthetic import make_sparse_blob
    from uresnet_pytorch_amd.models import SparseUResNet, SparseSegmentationLoss

    rank, world, local_rank = parallel.init_distributed()
    assert world == args.gpus, 'launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world)
    assert torch.cuda.is_available(), 'bench.py needs a GPU'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    L = urn_lib.load()

    flags = SimpleNamespace(DATA_DIM=3, URESNET_FILTERS=FILTERS, URESNET_NUM_STRIDES=STRIDES, SPATIAL_SIZE=SPATIAL,
                            NUM_CLASS=NCLASS)
    torch.manual_seed(0)
    model = SparseUResNet(flags).to(dev).train()
    parallel.broadcast_parameters(model)
    crit = SparseSegmentationLoss(flags)
    E = args.events_per_gpu
    blob = make_sparse_blob([rank * E + e for e in range(E)], SPATIAL, VOXELS)
    data = torch.from_numpy(blob['data']).to(dev)
    label = torch.from_numpy(blob['label']).to(dev)
    grads = parallel.FlatGradients(model)
    opt = parallel.FlatAdam(grads, lr=1e-3)   # torch.optim.Adam semantics, one pass over the flat buffers
    voxels_per_rank = int(data.shape[0])

    # gradient all-reduce (SUM): the decoder + bottom + head suffix of the flat buffer from inside the backward pass, overlapped
    # with its encoder half, the prefix behind it (parallel.OverlappedAllReduce; nothing to reduce at one rank)
    overlap = parallel.OverlappedAllReduce(grads)

    seed = torch.ones((), device=dev)          # d loss / d loss, as trainval.backward hands it over (no ones-fill launch per step)

    def step():
        grads.zero()
        out = model(data)
        loss, _ = crit(out, [data], [label], None)
        overlap.arm(model)
        loss.backward(seed)
        overlap.finish()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    # Python's cyclic collector: a full (generation-2) pass over the ~10^6 objects that importing torch leaves behind takes
    # ~70 ms and comes once every ~250 steps (tools/hiccup.py) -- 25 steps' worth of GPU time.  Everything alive after the
    # warm-up is long-lived: moved to the permanent generation, later passes only look at what the steps allocate.
    import gc
    gc.collect()
    gc.freeze()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    ms_per_step = 1e3 * dt / args.steps
    total_voxels = voxels_per_rank * world
    value = total_voxels * args.steps / dt

    result = None
    # roofline leg: instrumented steps after the timed region (HIP events on the launch stream).  EVERY rank runs
    # them (the step contains a collective); only rank 0 records.
    PSTEPS = 3
    if rank == 0:
        urn_lib.check(L.urn_prof_enable(1))
    for _ in range(PSTEPS):
        step()
    torch.cuda.synchronize()
    if rank == 0:
        from uresnet_pytorch_amd import sparse_ops as so
        ms = ctypes.c_double(); n = ctypes.c_int64()
        urn_lib.check(L.urn_prof_read(0, ctypes.byref(ms), ctypes.byref(n)))
        ims = ctypes.c_double(); inn = ctypes.c_int64()
        urn_lib.check(L.urn_prof_read(2, ctypes.byref(ims), ctypes.byref(inn)))
        urn_lib.check(L.urn_prof_enable(0))
        geo = so.SparseGeometry(data[:, :4].to(torch.int32), SPATIAL, STRIDES)
        flops_step, launches_step = conv_launch_flops(model, geo)
        # integer phase (site hash + unique of every level, strided tables, 27-probe rulebook of every level): HBM-bound.
        # Algorithmic bytes per SURVEY 8(d): 4*(d+1)*N coordinates + 16*N hash slot traffic + 4*27*N table, per level.
        int_bytes = sum((4 * 4 + 16 + 4 * 27) * int(nl) for nl in geo.n)
        int_gbs = int_bytes * PSTEPS / (ims.value * 1e-3) / 1e9 if ims.value > 0 else 0.0
        roofline_hbm = {
            'bound': 'hbm', 'kernel': 'integer phase: k_insert_lv/k_flag_count_lv/k_assign_lv/k_links_lv (all levels per launch) + '
                                      'k_rulebook_subm_multi (%d library calls per step)' % (inn.value // PSTEPS),
            'achieved': round(int_gbs, 1), 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': round(int_gbs / PEAK_HBM_GBS, 5),
            'traffic': None, 'algorithmic_mb_per_step': round(int_bytes / 1e6, 2),
            'us_per_step': round(1e3 * ims.value / PSTEPS, 1),
        }
        achieved = (flops_step * PSTEPS) / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
        # HBM traffic of the same kernel: measured in separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) and
        # committed under profiles/ (bench.py cannot read PMC counters itself); null when no measurement is committed
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'r03_pmc_traffic.json')
        if os.path.exists(tpath):
            traffic = round(json.load(open(tpath)).get('gconv_traffic_bytes_per_launch', 0.0)) or None
        roofline = {
            'bound': 'mfma', 'kernel': 'k_gconv_pairs<KC,NC,XF,DEEP,PREC> (compacted rule lists) / k_gconv_tile<KS,RB,CB> by shape: all gather-conv forward + input-gradient launches',
            'achieved': round(achieved, 3), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 5), 'traffic': traffic,
            'traffic_unit': 'bytes per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/r03_pmc_traffic.json)',
            'launches_per_step': int(n.value // PSTEPS), 'avg_launch_us': round(1e3 * ms.value / max(n.value, 1), 2),
            'algorithmic_gflop_per_step': round(flops_step / 1e9, 3),
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:     # the CPU leg is reported at N=1 only
            # SCN-style fp32 CPU path (per offset: gather -> sgemm -> scatter-add; oracle/cpu_port.py, checked against the
            # oracle in tests/test_oracle_sparse.py) on all host threads: 2 warm-ups, median of 5 forward+backward passes
            # of the SAME event.  (The fp64-accumulating checker oracle/sparse_ref.c is ~10x slower and is not a baseline.)
            from oracle import cpu_port
            P = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if 'running' not in k}
            med, ts, threads = cpu_port.time_step(P, FILTERS, STRIDES, NCLASS, SPATIAL, blob['data'], blob['label'],
                                                  warmup=2, repeats=5, threads='auto')
            omp = getattr(cpu_port.time_step, 'last_kernel', 'torch') == 'omp'
            cpu = {'value': round(voxels_per_rank / med, 1), 'unit': 'active-voxels/s', 'cores': threads, 'kind': 'port',
                   'sample': '%d-voxel event(s) of the bench, forward+backward, 2 warm-ups + median of 5 (%.3f s; all: %s)'
                             % (voxels_per_rank, med, ' '.join('%.3f' % t for t in ts)),
                   'method': ('fp32 gather convolution in C with OpenMP (oracle/cpu_fast.c: one parallel loop over tiles of 64 output '
                              'rows, offset by offset inside a tile; private weight-gradient copies per thread), BatchNorm / head / '
                              'loss as torch CPU ops, the oracle rulebook; ' if omp else
                              'fp32 per-offset gather -> torch.mm -> scatter-add with the oracle rulebook, torch CPU autograd; ') +
                             'thread count = fastest of a one-pass sweep: ' +
                             ', '.join('%d thr %.2f s' % kv for kv in sorted(cpu_port.time_step.last_sweep.items())),
                   'host_cpus': os.cpu_count()}
        result = {
            'metric': 'active-voxels/sec fwd+bwd, 512^3 sparse 5-class U-ResNet', 'value': round(value, 1),
            'unit': 'active-voxels/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'uresnet_sparse -dd 3 -ss 512 -nc 5 -uf 16 -uns 5, %d event(s)/GPU x %d active voxels, '
                                   'fp32, step = fwd+loss+bwd+grad all-reduce(SUM)+Adam' % (E, VOXELS),
                       'events_per_gpu': E, 'voxels_per_event': VOXELS, 'parallelism': 'dp%d (events sharded, '
                       'gradient all-reduce in two pieces, the first overlapped with the backward pass)' % world},
            'roofline': roofline, 'roofline_hbm': roofline_hbm, 'cpu_baseline': cpu,
        }
    if world > 1:
        dist.barrier()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
