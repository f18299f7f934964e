#!/usr/bin/env python
"""bench.py -- cost-volume Mvoxels/s of the fused RPC warp + variance build on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): cost-volume Mvoxels/s = B*D*H*W / t for the fused warp + variance build,
3-view, 768x384 planes, 64 height hypotheses, C=32 feature channels (configs[1], the shape the
metric is quoted on).  One "step" = one pass of the hot path over one synthetic tile = ONE launch
of smvs_rpc_costvol_fwd that writes the whole (1,32,64,384,768) variance volume, inputs already
resident in HBM.

Multi-GPU (one process per GPU): the height-hypothesis axis is sharded -- rank r builds planes
[64 r, 64 (r+1)) of a 64*N-hypothesis sweep over the same tile (per-GPU work fixed => "weak").
The build has no exchange step; the only exchange on the path is the all-reduce of the (3,H,W)
float64 regression partials after the regulariser (satmvs_amd/shard.py), which is exercised and
timed once outside the timed region and reported as "exchange".

The JSON line also carries
  roofline     HBM roofline of the dominant kernel: algorithmic bytes (138.25 B/voxel: 4*C write +
               4 height read + 4*C*V/D feature reads) / mean launch time from HIP events
               recorded on the launch stream, against the 8 TB/s HBM3E peak.
  cpu_baseline the CPU oracle (oracle/oracle.c, OpenMP, all host cores) timed on a bounded sample
               of the same workload -- a reported baseline only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
WORKLOADS = {
    # name: (V, C, D, H, W)
    "cfg2_rpc_3view_768x384x64_c32": (3, 32, 64, 384, 768),
    "stage3_rpc_3view_768x384x8_c8": (3, 8, 8, 384, 768),
    "cfg4_rpc_5view_1536x768x8_c32": (5, 32, 8, 768, 1536),
}


def algorithmic_bytes_per_voxel(V, C, D):
    return 4.0 * C + 4.0 + 4.0 * C * V / D


def make_inputs(V, C, D_local, D_total, d_off, H, W, dev):
    """Deterministic synthetic tile: seeded TLC-shaped RPCs, randn features,
    heights linspace(0,400,D_total) broadcast to per-voxel (B,D,H,W) float32."""
    from satmvs_amd import rpc_synth
    rpc = torch.from_numpy(rpc_synth.make_view_rpcs(V, H, W, seed=0)[None]).to(dev)
    g = torch.Generator(device="cpu").manual_seed(0)
    feats = [torch.randn((1, C, H, W), generator=g, dtype=torch.float32).to(dev) for _ in range(V)]
    planes = torch.linspace(0.0, 400.0, D_total, dtype=torch.float32)[d_off:d_off + D_local]
    depth = planes.view(1, D_local, 1, 1).expand(1, D_local, H, W).contiguous().to(dev)
    return feats, rpc, depth


def cpu_baseline(V, C, D, H, W, budget_s=12.0):
    """Time the oracle on a bounded sample: whole passes over the same workload until ~budget_s have elapsed."""
    from oracle import oracle as orc
    from satmvs_amd import rpc_synth
    orc.build()
    rng = np.random.default_rng(0)
    feats = [rng.standard_normal((1, C, H, W)).astype(np.float32) for _ in range(V)]
    rpc = rpc_synth.make_view_rpcs(V, H, W, seed=0)[None]
    depth = np.ascontiguousarray(np.broadcast_to(np.linspace(0, 400, D, dtype=np.float32).reshape(1, D, 1, 1), (1, D, H, W)))
    out = np.zeros((1, C, D, H, W), np.float32)
    orc.costvol_variance(feats, rpc, depth, "rpc", d_begin=0, d_end=D, out=out)      # warm-up (threads, page faults)
    passes, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < budget_s and passes < 1000:
        orc.costvol_variance(feats, rpc, depth, "rpc", d_begin=0, d_end=D, out=out)
        passes += 1
        dt = time.perf_counter() - t0
    return {"value": round(passes * D * H * W / dt / 1e6, 3), "unit": "Mvox/s", "cores": orc.num_threads(),
            "kind": "port",
            "sample": "%d passes over the same %dx%dx%d tile (V=%d,C=%d) in %.1f s, oracle/oracle.c with OpenMP" % (
                passes, W, H, D, V, C, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="cfg2_rpc_3view_768x384x64_c32", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    # SMVS_BENCH_BACKEND=gloo + SMVS_BENCH_ONE_DEVICE=1: rehearsal of the multi-rank code path on a box
    # with a single GPU (all ranks share cuda:0, collectives staged through the host).  Not a measurement.
    one_device = os.environ.get("SMVS_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("SMVS_BENCH_BACKEND", "nccl"),    # "nccl" is RCCL on ROCm
                                rank=rank, world_size=world)

    from satmvs_amd import _lib
    _lib.load()
    V, C, D, H, W = WORKLOADS[args.workload]
    D_total = D * world
    feats, rpc, depth = make_inputs(V, C, D, D_total, rank * D, H, W, dev)
    out = torch.empty((1, C, D, H, W), dtype=torch.float32, device=dev)
    srcs = _lib.ptr_array(feats[1:])
    stream = _lib.current_stream(dev)

    def step():
        _lib.call("smvs_rpc_costvol_fwd", _lib.ptr(feats[0]), srcs, V - 1, _lib.ptr(rpc), _lib.ptr(depth), 1,
                  _lib.ptr(out), 1, C, D, H, W, 0, D, D, 0, stream)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for a, b in ev:                      # events live on the launch stream (torch's current stream)
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    exchange = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        from satmvs_amd import shard
        state = torch.rand((3, 1, H, W), dtype=torch.float64, device=dev)
        shard.allreduce_regression_state(state)             # warm-up (communicator setup)
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        shard.allreduce_regression_state(state)
        torch.cuda.synchronize()
        exchange = {"op": "all_reduce(sum,sum,max) of (3,1,%d,%d) f64 regression partials" % (H, W),
                    "bytes": int(state.numel() * 8), "ms": round((time.perf_counter() - t1) * 1e3, 3)}

    if rank == 0:
        vox_per_step = D * H * W * world
        ms_per_step = elapsed / args.steps * 1e3
        bpv = algorithmic_bytes_per_voxel(V, C, D)
        achieved = bpv * D * H * W / (kern_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(args.workload)
            except Exception:
                traffic = None
        line = {
            "metric": "cost-volume Mvoxels/s (fused RPC warp + variance build)",
            "value": round(vox_per_step / (elapsed / args.steps) / 1e6, 1),
            "unit": "Mvox/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 features / f64 RPC geometry", "data": "synthetic",
            "config": {"workload": args.workload, "views": V, "channels": C, "planes_per_gpu": D,
                       "planes_total": D_total, "H": H, "W": W, "depth_values": "per-voxel (B,D,H,W)",
                       "sharding": "height planes, %d per GPU" % D},
            "roofline": {"bound": "hbm", "kernel": "%s<rpc,%d,%d>" % (
                             "costvol_fwd_kernel" if (os.environ.get("SMVS_COSTVOL_KERNEL", "").startswith("di")
                                                      or V - 1 > 2 or C not in (16, 32))      # dispatch rule of costvol.hip
                             else "costvol_dma_kernel", V - 1, C),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "bytes_per_voxel": bpv, "kernel_ms": round(kern_ms, 4)},
        }
        if exchange is not None:
            line["exchange"] = exchange
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(V, C, D, H, W)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
