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

Multi-GPU (one process per GPU): the height-hypothesis axis of the SAME tile is sharded -- rank r
builds planes plane_range(64, r, N) with replicated features / RPCs (fixed total work => "strong"),
and every step ends with the path's one exchange, the all-reduce of the (3,1,H,W) float64 regression
partials (satmvs_amd/shard.py), inside the timed region; its stand-alone time is reported as
"exchange".  `value` is always the whole tile's voxels / step time.

Before anything is timed the kernel is launched for --prewarm-seconds of wall clock (default 0.6 s)
so that a short --warmup still measures steady clocks.  "extra" carries short runs of the other
BASELINE shapes (cfg4 5-view shard, stage-3 C=8 planes, cfg5 homography volume, cfg3 cascade forward).

The JSON line also carries
  roofline     HBM roofline of the dominant kernel: algorithmic bytes (138.25 B/voxel: 4*C write +
               4 height read + 4*C*V/D feature reads) / mean launch time from HIP events
               recorded on the launch stream, against the 8 TB/s HBM3E peak.
  cpu_baseline the CPU oracle (oracle/oracle.c, OpenMP, all host cores) timed on a bounded sample
               of the same workload -- a reported baseline only.
  cpu_baseline_torch  the reference's torch device=cpu operator sequence (oracle/torch_composite.py) on a
               bounded sample of planes of the same tile -- what north_star calls "the reference's CPU path".
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
# height hypotheses of the side workloads (metres): the planes a real run would hand to that launch
SIDE_HEIGHTS = {
    "stage3_rpc_3view_768x384x8_c8": (190.0, 207.5),        # stage 3 of the cascade: 8 planes 2.5 m apart around the surface
    "cfg4_rpc_5view_1536x768x8_c32": (0.0, 400.0 * 7 / 63),  # one GPU's shard: the first 8 of 64 planes over 0..400 m
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


def cpu_baseline_torch(V, C, D, H, W, budget_s=8.0):
    """The reference's CPU (torch, device=cpu) path: its operator sequence as a stock-PyTorch composite
    (oracle/torch_composite.py, bit-identical to the reference on the golden fixture), plane at a time like the pred
    loop, on a bounded sample of planes of the same tile with all host cores."""
    from oracle import torch_composite as tc
    from satmvs_amd import rpc_synth
    # intra-op threads: the reference's small element-wise ops stop scaling early (GPU-box host, s per plane: 8-32 threads
    # 0.30, 64 threads 0.52, 128 threads 1.0, 256 threads 34) -- time it where it is fastest
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    g = torch.Generator(device="cpu").manual_seed(0)
    feats = [torch.randn((1, C, H, W), generator=g, dtype=torch.float32) for _ in range(V)]
    rpc = torch.from_numpy(rpc_synth.make_view_rpcs(V, H, W, seed=0)[None])
    depth = torch.linspace(0.0, 400.0, D, dtype=torch.float32).view(1, D, 1, 1).expand(1, D, H, W).contiguous()
    tc.variance_planes(feats, rpc, depth, 0, 1)                                   # warm-up (thread pool, allocator)
    planes, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < budget_s and planes < D:
        tc.variance_planes(feats, rpc, depth, planes, planes + 1)
        planes += 1
        dt = time.perf_counter() - t0
    return {"value": round(planes * H * W / dt / 1e6, 3), "unit": "Mvox/s", "cores": torch.get_num_threads(), "kind": "torch",
            "sample": "%d of the %d planes of the same %dx%d tile (V=%d,C=%d), plane at a time, in %.1f s; "
                      "oracle/torch_composite.py = the reference's torch operator sequence on device=cpu" % (planes, D, W, H, V, C, dt)}


def kernel_source_hash():
    """sha256 over the sources of the dominant kernel: ties profiles/pmc_traffic.json to the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    for f in ("costvol.hip", "costvol_fused.hip", "costvol_kernels.h", "smvs_device.h"):
        with open(os.path.join(ROOT, "satmvs_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(workload):
    """HBM bytes per launch from the committed PMC passes -- only if they were taken on the current kernel sources."""
    tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(tf)).get(workload)
        if isinstance(rec, dict) and rec.get("source_sha256") == kernel_source_hash():
            return int(rec["bytes"])
    except Exception:
        pass
    return None


def tile_build(_lib, feats, rpc, depth, out, V, C, D, H, W, stream, plane_coef=True):
    """(step, kernel, fold) of one tile through the C ABI.  step = what a caller issues per tile: smvs_rpc_plane_coef (folds the
    source cubics at every plane's height: one tiny launch) + smvs_rpc_costvol_fwd_pc (the build; its waves take the bivariate
    cubics where their heights are their planes', the trivariate ones elsewhere).  kernel = the build alone on already folded
    coefficients (the dominant kernel, what the roofline is quoted on).  plane_coef=False: smvs_rpc_costvol_fwd, the
    trivariate chain for every voxel (round 5's path)."""
    srcs = _lib.ptr_array(feats[1:])
    keep = [srcs]
    if not plane_coef:
        def kernel():
            _lib.call("smvs_rpc_costvol_fwd", _lib.ptr(feats[0]), srcs, V - 1, _lib.ptr(rpc), _lib.ptr(depth), 1,
                      _lib.ptr(out), 1, C, D, H, W, 0, D, D, 0, stream)
        kernel.keep = keep
        return kernel, kernel, (lambda: None)
    pc = torch.empty(_lib.load().smvs_rpc_plane_coef_bytes(1, V - 1, D) // 8, dtype=torch.float64, device=out.device)
    keep.append(pc)

    def fold():
        _lib.call("smvs_rpc_plane_coef", _lib.ptr(rpc), _lib.ptr(depth), 1, _lib.ptr(pc), 1, V - 1, D, H, W, 0, D, stream)

    def kernel():
        _lib.call("smvs_rpc_costvol_fwd_pc", _lib.ptr(feats[0]), srcs, V - 1, _lib.ptr(rpc), _lib.ptr(depth), 1, _lib.ptr(pc),
                  _lib.ptr(out), 1, C, D, H, W, 0, D, D, 0, stream)

    def step():
        fold()
        kernel()
    step.keep = keep
    fold()
    return step, kernel, fold


def prewarm(step, seconds):
    """Untimed launches for `seconds` of wall clock: the timed region then runs at steady (power-limited) clocks
    whatever --warmup says; a cold MI355X ramps for ~0.2 s and would charge that to the first launches."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            step()
        torch.cuda.synchronize()


def time_steps(step, steps, barrier=lambda: None):
    """(wall seconds for `steps` launches, mean HIP-event ms of one launch); events live on the launch stream."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for a, b in ev:
        a.record()
        step()
        b.record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    return elapsed, float(np.mean([a.elapsed_time(b) for a, b in ev]))


def rank_identities(dist, dev, world, rank, one_device):
    """Evidence that the collective really spanned `world` devices: every rank's (rank, device index, PCI bus id, device name,
    host pid) all-gathered through the process group itself, the backend's name and -- for "nccl" (= RCCL on ROCm) -- the
    library version.  Asserts the bus ids are distinct unless this is the one-GPU rehearsal (SMVS_BENCH_ONE_DEVICE=1)."""
    props = torch.cuda.get_device_properties(dev)
    try:
        bus = torch.cuda.get_device_properties(dev).pci_bus_id
        bus = "%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), bus, getattr(props, "pci_device_id", 0))
    except AttributeError:
        bus = "unknown"
    uuid = str(getattr(props, "uuid", ""))
    mine = {"rank": rank, "device_index": dev.index, "pci_bus_id": bus, "uuid": uuid, "name": props.name, "pid": os.getpid()}
    if dist is None:
        return {"backend": None, "ranks": [mine]}
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    backend = dist.get_backend()
    info = {"backend": backend, "ranks": everyone, "distinct_devices": len({(r["pci_bus_id"], r["uuid"]) for r in everyone})}
    if backend == "nccl":
        try:
            info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:                                  # noqa: BLE001 -- reported, never fatal
            info["rccl_version"] = "unavailable (%s)" % type(e).__name__
    if not one_device and info["distinct_devices"] != world:
        raise SystemExit("bench.py --gpus %d: the ranks report only %d distinct devices: %r" % (world, info["distinct_devices"], everyone))
    return info


def kernel_name(V, C, planes=4):
    direct = C not in (8, 16, 32)                           # dispatch rule of costvol.hip (launch_ct)
    if direct:
        return "costvol_fwd_kernel<rpc,%d,%d>" % (V - 1, C)
    dp = 1 if planes == 1 else 2 if (planes == 2 or V - 1 > 4) else 8 if (planes % 8 == 0 and V - 1 <= 2 and C == 32) else 4
    from satmvs_amd import _lib
    if 3 <= V - 1 <= 4 and planes % 8 == 0:                 # shared-box form: 2 x 2 waves, one box per source for 4 rows x 8 planes
        return "costvol_dma_kernel<rpc,%d,%d,4,%s,shared 2x2>" % (V - 1, C, _lib.get_arith())
    return "costvol_dma_kernel<rpc,%d,%d,%d,%s>" % (V - 1, C, dp, _lib.get_arith())


def side_workloads(dev, stream):
    """Other BASELINE shapes, a few launches each (reported under "extra", never as `value`): the 5-view 1536x768 shard of
    cfg4, a stage-3 C=8 plane group, the cfg5 homography volume, and one cfg3 inference cascade forward."""
    from satmvs_amd import _lib
    extra = {}
    for name in ("cfg4_rpc_5view_1536x768x8_c32", "stage3_rpc_3view_768x384x8_c8"):
        V, C, D, H, W = WORKLOADS[name]
        feats, rpc, depth = make_inputs(V, C, D, D, 0, H, W, dev)
        lo, hi = SIDE_HEIGHTS[name]
        depth = torch.linspace(lo, hi, D, dtype=torch.float32).view(1, D, 1, 1).expand(1, D, H, W).contiguous().to(dev)
        out = torch.empty((1, C, D, H, W), dtype=torch.float32, device=dev)
        step, kernel, _ = tile_build(_lib, feats, rpc, depth, out, V, C, D, H, W, stream)
        prewarm(step, 0.3)                                      # like the headline: the inputs were just built on the host, the clocks have dropped
        _, ms_step = time_steps(step, 30)
        _, ms = time_steps(kernel, 30)
        bpv = algorithmic_bytes_per_voxel(V, C, D)
        extra[name] = {"kernel": kernel_name(V, C, D), "ms": round(ms, 4), "ms_with_fold": round(ms_step, 4), "prewarm_seconds": 0.3,
                       "Mvox/s": round(D * H * W / ms_step / 1e3, 1),
                       "roofline_frac": round(bpv * D * H * W / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "note": "ms / roofline_frac: the build kernel on folded plane coefficients; ms_with_fold / Mvox/s: smvs_rpc_plane_coef + the build, as a caller issues them"}
        del feats, out, step, kernel
    # cfg2 with per-pixel jittered hypotheses (SURVEY 8d's second height variant: what cascade stages 2-3 hand over --
    # plane + N(0, 2 m) per pixel, wider tap boxes than plane-constant heights)
    V, C, D, H, W = WORKLOADS["cfg2_rpc_3view_768x384x64_c32"]
    feats, rpc, depth = make_inputs(V, C, D, D, 0, H, W, dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    depth = (depth.cpu() + 2.0 * torch.randn((1, D, H, W), generator=g)).contiguous().to(dev)
    out = torch.empty((1, C, D, H, W), dtype=torch.float32, device=dev)
    srcs = _lib.ptr_array(feats[1:])

    plain = make_inputs(V, C, D, D, 0, H, W, dev)[2]
    steps_by_depth = {id(plain): tile_build(_lib, feats, rpc, plain, out, V, C, D, H, W, stream)[0],
                      id(depth): tile_build(_lib, feats, rpc, depth, out, V, C, D, H, W, stream, plane_coef=False)[0]}
    jit_folded = tile_build(_lib, feats, rpc, depth, out, V, C, D, H, W, stream)[0]

    def launch_with(dd):                                        # plain heights: fold + build; per-pixel heights: smvs_rpc_costvol_fwd,
        steps_by_depth[id(dd)]()                                # the entry variance_cost_volume picks for them
    # the kernel runs at the board power limit, so a figure depends on what ran before it: plain and jittered launches ALTERNATE
    # here (same thermal state) and the record carries both
    for _ in range(20):
        launch_with(plain); launch_with(depth); jit_folded()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(40)]
    torch.cuda.synchronize()
    for a, b, c, d in evs:
        a.record(); launch_with(plain); b.record(); launch_with(depth); c.record(); jit_folded(); d.record()
    torch.cuda.synchronize()
    ms_plain = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))
    ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))
    ms_mis = float(np.mean([e[2].elapsed_time(e[3]) for e in evs]))
    bpv = algorithmic_bytes_per_voxel(V, C, D)
    extra["cfg2_jittered_heights_768x384x64_c32"] = {
        "kernel": kernel_name(V, C, D), "ms": round(ms, 4), "Mvox/s": round(D * H * W / ms / 1e3, 1),
        "roofline_frac": round(bpv * D * H * W / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "ms_plain_heights_interleaved": round(ms_plain, 4), "ms_through_the_folded_entry": round(ms_mis, 4),
        "note": "same tile, heights = plane + N(0, 2 m) per pixel, through smvs_rpc_costvol_fwd (the entry variance_cost_volume picks for "
                "per-pixel heights); timed alternating with plain-height launches (fold + smvs_rpc_costvol_fwd_pc) and with the same "
                "jittered tile sent through the folded entry, whose waves run the bivariate geometry, fail the height check and redo it"}
    # the trivariate chain on the plain tile (smvs_rpc_costvol_fwd = no folded coefficients: round 5's path), alternating with the default
    tri = tile_build(_lib, feats, rpc, plain, out, V, C, D, H, W, stream, plane_coef=False)[0]
    for _ in range(20):
        launch_with(plain); tri()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(40)]
    torch.cuda.synchronize()
    for a, b, c in evs:
        a.record(); launch_with(plain); b.record(); tri(); c.record()
    torch.cuda.synchronize()
    ms_pc = float(np.mean([a.elapsed_time(b) for a, b, c in evs]))
    ms_tri = float(np.mean([b.elapsed_time(c) for a, b, c in evs]))
    extra["cfg2_trivariate_chain_768x384x64_c32"] = {
        "ms": round(ms_tri, 4), "roofline_frac": round(bpv * D * H * W / (ms_tri * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "ms_plane_coefficients_interleaved": round(ms_pc, 4),
        "note": "smvs_rpc_costvol_fwd (every voxel through the 20-coefficient cubics) alternating with fold + smvs_rpc_costvol_fwd_pc on the same tile"}
    # the other arithmetic instance on the same tile (smvs_set_arith): launches ALTERNATE with the default's, and the two volumes
    # are compared voxel by voxel
    default_mode = _lib.get_arith()
    other = "exact" if default_mode == "fused" else "fused"
    out2 = torch.empty_like(out)

    builds = {id(out): tile_build(_lib, feats, rpc, plain, out, V, C, D, H, W, stream)[0],
              id(out2): tile_build(_lib, feats, rpc, plain, out2, V, C, D, H, W, stream)[0]}

    def launch_mode(mode, dst):
        _lib.set_arith(mode)
        builds[id(dst)]()
    try:
        for _ in range(20):
            launch_mode(default_mode, out); launch_mode(other, out2)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(40)]
        torch.cuda.synchronize()
        for a, b, c in evs:
            a.record(); launch_mode(default_mode, out); b.record(); launch_mode(other, out2); c.record()
        torch.cuda.synchronize()
        ms_def = float(np.mean([a.elapsed_time(b) for a, b, c in evs]))
        ms_oth = float(np.mean([b.elapsed_time(c) for a, b, c in evs]))
        dmax = float((out - out2).abs().max())
        rel = float(((out - out2).abs() / out2.abs().clamp_min(1.0)).max())
        extra["cfg2_%s_arithmetic_768x384x64_c32" % other] = {
            "ms": round(ms_oth, 4), "Mvox/s": round(D * H * W / ms_oth / 1e3, 1),
            "roofline_frac": round(bpv * D * H * W / (ms_oth * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "ms_%s_interleaved" % default_mode: round(ms_def, 4),
            "max_abs_difference_between_the_two_volumes": float("%.3g" % dmax),
            "max_difference_over_1e-5_max(1,|v|)": float("%.3g" % (rel / 1e-5)),
            "note": "smvs_set_arith(%s): %s; same tile, launches alternating with the default (%s) instance" % (
                other, "the reference's float32 rounding sequence, bit-identical to the oracle" if other == "exact" else "fused arithmetic", default_mode)}
    finally:
        _lib.set_arith(default_mode)
    del plain, out2, builds, steps_by_depth, tri
    del feats, out, depth
    # cfg5: pinhole (homography) volume, 3-view 768x384x64, C=32
    V, C, D, H, W = 3, 32, 64, 384, 768
    g = torch.Generator(device="cpu").manual_seed(0)
    feats = [torch.randn((1, C, H, W), generator=g, dtype=torch.float32).to(dev) for _ in range(V)]
    proj = np.zeros((1, V, 4, 4))
    for v in range(V):
        f = 1.1 * W
        K = np.array([[f, 0, W / 2.0, 0], [0, f, H / 2.0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1]])
        E = np.eye(4)
        E[:3, 3] = [25.0 * v * (-1) ** v, 3.0 * v, 0.5 * v]
        proj[0, v] = K @ E
    from satmvs_amd.modules import warping
    projt = torch.from_numpy(proj).to(dev)
    depth = torch.linspace(400.0, 700.0, D).view(1, D, 1, 1).expand(1, D, H, W).contiguous().to(dev)

    def step5():
        return warping.variance_cost_volume(feats, projt, depth, "pinhole")
    for _ in range(5):
        step5()
    _, ms = time_steps(step5, 20)
    extra["cfg5_pinhole_3view_768x384x64_c32"] = {"ms": round(ms, 4), "Mvox/s": round(D * H * W / ms / 1e3, 1),
                                                  "note": "through the Python surface (compose + volume allocation included)"}
    del feats, depth
    # training path of the headline tile: forward + backward of the variance volume through the autograd surface
    try:
        V, C, D, H, W = WORKLOADS["cfg2_rpc_3view_768x384x64_c32"]
        tf_, trpc, tdepth = make_inputs(V, C, D, D, 0, H, W, dev)
        tf_ = [f.requires_grad_(True) for f in tf_]
        vol = warping.variance_cost_volume(tf_, trpc, tdepth, "rpc")
        gout = torch.randn_like(vol)
        for _ in range(2):
            warping.variance_cost_volume(tf_, trpc, tdepth, "rpc").backward(gout)
        del vol

        def step_f():
            return warping.variance_cost_volume(tf_, trpc, tdepth, "rpc")

        def step_fb():
            warping.variance_cost_volume(tf_, trpc, tdepth, "rpc").backward(gout)
        _, ms_f = time_steps(step_f, 10)
        _, ms_fb = time_steps(step_fb, 10)
        extra["cfg2_training_costvol_fwd_bwd"] = {"ms_forward": round(ms_f, 3), "ms_forward_backward": round(ms_fb, 3),
                                                  "ms_backward": round(ms_fb - ms_f, 3), "kernel": "costvol_bwd_kernel<0,2,8>",
                                                  "note": "smvs_costvol_bwd through torch.autograd (volume allocation and the .grad accumulation included)"}
        del tf_, gout
    except Exception as e:
        extra["cfg2_training_costvol_fwd_bwd"] = {"error": repr(e)[:200]}
    # height error against the reference path on identical inputs -- the reference's own outputs for its seeded
    # Infer_CascadeREDNet are committed as a fixture (tests/golden/cascade.npz: data, generated by tests/golden/gen_golden.py)
    try:
        from satmvs_amd import rpc_synth as _rs
        from satmvs_amd.networks.casred import Infer_CascadeREDNet as _Net
        gz = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "cascade.npz"))
        torch.manual_seed(int(gz["red.seed"]))
        gnet = _Net("rpc", min_interval=2.5, ndepths=[int(v) for v in gz["ndepths"]]).to(dev).eval()
        grpc = gz["rpc"]
        gpm = {"stage1": torch.from_numpy(_rs.rescale_rpc(grpc, 4)).to(dev), "stage2": torch.from_numpy(_rs.rescale_rpc(grpc, 2)).to(dev),
               "stage3": torch.from_numpy(grpc).to(dev)}
        with torch.no_grad():
            gout = gnet(torch.from_numpy(gz["imgs"]).to(dev), gpm, torch.from_numpy(gz["dv"]).to(dev))
        rec = {"fixture": "tests/golden/cascade.npz (the reference's Infer_CascadeREDNet outputs, same seed and inputs)", "target_m": 1e-3}
        for st in ("stage1", "stage2", "stage3"):
            d = np.abs(gout[st]["depth"].cpu().numpy().astype(np.float64) - gz["redinf.%s.depth" % st].astype(np.float64))
            rec[st] = {"max_abs_m": float("%.3g" % d.max()), "mae_m": float("%.3g" % d.mean())}
        extra["height_parity_vs_reference"] = rec
        del gnet, gout
        # ... and at the timed size, both arithmetic modes: the well-conditioned 768x384 cascade of tests/test_full_size_red_conditioned.py
        # (native pipeline vs a float64 evaluation on the reference's variance volume), in a process of its own
        try:
            import subprocess
            tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "conditioned_parity.py")
            torch.cuda.empty_cache()
            r = subprocess.run([sys.executable, tool, "redinf"], capture_output=True, text=True, timeout=300)
            rec["conditioned_768x384"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-300:]}
        except Exception as e:                               # noqa: BLE001 -- a side figure must never take the headline down
            rec["conditioned_768x384"] = {"error": repr(e)[:200]}
    except Exception as e:
        extra["height_parity_vs_reference"] = {"error": repr(e)[:200]}
    # cfg3: one inference cascade forward (FeatureNet + three stages of variance / RED / regression), 48/32/8 planes
    try:
        from satmvs_amd import rpc_synth
        from satmvs_amd.networks.casred import Infer_CascadeREDNet
        torch.manual_seed(0)
        net = Infer_CascadeREDNet("rpc", ndepths=[48, 32, 8]).to(dev).eval()
        imgs = torch.randn(1, 3, 3, H, W, device=dev)
        rpc = rpc_synth.make_view_rpcs(3, H, W, seed=0)[None]
        pm = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 4)).to(dev),
              "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc, 2)).to(dev), "stage3": torch.from_numpy(rpc).to(dev)}
        dv = torch.tensor([[0.0, 400.0]], device=dev)
        with torch.no_grad():
            for _ in range(3):
                net(imgs, pm, dv)
            torch.cuda.synchronize()
            ts, issue = [], []
            for _ in range(12):                               # host-paced plane loop: report the median forward, not a mean with outliers
                t0 = time.perf_counter()
                net(imgs, pm, dv)
                issue.append(round((time.perf_counter() - t0) * 1e3, 2))      # host done issuing
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
            order = [round(t, 2) for t in ts]
            ts.sort()
        rec = {"ms_per_forward": round(ts[len(ts) // 2], 2), "ms_min": round(ts[0], 2), "ms_max": round(ts[-1], 2),
               "ms_in_order": order, "host_issue_ms_in_order": issue,
               "note": "Infer_CascadeREDNet, random weights, B=1, median of 12 forwards.  An isolated 2-3x forward shows up in 0-1 of the 12 at a "
                       "position that changes from run to run, with a normal host issue time (host_issue_ms_in_order): a device-side stall "
                       "between kernels of this process's GPU (clock / power-state transition after the heavy side workloads), not pipeline logic -- "
                       "180 consecutive forwards in a fresh process stay within 7.5-8.1 ms (tools/cascade_outliers.py, profiles/r04_cascade_outliers.txt)"}
        # self-check of what was just timed: the same forward on the stock torch / MIOpen composites (same weights, same inputs)
        try:
            with torch.no_grad():
                nat = net(imgs, pm, dv)
                for k in ("SMVS_RED_TORCH", "SMVS_COSTREG_TORCH", "SMVS_FEATNET_TORCH"):
                    os.environ[k] = "1"
                try:
                    comp = net(imgs, pm, dv)
                finally:
                    for k in ("SMVS_RED_TORCH", "SMVS_COSTREG_TORCH", "SMVS_FEATNET_TORCH"):
                        del os.environ[k]
            diff = {st: (nat[st]["depth"].double() - comp[st]["depth"].double()).abs() for st in ("stage1", "stage2", "stage3")}
            rec["max_abs_m_vs_composite"] = {st: float("%.3g" % d.max()) for st, d in diff.items()}
            rec["mean_abs_m_vs_composite"] = {st: float("%.3g" % d.mean()) for st, d in diff.items()}
            rec["vs_composite_note"] = ("free-running, random weights: float32 round-off of either convolution implementation through 32-48 recurrent "
                                        "steps and a nearly flat softmax; against a float64 evaluation of each stage the native pipeline and the MIOpen composite "
                                        "are equally far (profiles/r04_cascade_float64.txt, tests/test_full_size_cascade.py)")
            del nat, comp, diff
        except Exception as e:
            rec["max_abs_m_vs_composite"] = {"error": repr(e)[:200]}
        # the plane loop of one tile is a chain of dependent small kernels (latency-bound); a scene is many tiles, and the
        # kernels take the batch in their grids: 8 tiles per forward
        B8 = 8
        imgs8 = torch.randn(B8, 3, 3, H, W, device=dev)
        rpc8 = np.stack([rpc_synth.make_view_rpcs(3, H, W, seed=b) for b in range(B8)])
        pm8 = {"stage1": torch.from_numpy(rpc_synth.rescale_rpc(rpc8, 4)).to(dev),
               "stage2": torch.from_numpy(rpc_synth.rescale_rpc(rpc8, 2)).to(dev), "stage3": torch.from_numpy(rpc8).to(dev)}
        dv8 = torch.tensor([[0.0, 400.0]] * B8, device=dev)
        with torch.no_grad():
            for _ in range(2):
                net(imgs8, pm8, dv8)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                net(imgs8, pm8, dv8)
            torch.cuda.synchronize()
        rec["ms_per_tile_at_8_tiles_per_forward"] = round((time.perf_counter() - t0) / 4 / B8 * 1e3, 2)
        extra["cfg3_casred_cascade_48_32_8_768x384"] = rec
    except Exception as e:                                  # a side figure must never take the headline down
        extra["cfg3_casred_cascade_48_32_8_768x384"] = {"error": repr(e)[:200]}
    # the training step of the same network (row f-2), captured in one HIP graph: in a process of its own -- a graph capture holds a
    # private memory pool, and whatever happens there must not reach the headline
    import re
    import subprocess
    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "bench_train_graph.py")
    try:
        r = subprocess.run([sys.executable, tool, "9"], capture_output=True, text=True, timeout=240)
        m = re.search(r"median ([0-9.]+) ms \(min ([0-9.]+), max ([0-9.]+)\), loss ([0-9.eE+-]+)", r.stdout)
        if m:
            extra["cfg3_casred_training_step_48_32_8_768x384"] = {
                "ms_per_step": float(m.group(1)), "ms_min": float(m.group(2)), "ms_max": float(m.group(3)),
                "note": "CascadeREDNet.train(): forward + smooth-L1 cascade loss + backward + RMSprop, B=1, random weights, one HIP graph "
                        "(satmvs_amd.train_graph); tools/bench_train_graph.py in a subprocess, median of 9 replays; round 3: 126.0 ms"}
        else:
            extra["cfg3_casred_training_step_48_32_8_768x384"] = {"error": (r.stderr or r.stdout)[-200:]}
    except Exception as e:
        extra["cfg3_casred_training_step_48_32_8_768x384"] = {"error": repr(e)[:200]}
    # ... and of the two cascades with the 3-D regulariser (CostRegNet: every 3x3x3 layer and BatchNorm3d + ReLU native under autograd)
    for model, key in (("casmvs", "cfg3_casmvs_training_step_48_32_8_768x384"), ("ucs", "cfg3_ucs_training_step_48_32_8_768x384")):
        try:
            r = subprocess.run([sys.executable, tool, "9", model], capture_output=True, text=True, timeout=240)
            m = re.search(r"median ([0-9.]+) ms \(min ([0-9.]+), max ([0-9.]+)\), loss ([0-9.eE+-]+)", r.stdout)
            if m:
                extra[key] = {"ms_per_step": float(m.group(1)), "ms_min": float(m.group(2)), "ms_max": float(m.group(3)),
                              "note": "%s.train(): forward + smooth-L1 cascade loss + backward + RMSprop, B=1, random weights, one HIP graph; "
                                      "with torch's own operators (MIOpen's naive 3-D weight gradient) the same step took 1984 ms on this "
                                      "image (profiles/r05_train_step_casmvs_before.txt)" % ("CascadeMVSNet" if model == "casmvs" else "UCSNet")}
            else:
                extra[key] = {"error": (r.stderr or r.stdout)[-200:]}
        except Exception as e:
            extra[key] = {"error": repr(e)[:200]}
    return extra


def cfg4_strong(dev, stream, rank, world, dist, barrier):
    """BASELINE configs[3]: 5-view 1536x768, 64 height planes split over the ranks (plane_range), C=32 operator-level
    volume, regression slab all-reduced inside every step -- the workload north_star's ">= 6x at 8 GPUs" is stated on.
    Every rank calls this; returns the record on rank 0 (whole-sweep voxels / max-over-ranks time)."""
    from satmvs_amd import _lib, shard
    V, C, D, H, W = 5, 32, 64, 768, 1536
    lo, hi = shard.plane_range(D, rank, world)
    nd = hi - lo
    feats, rpc, depth = make_inputs(V, C, nd, D, lo, H, W, dev)
    out = torch.empty((1, C, max(nd, 1), H, W), dtype=torch.float32, device=dev)
    build = tile_build(_lib, feats, rpc, depth, out, V, C, nd, H, W, stream)[0] if nd > 0 else None
    state = torch.rand((3, 1, H, W), dtype=torch.float64, device=dev) if world > 1 else None

    ex_stream = torch.cuda.Stream(device=dev) if state is not None else None

    def step(overlap=True):
        if nd > 0:
            build()                                             # smvs_rpc_plane_coef + smvs_rpc_costvol_fwd_pc
        if state is not None and overlap:                       # see main(): the exchange trails the step's kernel on its own stream
            ex_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(ex_stream):
                shard.allreduce_regression_state(state)
        elif state is not None:
            shard.allreduce_regression_state(state)
    for _ in range(40):                                        # a fixed count (the step holds a collective): ~0.2 s at one GPU -- steady clocks
        step()
    steps = 10
    elapsed, _ = time_steps(step, steps, barrier)
    serial_ms = exch = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        for _ in range(2):
            step(False)
        se, _ = time_steps(lambda: step(False), steps, barrier)
        t = torch.tensor([se], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        serial_ms = float(t.item()) / steps * 1e3
        _, ex_ms = time_steps(lambda: shard.allreduce_regression_state(state), 10, barrier)
        slab = int(state.numel() * 8)
        exch = {"bytes": slab, "bytes_sent_per_rank": int(2 * (world - 1) * slab // world), "ms": round(ex_ms, 4),
                "GB/s_per_rank_and_direction": round(2 * (world - 1) * slab / world / (ex_ms * 1e-3) / 1e9, 2)}
    ms = elapsed / steps * 1e3
    bpv = algorithmic_bytes_per_voxel(V, C, D)
    return {"workload": "cfg4_rpc_5view_1536x768x64_c32", "planes_per_gpu": nd, "ms_per_step": round(ms, 4), "exchange": exch,
            "Mvox/s": round(D * H * W / ms / 1e3, 1), "scaling": "strong",
            "roofline_frac_aggregate": round(bpv * D * H * W / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4),
            "exchange_overlapped": world > 1, "ms_per_step_not_overlapped": None if serial_ms is None else round(serial_ms, 4),
            "note": "whole 64-plane sweep / step time incl. the (3,1,768,1536) f64 slab all-reduce (28 MB) when N > 1"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="cfg2_rpc_3view_768x384x64_c32", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the side workloads reported under \"extra\"")
    ap.add_argument("--prewarm-seconds", type=float, default=0.6)
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

    from satmvs_amd import _lib, shard
    _lib.load()
    V, C, D, H, W = WORKLOADS[args.workload]
    # Strong scaling: the D planes of the BASELINE tile are split over the ranks (plane_range), every rank holds the
    # replicated features / RPCs, builds its planes, and the step ends with the path's one exchange: the all-reduce
    # of the (3,1,H,W) float64 regression partials (satmvs_amd/shard.py).  N = 1: the whole volume, no exchange.
    lo, hi = shard.plane_range(D, rank, world)
    D_local = hi - lo
    feats, rpc, depth = make_inputs(V, C, D_local, D, lo, H, W, dev)
    if args.workload in SIDE_HEIGHTS:                        # the planes a real run hands to that launch, as in side_workloads()
        h0, h1 = SIDE_HEIGHTS[args.workload]
        depth = torch.linspace(h0, h1, D, dtype=torch.float32)[lo:hi].view(1, D_local, 1, 1).expand(1, D_local, H, W).contiguous().to(dev)
    out = torch.empty((1, C, max(D_local, 1), H, W), dtype=torch.float32, device=dev)
    stream = _lib.current_stream(dev)
    state = torch.rand((3, 1, H, W), dtype=torch.float64, device=dev) if world > 1 else None
    # One step = what a caller issues per tile: smvs_rpc_plane_coef (the source cubics folded at every plane's height, one tiny
    # launch) + smvs_rpc_costvol_fwd_pc (the build).  Both are inside the timed region; `kernel` (the build alone) is timed
    # separately for the roofline of the dominant kernel.
    use_pc = os.environ.get("SMVS_BENCH_TRIVARIATE") != "1"       # A/B only: 1 = smvs_rpc_costvol_fwd, the trivariate chain for every voxel
    # a rank's shard below variance_cost_volume's fold threshold (8 Mi voxels: N >= 4 at cfg2) is built as that function would build it
    from satmvs_amd.modules.warping import _FOLD_MIN_VOXELS
    use_pc = use_pc and D_local * H * W >= _FOLD_MIN_VOXELS
    launch, kernel, _ = tile_build(_lib, feats, rpc, depth, out, V, C, D_local, H, W, stream, plane_coef=use_pc) if D_local > 0 else ((lambda: None),) * 3

    # A scene is a stream of tiles: the exchange of tile k waits for tile k's kernel (stream order through wait_stream) but tile
    # k+1's kernel does not wait for it -- it runs on its own stream, RCCL's kernels under the next build.  Exchanges stay in
    # order on that stream; everything drains inside the timed region (device-wide synchronize in time_steps).
    ex_stream = torch.cuda.Stream(device=dev) if state is not None else None

    def step():
        launch()
        if state is not None:
            ex_stream.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(ex_stream):
                shard.allreduce_regression_state(state)

    def barrier():
        if dist is not None:
            dist.barrier()

    prewarm(launch, args.prewarm_seconds)
    for _ in range(args.warmup):
        step()
    elapsed, _ = time_steps(step, args.steps, barrier)
    # the dominant kernel alone (no fold, no exchange) for the roofline: HIP events around each launch; and fold + build together
    _, kern_ms = time_steps(kernel, min(args.steps, 100))
    _, both_ms = time_steps(launch, min(args.steps, 100))

    exchange = None
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        _, ex_ms = time_steps(lambda: shard.allreduce_regression_state(state), 20, barrier)

        # the same steps with the exchange IN stream order (build -> exchange -> next build): single-tile latency, beside the
        # tile-stream throughput that `value` reports
        def step_serial():
            launch()
            shard.allreduce_regression_state(state)
        for _ in range(3):
            step_serial()
        serial_elapsed, _ = time_steps(step_serial, min(args.steps, 50), barrier)
        t = torch.tensor([serial_elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        serial_ms = float(t.item()) / min(args.steps, 50) * 1e3
        slab = int(state.numel() * 8)
        exchange = {"op": "reduce-scatter (all_to_all of pixel chunks + rank-ordered local sum,sum,max) + all_gather of the (3,1,%d,%d) f64 regression partials, inside the timed step" % (H, W),
                    "bytes": slab, "bytes_sent_per_rank": int(2 * (world - 1) * slab // world), "ms": round(ex_ms, 4),
                    "GB/s_per_rank_and_direction": round(2 * (world - 1) * slab / world / (ex_ms * 1e-3) / 1e9, 2),
                    "overlap": "issued behind the step's kernel on its own stream; the next step's kernel does not wait for it (a stream of tiles)",
                    "ms_per_step_not_overlapped": round(serial_ms, 4),
                    "device_work": "all_to_all_single out of the slab + smvs_regress_fold (one kernel) + in-place all_gather_into_tensor"}

    ids = rank_identities(dist, dev, world, rank, one_device)   # collective: every rank takes part
    cfg4 = None
    if not args.no_extra:
        del out
        torch.cuda.empty_cache()
        cfg4 = cfg4_strong(dev, stream, rank, world, dist, barrier)     # collective: every rank takes part

    if rank == 0:
        vox_per_step = D * H * W                              # the whole tile, whatever the rank count
        ms_per_step = elapsed / args.steps * 1e3
        bpv = algorithmic_bytes_per_voxel(V, C, D)
        achieved = bpv * D_local * H * W / (kern_ms * 1e-3) / 1e9
        line = {
            "metric": "cost-volume Mvoxels/s (fused RPC warp + variance build)",
            "value": round(vox_per_step / (elapsed / args.steps) / 1e6, 1),
            "unit": "Mvox/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32 features / f64 RPC geometry", "data": "synthetic",
            "config": {"workload": args.workload, "arithmetic": _lib.get_arith() + " (smvs_set_arith; the library default is fused: float64 geometry and float32 taps "
                       "bit-identical to the reference's, variance within 1e-5 max(1,|v|) of it; the exact instance is timed under extra)", "views": V, "channels": C, "planes_per_gpu": D_local,
                       "planes_total": D, "H": H, "W": W, "depth_values": "per-voxel (B,D,H,W)",
                       "sharding": "height planes of one tile split over the ranks, regression partials exchanged each step (overlapping the next step's build when N > 1)",
                       "prewarm_seconds": args.prewarm_seconds},
            "roofline": {"bound": "hbm", "kernel": kernel_name(V, C, D_local),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(args.workload) if world == 1 else None,
                         "bytes_per_voxel": bpv, "kernel_ms": round(kern_ms, 4), "fold_plus_kernel_ms": round(both_ms, 4),
                         "note": ("one step = smvs_rpc_plane_coef (fold of the source cubics per plane, 3-5 us) + smvs_rpc_costvol_fwd_pc; `value` and "
                                  "ms_per_step time both, kernel_ms / achieved / frac are the build kernel alone (HIP events on its stream)") if use_pc else
                                 "one step = smvs_rpc_costvol_fwd (a shard below 8 Mi voxels is not worth the fold's launch: variance_cost_volume's rule)"},
        }
        line["devices"] = ids
        if exchange is not None:
            line["exchange"] = exchange
            line["exchange_overlapped"] = True              # `value` is tile-stream throughput; exchange.ms_per_step_not_overlapped is the single-tile latency
        if not args.no_extra:
            line["extra"] = {"cfg4_strong_scaling": cfg4}
            if world == 1:
                line["extra"].update(side_workloads(dev, stream))
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(V, C, D, H, W)
            line["cpu_baseline_torch"] = cpu_baseline_torch(V, C, D, H, W)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
