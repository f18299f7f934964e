"""Two ranks sharing cuda:0 (gloo rendezvous, collectives staged through the host): the height-plane sharded pred path
-- native cost-volume planes + RED plane loop per shard, hidden-state / accumulator hand-off, final broadcast -- must
reproduce the single-process result BIT FOR BIT on every rank.  This exercises satmvs_amd/shard.py end to end with the
HIP kernels on a one-GPU box; RCCL itself (backend "nccl") needs two devices and is covered by bench.py --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, recurrent):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from satmvs_amd import shard
    from satmvs_amd.modules.module import slice_RED_Regularization
    from satmvs_amd.networks.casred import compute_depth_when_pred
    g = np.load(os.path.join(ROOT, "tests", "golden", "red_pred.npz"))
    reg = slice_RED_Regularization(8, 8).eval()
    reg.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")})
    reg = reg.to(dev)
    feats = [torch.from_numpy(f).to(dev) for f in g["feats"]]
    rpc, dv = torch.from_numpy(g["rpc"]).to(dev), torch.from_numpy(g["depth"]).to(dev)
    with torch.no_grad():
        single = compute_depth_when_pred(feats, rpc, dv, dv.shape[1], reg, "rpc", False)
        shrd = shard.sharded_compute_depth_when_pred(feats, rpc, dv, dv.shape[1], reg, "rpc", False,
                                                     recurrent_handoff=recurrent)
        # a later cascade stage: hypotheses arrive as a GeneratedHeights description (no (B,D,H,W) tensor), 6 planes
        # 2.5 m apart around the height map the first call produced (the sharded path must take it)
        from satmvs_amd.modules.depth_range import GeneratedHeights
        h, w = feats[0].shape[2:]
        gen = GeneratedHeights(single["depth"], 6, 2.5, (h, w), (h, w))
        gsingle = compute_depth_when_pred(feats, rpc, gen, 6, reg, "rpc", False)
        gshrd = shard.sharded_compute_depth_when_pred(feats, rpc, gen, 6, reg, "rpc", False, recurrent_handoff=recurrent)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
             single_depth=single["depth"].cpu().numpy(), single_conf=single["photometric_confidence"].cpu().numpy(),
             depth=shrd["depth"].cpu().numpy(), conf=shrd["photometric_confidence"].cpu().numpy(),
             gen_single=gsingle["depth"].cpu().numpy(), gen_depth=gshrd["depth"].cpu().numpy(),
             planes=np.array(shard.plane_range(dv.shape[1], rank, world)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["exact", "fused"])
def test_sharded_red_pred_two_ranks_bit_identical(tmp_path, mode):
    """... in the suite's arithmetic ("exact") and in the library's default ("fused": the ranks inherit SMVS_ARITH): plane windows of the
    fused instance are bit-identical to the whole sweep too, so the sharded result equals the single-GPU one in either mode."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    saved = os.environ.get("SMVS_ARITH")
    os.environ["SMVS_ARITH"] = mode
    try:
        mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), True), nprocs=2, join=True)
    finally:
        os.environ["SMVS_ARITH"] = saved if saved is not None else "exact"
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(2)]
    assert tuple(r[0]["planes"]) != tuple(r[1]["planes"]) and r[0]["planes"][1] == r[1]["planes"][0]   # a real split
    for k in range(2):
        assert np.array_equal(r[k]["depth"], r[k]["single_depth"]), "rank %d depth differs from the single-GPU run" % k
        assert np.array_equal(r[k]["conf"], r[k]["single_conf"])
    assert np.array_equal(r[0]["depth"], r[1]["depth"]) and np.array_equal(r[0]["conf"], r[1]["conf"])
    for k in range(2):                                      # generated (stage 2/3 style) hypotheses through the sharded path
        assert np.array_equal(r[k]["gen_depth"], r[k]["gen_single"]), "rank %d: generated-heights stage differs" % k
    g = np.load(os.path.join(ROOT, "tests", "golden", "red_pred.npz"))
    assert np.abs(r[0]["depth"] - g["pred_depth"]).max() <= 1e-3          # and it is the reference's height map


def test_sharded_without_handoff_differs_only_by_the_recurrence(tmp_path):
    """recurrent_handoff=False (each shard starts from zero states, one all-reduce) runs the parallel exchange path;
    it is NOT the reference's recurrence, so only finiteness and rank agreement are asserted."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), False), nprocs=2, join=True)
    r = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % k)) for k in range(2)]
    assert np.isfinite(r[0]["depth"]).all() and np.isfinite(r[0]["gen_depth"]).all()
    assert np.array_equal(r[0]["depth"], r[1]["depth"]) and np.array_equal(r[0]["conf"], r[1]["conf"])
    assert np.array_equal(r[0]["gen_depth"], r[1]["gen_depth"])


def _stream_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from satmvs_amd import shard
    from satmvs_amd.modules.module import slice_RED_Regularization
    from satmvs_amd.networks.casred import compute_depth_when_pred
    g = np.load(os.path.join(ROOT, "tests", "golden", "red_pred.npz"))
    reg = slice_RED_Regularization(8, 8).eval()
    reg.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")})
    reg = reg.to(dev)
    rpc, dv = torch.from_numpy(g["rpc"]).to(dev), torch.from_numpy(g["depth"]).to(dev)
    tiles = []
    for t in range(4):                                      # four different tiles: features scaled / rolled, hypotheses shifted
        feats = [torch.roll(torch.from_numpy(f).to(dev) * (1.0 + 0.25 * t), shifts=3 * t, dims=3) for f in g["feats"]]
        tiles.append((feats, rpc, (dv + 1.5 * t).contiguous()))
    D = dv.shape[1]
    with torch.no_grad():
        singles = [compute_depth_when_pred(f, r, d, D, reg, "rpc", False) for f, r, d in tiles]
        outs = shard.sharded_pred_stream(tiles, D, reg, "rpc", False)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "stream%d.npz" % rank),
             **{"single%d" % t: s["depth"].cpu().numpy() for t, s in enumerate(singles)},
             **{"depth%d" % t: o["depth"].cpu().numpy() for t, o in enumerate(outs)},
             **{"conf%d" % t: o["photometric_confidence"].cpu().numpy() for t, o in enumerate(outs)},
             **{"sconf%d" % t: s["photometric_confidence"].cpu().numpy() for t, s in enumerate(singles)})
    dist.barrier()
    dist.destroy_process_group()


def test_tile_pipelined_stream_two_ranks_bit_identical(tmp_path):
    """shard.sharded_pred_stream: 4 tiles through 2 ranks (rank 1 continues tile t while rank 0 already runs tile t+1);
    every tile's height map and confidence equal the single-process ones bit for bit on both ranks."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    mp.spawn(_stream_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r = [np.load(os.path.join(str(tmp_path), "stream%d.npz" % k)) for k in range(2)]
    for k in range(2):
        for t in range(4):
            assert np.array_equal(r[k]["depth%d" % t], r[k]["single%d" % t]), "rank %d tile %d" % (k, t)
            assert np.array_equal(r[k]["conf%d" % t], r[k]["sconf%d" % t]), "rank %d tile %d" % (k, t)
    assert not np.array_equal(r[0]["depth0"], r[0]["depth1"])                   # the tiles really differ


def test_bench_two_rank_rehearsal_on_one_device(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), both ranks on cuda:0
    with the gloo rendezvous: the strong-scaling step (plane shard + slab exchange inside the timed region, cfg2 and
    cfg4) runs end to end and prints its JSON line.  Not a measurement -- the N>1 numbers come from the driver's 8-GPU
    node -- but the code path the SCALE run takes is executed on every GPU test run."""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, SMVS_BENCH_ONE_DEVICE="1", SMVS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--prewarm-seconds", "0.05"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["planes_per_gpu"] == 32 and line["scaling"] == "strong"
    assert line["exchange"]["bytes"] == 3 * 384 * 768 * 8 and line["exchange"]["ms"] > 0
    assert line["extra"]["cfg4_strong_scaling"]["planes_per_gpu"] == 32
    assert line["value"] > 0 and "cpu_baseline" not in line
    # the fields that let a real SCALE run be checked: backend, every rank's device, bytes on the wire
    assert line["devices"]["backend"] == "gloo" and [r["rank"] for r in line["devices"]["ranks"]] == [0, 1]
    assert all(r["pci_bus_id"] and r["pid"] > 0 for r in line["devices"]["ranks"])
    assert line["exchange"]["bytes_sent_per_rank"] == 3 * 384 * 768 * 8 and line["exchange"]["ms_per_step_not_overlapped"] > 0
    c4 = line["extra"]["cfg4_strong_scaling"]
    assert c4["exchange"]["bytes"] == 3 * 768 * 1536 * 8 and c4["exchange"]["ms"] > 0 and c4["ms_per_step_not_overlapped"] > 0


def _nccl_smoke(port, out_path):
    import sys
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1)           # "nccl" is RCCL on ROCm
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    from satmvs_amd import shard
    g = torch.Generator(device="cpu").manual_seed(3)
    state = torch.randn((3, 2, 24, 40), generator=g, dtype=torch.float64).to(dev)
    want = state.clone()
    shard.allreduce_regression_state(state, force=True)             # all_to_all_single + smvs_regress_fold + in-place all_gather_into_tensor
    ragged = torch.randn((3, 1, 5, 7), generator=g, dtype=torch.float64).to(dev)
    want2 = ragged.clone()
    shard.allreduce_regression_state(ragged, force=True)
    b = torch.arange(16, dtype=torch.float32, device=dev)
    dist.broadcast(b, src=0)                                        # the last shard's broadcast of the finished sums
    torch.cuda.synchronize()
    ok = bool(torch.equal(state, want) and torch.equal(ragged, want2) and b.sum().item() == 120.0)
    with open(out_path, "w") as f:
        f.write("%s %s %s" % (ok, dist.get_backend(), ".".join(str(x) for x in torch.cuda.nccl.version())))
    dist.destroy_process_group()


def test_rccl_entry_points_smoke_world_size_one(tmp_path):
    """The exchange of shard.allreduce_regression_state through backend "nccl" (RCCL) with a group of ONE rank: every RCCL
    entry point the N > 1 path uses (all_to_all_single out of the slab, in-place all_gather_into_tensor, broadcast) is
    called once on the real backend with device buffers and the fold kernel between them; the slab must come back
    unchanged.  The multi-rank numerics are covered by the gloo tests above; this one only proves the calls are legal
    for RCCL (a one-GPU box has no second device to talk to)."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    out = str(tmp_path / "smoke.txt")
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_smoke, args=(_free_port(), out))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    ok, backend, version = open(out).read().split()
    assert ok == "True" and backend == "nccl" and version[0].isdigit()


@pytest.mark.parametrize("world,B,H,W", [(8, 1, 384, 768), (2, 2, 24, 40), (4, 1, 8, 24)])
def test_regress_fold_kernel_is_the_rank_ordered_sum_sum_max(world, B, H, W):
    """smvs_regress_fold (the reduce step of shard.allreduce_regression_state on RCCL): rank r's chunk of the flattened
    (3,B,H,W) slab from `world` received copies, rows 0-1 summed and row 2 maxed in rank order, for every rank's chunk --
    against the float64 fold in numpy, bit for bit; chunks that straddle the row boundaries included."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import sys
    sys.path.insert(0, ROOT)
    from satmvs_amd import _lib
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    row = B * H * W
    n = 3 * row
    assert n % world == 0
    chunk = n // world
    slabs = rng.standard_normal((world, n))                     # every rank's slab
    ref = slabs[0].copy()
    for r in range(1, world):                                   # rank order
        ref[:2 * row] = ref[:2 * row] + slabs[r, :2 * row]
        ref[2 * row:] = np.maximum(ref[2 * row:], slabs[r, 2 * row:])
    out = torch.zeros(n, dtype=torch.float64, device=dev)
    for rank in range(world):
        recv = torch.from_numpy(np.ascontiguousarray(slabs[:, rank * chunk:(rank + 1) * chunk])).to(dev)
        _lib.call("smvs_regress_fold", _lib.ptr(recv), _lib.ptr(out[rank * chunk:]), world, chunk, rank * chunk, row, _lib.current_stream(dev))
    assert np.array_equal(out.cpu().numpy(), ref)
