"""Multi-process (world_size 2, gloo, CPU) test of the height-plane sharding host logic: each rank
folds ITS planes into the float64 accumulators (here with the CPU oracle standing in for the GPU
kernels -- this file tests satmvs_amd/shard.py, not the kernels), one all-reduce (sum,sum,max)
yields the same height map on every rank as the single-process run and as the golden vector."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from satmvs_amd import shard
    g = np.load(os.path.join(ROOT, "tests", "golden", "regress.npz"))
    reg, dv = g["reg"], g["depth_values"]
    B, D, H, W = reg.shape
    lo, hi = shard.plane_range(D, rank, world)
    acc = orc.StreamRegress(B, H, W)
    for d in range(lo, hi):
        acc.step(reg[:, d], dv, d)
    state = torch.from_numpy(np.stack([acc.exp_sum[:, 0], acc.depth_img[:, 0], acc.max_prob[:, 0]]))
    shard.allreduce_regression_state(state)
    depth, conf = shard.finish_regression(state)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), depth=depth.numpy(), conf=conf.numpy(), lo=lo, hi=hi)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_regression_world2(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    g = np.load(os.path.join(ROOT, "tests", "golden", "regress.npz"))
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert (int(r0["lo"]), int(r0["hi"]), int(r1["lo"]), int(r1["hi"])) == (0, 4, 4, 7)
    np.testing.assert_array_equal(r0["depth"], r1["depth"])
    np.testing.assert_allclose(r0["depth"], g["st_depth"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(r0["conf"], g["st_conf"], rtol=1e-6)


def test_allreduce_is_identity_without_process_group():
    from satmvs_amd import shard
    s = torch.rand(3, 1, 4, 5, dtype=torch.float64)
    t = s.clone()
    assert shard.allreduce_regression_state(t) is t and torch.equal(s, t)


class _FakeRecurrentRegulariser:
    """Stand-in for the native RED plane pipeline on the CPU (this file tests the host logic of satmvs_amd/shard.py, not the
    kernels): four 'hidden states' and the (3,B,H,W) accumulators evolve plane by plane in a way that depends on every
    previous plane, so any wrong hand-off, plane range or tile mix-up changes the bits."""

    @staticmethod
    def initial_states(b, h, w, device):
        return [torch.zeros((b, c, h >> g, w >> g), dtype=torch.float32, device=device) for g, c in enumerate((2, 3, 4, 5))]

    def _use_native(self, ref):
        return True

    def native_pred_planes(self, features, proj, dv, geo_model, use_qc, states, acc_state, lo, hi):
        ref = features[0]
        for d in range(lo, hi):
            x = (ref.mean(1, keepdim=True) * float(d + 1) + features[1].amax(1, keepdim=True)).float()
            for g, st in enumerate(states):
                pooled = torch.nn.functional.avg_pool2d(x, 1 << g) if g else x
                st.mul_(0.75).add_(torch.tanh(pooled + st.sum(1, keepdim=True)) * (0.1 + 0.05 * g))
            reg = states[0].sum(1) + torch.nn.functional.interpolate(states[3].sum(1, keepdim=True), scale_factor=8.0)[:, 0]
            pr = torch.exp(reg.double())
            acc_state[0] += pr
            acc_state[1] += pr * dv[:, d].double().view(-1, 1, 1)
            torch.maximum(acc_state[2], pr, out=acc_state[2])


class _CpuAcc:
    def __init__(self, b, h, w, device):
        self.state = torch.zeros((3, b, h, w), dtype=torch.float64, device=device)

    def result(self):
        den = self.state[0] + 1e-10
        return (self.state[1] / den).float(), (self.state[2] / den).float()


def _stream_tiles():
    g = torch.Generator().manual_seed(4)
    tiles = []
    for t in range(5):
        feats = [torch.randn((1, 4, 16, 24), generator=g) for _ in range(2)]
        dv = torch.linspace(10.0 + t, 90.0 + 2 * t, 7).view(1, 7)
        tiles.append((feats, None, dv))
    return tiles


def _stream_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from satmvs_amd import shard
    from satmvs_amd.modules import module
    module.StreamingRegression = _CpuAcc                   # the accumulator class shard.py instantiates (CPU stand-in)
    outs = shard.sharded_pred_stream(_stream_tiles(), 7, _FakeRecurrentRegulariser())
    np.savez(os.path.join(out_dir, "s%d.npz" % rank), **{"d%d" % i: o["depth"].numpy() for i, o in enumerate(outs)},
             **{"c%d" % i: o["photometric_confidence"].numpy() for i, o in enumerate(outs)})
    dist.barrier()
    dist.destroy_process_group()


def test_tile_pipelined_stream_world3_host_logic(tmp_path):
    """shard.sharded_pred_stream on the CPU with a stand-in recurrent regulariser: 5 tiles through 3 ranks (plane ranges
    3/2/2), non-blocking hand-offs, results broadcast at the end -- every tile equals the single-process run bit for bit
    on every rank."""
    from satmvs_amd import shard
    from satmvs_amd.modules import module
    port = _free_port()
    mp.spawn(_stream_worker, args=(3, port, str(tmp_path)), nprocs=3, join=True)
    saved = module.StreamingRegression
    module.StreamingRegression = _CpuAcc
    try:
        single = shard.sharded_pred_stream(_stream_tiles(), 7, _FakeRecurrentRegulariser())      # no process group: one rank, all planes
    finally:
        module.StreamingRegression = saved
    for k in range(3):
        r = np.load(tmp_path / ("s%d.npz" % k))
        for i, o in enumerate(single):
            assert np.array_equal(r["d%d" % i], o["depth"].numpy()), "rank %d tile %d" % (k, i)
            assert np.array_equal(r["c%d" % i], o["photometric_confidence"].numpy())
    assert not np.array_equal(single[0]["depth"].numpy(), single[1]["depth"].numpy())
