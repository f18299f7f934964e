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
