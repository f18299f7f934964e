"""Height-plane sharding across the GPUs of one node (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests of the host logic).

The warp + variance build has no cross-plane dependence (networks/casred.py:191-212 of the
reference: the loop body only meets other planes in the regulariser), so rank g of G builds planes
[plane_range(D, g, G)) with replicated feature maps and camera parameters and NO communication.
The only exchange on the path is the regression: the pred loop's accumulators
(exp_sum, depth_img, max_prob -- casred.py:182-184) are additive over planes, so one all-reduce
(sum, sum, max) of a (3,B,H,W) float64 slab yields the height map on every rank.  The C-channel
variance volume is never gathered (2.4 GB at the metric shape vs 7 MB for the slab).

Caveat kept explicit: the RED regulariser is recurrent over planes (modules/module.py:625-644), so
for exact parity its four hidden states -- and, with them, the regression accumulators -- are handed
from shard g to shard g+1 (`recurrent_handoff=True`, point-to-point, sequential within one tile; the
last shard broadcasts the finished sums, which equal the single-GPU ones bit for bit); plane-local
regularisers shard with no hand-off and one all-reduce.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def plane_range(num_planes, rank, world):
    """Contiguous, balanced split of `num_planes`: the first (num_planes % world) ranks get one more."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %d/%d" % (rank, world))
    q, r = divmod(num_planes, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def allreduce_regression_state(state, group=None, force=False):
    """In-place reduction of the (3,B,H,W) float64 accumulators over the ranks: rows 0,1 summed, row 2 maxed.

    A reduce-scatter + all-gather that carries all three rows at once, with no layout copies: the flattened slab is cut into
    `world` contiguous chunks; every rank sends chunk r of its slab to rank r (all_to_all straight out of the slab: on xGMI
    seven direct point-to-point transfers in parallel, 7/8 of 7 MB out per rank at 768x384), folds the copies it received in
    rank order with ONE kernel (smvs_regress_fold: sum, sum, max by the row an element belongs to -- the same association
    on every rank and run, so the result is deterministic) straight into its chunk of the slab, and the chunks are
    all-gathered IN PLACE (input = this rank's chunk of the output).  Per exchange on the device: 2 collectives + 1 kernel
    (round 3: 2 collectives + 6 torch kernels).  A slab whose length the rank count does not divide is padded (one copy in,
    one out); CPU tensors and the host-staged gloo rehearsal fold with torch operators.  `force` runs the collectives for a
    group of one rank as well (the result is the input): the smoke test of the RCCL entry points on a one-GPU box."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return state
    if state.shape[0] != 3:
        raise ValueError("state must be (3,B,H,W): [exp_sum, depth_img, max_prob]")
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    staged = _host_staged(state, group)
    row_len = state[0].numel()
    n = 3 * row_len
    chunk = (n + world - 1) // world
    direct = state.is_contiguous() and n == world * chunk and not staged
    if direct:
        flat = state.detach().view(-1)
    else:
        flat = torch.zeros((world * chunk,), dtype=state.dtype, device="cpu" if staged else state.device)
        flat[:n] = state.detach().reshape(-1).to(flat.device)   # zeros pad the tail: neutral for the sums, cut off again for the max row
    recv = torch.empty((world, chunk), dtype=flat.dtype, device=flat.device)            # [source rank][element of MY chunk]
    dist.all_to_all_single(recv.view(-1), flat, group=group)
    mine = flat[rank * chunk:(rank + 1) * chunk]
    if flat.is_cuda:
        from . import _lib
        with torch.cuda.device(flat.device):
            _lib.call("smvs_regress_fold", _lib.ptr(recv), _lib.ptr(mine), world, chunk, rank * chunk, row_len, _lib.current_stream(flat.device))
    else:
        idx = torch.arange(rank * chunk, (rank + 1) * chunk)
        is_max = (idx // row_len) >= 2
        acc = recv[0].clone()
        for r in range(1, world):                              # rank order, like the kernel
            acc = torch.where(is_max, torch.maximum(acc, recv[r]), acc + recv[r])
        mine.copy_(acc)
    dist.all_gather_into_tensor(flat, mine, group=group)       # in place: `mine` is this rank's slice of `flat`
    if not direct:
        state.copy_(flat[:n].reshape(state.shape).to(state.device))
    return state


def handoff_buffers(cost_regularization, b, h, w, device):
    """The recurrent hand-off of one tile as ONE message: a flat float32 buffer holding the regression accumulators
    ((3,B,H,W) float64, first: 8-byte aligned) followed by the four hidden states, and views of it that the plane pipeline
    updates in place.  Returns (flat, states, accumulator view)."""
    shapes = [tuple(t.shape) for t in cost_regularization.initial_states(b, h, w, torch.device("meta"))]
    acc_floats = 2 * 3 * b * h * w
    sizes = [int(torch.Size(sh).numel()) for sh in shapes]
    flat = torch.zeros((acc_floats + sum(sizes),), dtype=torch.float32, device=device)
    acc = flat[:acc_floats].view(torch.float64).view(3, b, h, w)
    states, o = [], acc_floats
    for sh, n in zip(shapes, sizes):
        states.append(flat[o:o + n].view(sh))
        o += n
    return flat, states, acc


def finish_regression(state):
    """(3,B,H,W) float64 -> depth, confidence (float32), networks/casred.py:234-236 (torch ops; any device)."""
    den = state[0] + 1e-10
    return (state[1] / den).float(), (state[2] / den).float()


def sharded_compute_depth_when_pred(features, proj_matrices, depth_values, num_depth, cost_regularization,
                                    geo_model="rpc", use_qc=False, group=None, recurrent_handoff=True):
    """compute_depth_when_pred (networks/casred.py:161-238) with the height planes sharded over `group`.

    Every rank passes the same replicated inputs and gets the same full-resolution result.
    """
    from .modules.depth_range import GeneratedHeights
    from .modules.module import StreamingRegression
    from .modules.warping import variance_cost_volume

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = plane_range(num_depth, rank, world)
    ref = features[0]
    b, _, h, w = ref.shape
    recurrent = hasattr(cost_regularization, "initial_states")
    chain = recurrent and recurrent_handoff and world > 1
    acc = StreamingRegression(b, h, w, ref.device)
    flat = None
    if chain:
        # The recurrence serialises the shards of one tile anyway, so the regression accumulators travel with the four
        # hidden states: rank g continues the float64 sums exactly where rank g-1 stopped and the result is the
        # single-GPU one bit for bit (a tree reduction would re-associate the float64 additions).  States and accumulators
        # are views of ONE buffer = one message per shard boundary (round 3: five, on RCCL's in-order point-to-point stream).
        flat, states, acc.state = handoff_buffers(cost_regularization, b, h, w, ref.device)
        if rank > 0:
            _recv(flat, _global_rank(group, rank - 1), group)
    else:
        states = cost_regularization.initial_states(b, h, w, ref.device) if recurrent else []
    # stages 2-3 of the inference cascades hand over a GeneratedHeights description instead of a (B,D,H,W) tensor
    # (modules/depth_range.py::stage_hypotheses): the native plane pipeline evaluates it per pixel, the composite loop
    # needs the tensor
    gen = isinstance(depth_values, GeneratedHeights)
    dv = depth_values if gen else depth_values.detach().to(torch.float32).contiguous()
    if recurrent and hasattr(cost_regularization, "native_pred_planes") and cost_regularization._use_native(ref):
        cost_regularization.native_pred_planes(features, proj_matrices, dv, geo_model, use_qc, states, acc.state, lo, hi)
    else:
        if gen:
            dv = depth_values.materialize()
        for d in range(lo, hi):
            plane = variance_cost_volume(features, proj_matrices, dv, geo_model, use_qc, d_begin=d, d_end=d + 1)
            if recurrent:
                reg, *states = cost_regularization(plane.squeeze(2), *states)
            else:
                reg = cost_regularization(plane.squeeze(2))
            acc.step(reg, dv, d)
    if chain:
        if rank < world - 1:
            for t, v in zip(states, handoff_views(flat, states)):
                if t.data_ptr() != v.data_ptr():                              # the composite loop returns fresh state tensors
                    v.copy_(t)
            _send(flat, _global_rank(group, rank + 1), group)
        _broadcast(acc.state, _global_rank(group, world - 1), group)          # the last shard holds the whole sum
    else:
        allreduce_regression_state(acc.state, group)
    depth, confidence = acc.result()
    return {"depth": depth, "photometric_confidence": confidence}


def sharded_pred_stream(tiles, num_depth, cost_regularization, geo_model="rpc", use_qc=False, group=None):
    """A STREAM of tiles through the plane-sharded RED pred path, pipelined over the ranks.

    The recurrence over planes serialises the shards of ONE tile (rank g can only start when rank g-1 hands over the
    hidden states and accumulators), so sharding a single tile buys memory, not latency.  Over a stream of tiles the
    ranks form a pipeline instead: while rank g+1 continues tile t, rank g already runs its planes of tile t+1 --
    rank g works on tile t-g at any moment, every GPU is busy once the pipe is full, and each tile's result is still the
    single-GPU one bit for bit (same kernels, same plane order, float64 sums continued, never re-associated).
    Hand-offs are non-blocking sends of fresh per-tile buffers; the finished accumulators stay on the last rank until
    the end of the stream and are broadcast once.

    tiles: iterable of (features, proj_matrices, depth_values) -- replicated on every rank, like
    sharded_compute_depth_when_pred.  Returns one {"depth", "photometric_confidence"} dict per tile on every rank."""
    from .modules.depth_range import GeneratedHeights
    from .modules.module import StreamingRegression

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = plane_range(num_depth, rank, world)
    native = None
    finished, pending = [], []
    for features, proj_matrices, depth_values in tiles:
        ref = features[0]
        b, _, h, w = ref.shape
        if native is None:
            native = hasattr(cost_regularization, "native_pred_planes") and cost_regularization._use_native(ref)
            if not native:
                raise RuntimeError("sharded_pred_stream drives the native RED plane pipeline (GPU tensors, no autograd)")
        acc = StreamingRegression(b, h, w, ref.device)
        flat, states, acc.state = handoff_buffers(cost_regularization, b, h, w, ref.device)   # fresh per tile: the previous tile's may still be in flight
        if rank > 0:
            _recv(flat, _global_rank(group, rank - 1), group)                   # one message per boundary and tile
        gen = isinstance(depth_values, GeneratedHeights)
        dv = depth_values if gen else depth_values.detach().to(torch.float32).contiguous()
        cost_regularization.native_pred_planes(features, proj_matrices, dv, geo_model, use_qc, states, acc.state, lo, hi)
        if rank < world - 1:
            pending.append(_isend(flat, _global_rank(group, rank + 1), group))
        finished.append(acc)
    for req, _keep in pending:
        req.wait()
    if world > 1:
        for a in finished:                                                      # the last rank holds every finished sum
            _broadcast(a.state, _global_rank(group, world - 1), group)
    out = []
    for a in finished:
        depth, confidence = a.result()
        out.append({"depth": depth, "photometric_confidence": confidence})
    return out


def handoff_views(flat, states):
    """The state views of a hand-off buffer, in order (shapes taken from `states`)."""
    o = flat.numel() - sum(int(t.numel()) for t in states)
    out = []
    for t in states:
        out.append(flat[o:o + t.numel()].view(t.shape))
        o += t.numel()
    return out


def _host_staged(t, group):
    """gloo moves only host memory point to point; RCCL ("nccl") takes device tensors as they are."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def _send(t, dst, group):
    dist.send(t.cpu() if _host_staged(t, group) else t, dst=dst, group=group)


def _isend(t, dst, group):
    """Non-blocking send; returns (request, buffer kept alive until the request completes)."""
    buf = t.cpu() if _host_staged(t, group) else t.contiguous()    # .cpu() waits for the producing kernels; RCCL orders itself behind the stream
    return dist.isend(buf, dst=dst, group=group), buf


def _broadcast(t, src, group):
    if _host_staged(t, group):
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def _recv(t, src, group):
    if _host_staged(t, group):
        h = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.recv(t, src=src, group=group)


def _global_rank(group, group_rank):
    if group is None:
        return group_rank
    return dist.get_global_rank(group, group_rank)
