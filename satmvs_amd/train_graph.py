"""One training step (reference: train.py:267-302 -- model forward, cas_mvsnet_loss, loss.backward(), optimizer.step())
captured in a HIP graph and replayed.

Why: the eager step is a chain of ~32 000 kernel launches (88 planes x the ConvGRU stack, forward and backward); at the
3-view 768x384 tile the GPU needs ~150 ms for them and the host ~310 ms to issue them (profiles/r03_train_step.txt).  A HIP
graph issues the whole step with one call: the step takes what the GPU takes.  The native operators are capturable as they
are -- they launch on torch's current stream, take their scratch from torch's allocator and never synchronise.

    step = GraphedTrainStep(model, optimizer, loss_fn)          # optimizer built with capturable=True
    for sample in loader:
        loss, outputs = step(sample["imgs"], sample["proj"], sample["depth_values"], sample["depth"], sample["mask"])

`loss_fn(outputs, *extra)` gets the model's output dict and the extra positional arguments of the call (ground truth, masks:
tensors or dicts / lists of tensors).  Arguments are copied into static buffers; a call with other shapes re-captures.
Returned tensors are the graph's static outputs: valid until the next call.  Python scalars the step reads are baked into the
graph.  The optimizer's are watched: every call compares the param groups' scalar hyper-parameters (lr, weight_decay, momentum,
alpha, eps, betas, ...) with what was captured and re-captures when one changed -- the reference's loop steps a MultiStepLR
every iteration (train.py:287), so a milestone costs one re-capture instead of silently training on at the old rate.  Scalars
inside loss_fn (loss weights) are the caller's: recapture() after changing them.
"""
from __future__ import annotations

import torch

from .modules.module import bump_param_epoch


def _map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(v, fn) for v in obj)
    return obj


def _zip_copy(dst, src):
    if torch.is_tensor(dst):
        dst.copy_(src, non_blocking=True)
    elif isinstance(dst, dict):
        for k in dst:
            _zip_copy(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s in zip(dst, src):
            _zip_copy(d, s)


def _hyper(optimizer):
    """Python-scalar hyper-parameters of every param group (tensors -- a capturable tensor lr -- live on the GPU and need no watch)."""
    out = []
    for g in optimizer.param_groups:
        out.append(tuple((k, v) for k, v in sorted(g.items())
                         if k != "params" and (isinstance(v, (int, float, bool, type(None))) or
                                               (isinstance(v, tuple) and all(isinstance(x, (int, float)) for x in v)))))
    return tuple(out)


def _signature(obj):
    if torch.is_tensor(obj):
        return (tuple(obj.shape), obj.dtype)
    if isinstance(obj, dict):
        return tuple((k, _signature(v)) for k, v in sorted(obj.items()))
    if isinstance(obj, (list, tuple)):
        return tuple(_signature(v) for v in obj)
    return obj


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, warmup=3):
        if not all(g.get("capturable", False) for g in optimizer.param_groups):
            raise ValueError("GraphedTrainStep needs an optimizer built with capturable=True (its step counters live on the GPU)")
        self.model, self.optimizer, self.loss_fn, self.warmup = model, optimizer, loss_fn, int(warmup)
        self._sig = None
        self._graph = None
        self._hyper_captured = None
        self.captures = 0                                     # how many times a graph was captured (shape / hyper-parameter changes)
        self._hyper_recaptures = 0                            # ... of which in a row because only a float hyper-parameter moved
        self._warned = False

    def _eager(self, args):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(*args[:3])
        loss = self.loss_fn(out, *args[3:])
        loss.backward()
        self.optimizer.step()
        return loss.detach(), _map(out, lambda t: t.detach())

    def _release(self):
        """Drop the captured graph and what was allocated from its private pool on its behalf: the packed-weight copies the native
        layers cached during the capture (modules/train_fns.py, modules/module.py) would otherwise pin that pool."""
        self._graph = None
        from .modules import module as _m
        from .modules import train_fns as _t
        # only the entries packed from THIS model's parameters (keyed on the source address; the weak reference names the storage):
        # other models and inference paths of the process keep theirs
        mine = {p.untyped_storage()._cdata for p in self.model.parameters()} | {b.untyped_storage()._cdata for b in self.model.buffers()}

        def from_model(ref):
            try:
                return (not ref.expired()) and ref.cdata in mine
            except Exception:                                 # noqa: BLE001 -- an unreadable reference: drop the entry
                return True
        for k in [k for k, v in _t._CONV_PACK.items() if from_model(v[2])]:
            del _t._CONV_PACK[k]
        with _m._PACK_CACHE_LOCK:
            for k in [k for k, v in _m._PACK_CACHE.items() if any(from_model(r) for r in v[1])]:
                del _m._PACK_CACHE[k]

    def _capture(self, args, warmup=None):
        import gc
        dev = args[0].device
        # Objects that own HIP resources (an earlier model's CUDAGraph and its private pool, streams, events) must not be finalised
        # by the cyclic garbage collector in the MIDDLE of the side-stream warm-up or of the capture: seen once in four runs of the GPU
        # suite as "Fatal Python error: Segmentation fault ... Garbage-collecting" inside the warm-up's first forward (round 6,
        # profiles/r06_gc_during_capture.txt).  torch.cuda.graph() itself collects before it captures; the warm-up runs before that, so
        # the same is done here, with the device idle, and the collector stays off until the graph is captured.
        torch.cuda.synchronize(dev)
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            self._capture_locked(args, warmup, dev)
        finally:
            if gc_was_on:
                gc.enable()

    def _capture_locked(self, args, warmup, dev):
        self._static = _map(args, lambda t: t.detach().clone())
        # the warm-up steps (allocator, MIOpen's find, lazily created optimizer state) must not train: parameters, buffers and
        # optimizer state are put back IN PLACE afterwards -- the graph holds their addresses
        params = [p for g in self.optimizer.param_groups for p in g["params"]]
        saved_p = [p.detach().clone() for p in params]
        saved_b = [b.detach().clone() for b in self.model.buffers()]
        saved_s = {id(t): t.detach().clone() for st in self.optimizer.state.values() for t in st.values() if torch.is_tensor(t)}
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self.warmup if warmup is None else warmup):
                self._eager(self._static)
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():
            for p, s in zip(params, saved_p):
                p.copy_(s)
            for b, s in zip(self.model.buffers(), saved_b):
                b.copy_(s)
            for st in self.optimizer.state.values():
                for t in st.values():
                    if torch.is_tensor(t):
                        if id(t) in saved_s:
                            t.copy_(saved_s[id(t)])
                        else:
                            t.zero_()                        # state created by the warm-up: torch's optimizers start from zeros
        torch.cuda.synchronize(dev)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._out = self._eager(self._static)
        self._hyper_captured = _hyper(self.optimizer)
        self.captures += 1

    def recapture(self):
        """Forget the captured graph: the next call captures again (after a change of learning rate, loss weights, ...)."""
        self._sig = None
        self._release()

    def __call__(self, *args):
        sig = _signature(args)
        hyper_only = sig == self._sig and self._graph is not None
        if sig != self._sig or _hyper(self.optimizer) != self._hyper_captured:
            self._release()                                   # the old graph (and its private pool) goes before capturing again
            # same shapes, only a float hyper-parameter moved (a per-iteration scheduler): allocator, MIOpen's choices and the optimizer
            # state are warm already -- one warm-up step re-creates the autograd graph's buffers, no more
            self._capture(args, warmup=1 if hyper_only else None)
            self._sig = sig
            self._hyper_recaptures = self._hyper_recaptures + 1 if hyper_only else 0
            if self._hyper_recaptures >= 3 and not self._warned:
                self._warned = True
                import warnings
                warnings.warn("GraphedTrainStep: re-captured %d times in a row because a float hyper-parameter of the optimizer changed "
                              "(a per-iteration learning-rate schedule?) -- every re-capture costs a warm-up step, a state restore and a "
                              "synchronize, far more than an eager step.  Give the optimizer a TENSOR lr (capturable: the graph reads it "
                              "from device memory, `optimizer.param_groups[i]['lr'].fill_(x)` needs no re-capture) or step the schedule "
                              "per epoch." % self._hyper_recaptures)
        else:
            self._hyper_recaptures = 0
            _zip_copy(self._static, args)
        self._graph.replay()
        bump_param_epoch()                                    # the replay stepped the parameters without touching autograd's version counters
        return self._out
