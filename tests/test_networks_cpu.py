"""Host-side checks of the network mirrors (no GPU): parameter trees, names, shapes and seeded
initial values are identical to the reference's, so reference checkpoints load and the golden
cascade outputs (tests/golden/cascade.npz, generated with these seeds) can be reproduced on the GPU."""
import numpy as np
import pytest
import torch


def _build(tag, nd):
    from satmvs_amd.networks import casmvs, casred, ucs
    if tag == "red":
        return casred.CascadeREDNet("rpc", min_interval=2.5, ndepths=nd)
    if tag == "casmvs":
        return casmvs.CascadeMVSNet("rpc", min_interval=2.5, ndepths=nd)
    return ucs.UCSNet("rpc", stage_configs=nd)


@pytest.mark.parametrize("tag", ["red", "casmvs", "ucs"])
def test_parameter_tree_matches_reference(golden, tag):
    g = golden("cascade")
    nd = [int(v) for v in g["ndepths"]]
    torch.manual_seed(int(g[tag + ".seed"]))
    net = _build(tag, nd)
    sd = {k: v for k, v in net.state_dict().items() if "num_batches_tracked" not in k}
    assert list(sd.keys()) == [str(n) for n in g[tag + ".param_names"]]
    sums = np.array([[float(v.double().sum()), float((v.double() ** 2).sum())] for v in sd.values()])
    np.testing.assert_allclose(sums, g[tag + ".param_sums"], rtol=1e-12, atol=1e-12)


def test_train_and_infer_red_share_parameters():
    """RED train and pred classes share parameter names (checkpoint/resume)."""
    from satmvs_amd.networks import casred
    a = casred.CascadeREDNet("rpc", ndepths=[16, 8, 8])
    b = casred.Infer_CascadeREDNet("rpc", ndepths=[16, 8, 8])
    assert list(a.state_dict().keys()) == list(b.state_dict().keys())
    b.load_state_dict(a.state_dict())


def test_red_slice_step_matches_golden(golden):
    """slice_RED_Regularization forward (stock torch ops, CPU) against the reference's outputs with
    the exported weights -- pins the module mirror itself (modules/module.py:653-693)."""
    from satmvs_amd.modules.module import slice_RED_Regularization
    g = golden("red_pred")
    reg = slice_RED_Regularization(8, 8).eval()
    reg.load_state_dict({k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")})
    x = torch.from_numpy(g["slice_x"])
    B, _, H, W = x.shape
    st = reg.initial_states(B, H, W, "cpu")
    with torch.no_grad():
        r1 = reg(x, *st)
        r2 = reg(x * 0.5, *r1[1:])
    np.testing.assert_allclose(r1[0].numpy(), g["slice_out1"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r2[0].numpy(), g["slice_out2"], rtol=1e-5, atol=1e-6)
    for i, k in enumerate(["slice_s1", "slice_s2", "slice_s3", "slice_s4"]):
        np.testing.assert_allclose(r2[1 + i].numpy(), g[k], rtol=1e-5, atol=1e-6)


def _fake_replicate(m):
    """What torch.nn.parallel.replicate leaves behind (no GPU needed): `_parameters` emptied, the broadcast copies
    set as plain tensor attributes, buffers and sub-modules re-linked."""
    rep = {mod: mod._replicate_for_data_parallel() for mod in m.modules()}
    for mod, r in rep.items():
        for k, child in mod._modules.items():
            r._modules[k] = rep[child] if child is not None else None
        for k, p in mod._parameters.items():
            if p is not None:
                setattr(r, k, p.detach().clone())
        for k, b in mod._buffers.items():
            r._buffers[k] = b
    return rep[m]


@pytest.mark.parametrize("kind", ["red", "costreg", "featnet_unet", "featnet_fpn"])
def test_native_weight_lookup_works_on_dataparallel_replicas(kind):
    """nn.DataParallel (train.py:129, predict.py:85) calls replicas whose named_parameters() is EMPTY; the native
    paths must find their tensors by attribute path (KeyError 'conv_gru1.gate_conv.weight')."""
    from satmvs_amd.modules import module as M
    torch.manual_seed(3)
    if kind == "red":
        m, names = M.slice_RED_Regularization(32, 8), M._REDCore._PARAM_ORDER
    elif kind == "costreg":
        m = M.CostRegNet(32, 8)
        names = m._names()
    else:
        m = M.FeatureNet(base_channels=8, num_stage=3, arch_mode=kind.split("_")[1])
        names = m._names()
    r = _fake_replicate(m)
    assert len(dict(r.named_parameters())) == 0
    got = M._tensors_by_path(r, names)
    sd = dict(m.named_parameters())
    sd.update(dict(m.named_buffers()))
    for n, t in zip(names, got):
        assert torch.equal(t, sd[n]), n
    # decision helper: no gradient wanted under no_grad even though the parent's parameters require grad
    with torch.no_grad():
        assert not M._autograd_needed(torch.zeros(1), M._tensors_by_path(m, names))
    assert M._autograd_needed(torch.zeros(1), M._tensors_by_path(m, names))


def test_pack_cache_never_serves_a_freed_models_weights():
    """The packed-weight cache is keyed by (address, version); an entry must die with the storages it was packed from,
    because the allocator reuses a freed parameter's address for the next model."""
    from satmvs_amd.modules import module as M
    M._PACK_CACHE.clear()
    built = []

    def build_for(t):
        def build():
            built.append(float(t.sum()))
            return t.clone()
        return build
    a = torch.ones(8)
    p1 = M._packed("k", "cpu", [a], build_for(a))
    assert M._packed("k", "cpu", [a], build_for(a)) is p1 and len(built) == 1          # hit
    view = a.detach()                                                                  # DataParallel's device-0 replica
    assert M._packed("k", "cpu", [view], build_for(view)) is p1 and len(built) == 1    # same storage: hit
    key = next(iter(M._PACK_CACHE))
    del a, view                                                                        # model freed
    b = torch.full((8,), 2.0)
    M._PACK_CACHE[(key[0], key[1], (b.data_ptr(), b._version))] = M._PACK_CACHE.pop(key)   # simulate address reuse
    p2 = M._packed("k", "cpu", [b], build_for(b))
    assert len(built) == 2 and float(p2.sum()) == 16.0


def test_groupnorm1_is_a_drop_in_for_nn_groupnorm_on_the_host():
    """modules.module.GroupNorm1 (the ConvGRU cells' norm, native on the GPU): same parameters and state_dict keys as the
    reference's nn.GroupNorm(1, C, 1e-5, True) (module.py:15-20), same values on CPU tensors, activation argument included."""
    from satmvs_amd.modules.module import ConvGRUCell2, GroupNorm1
    torch.manual_seed(1)
    a, b = GroupNorm1(1, 8, 1e-5, True), torch.nn.GroupNorm(1, 8, 1e-5, True)
    assert list(a.state_dict()) == list(b.state_dict()) == ["weight", "bias"]
    with torch.no_grad():
        a.weight.uniform_(0.5, 1.5); a.bias.uniform_(-1, 1)
    b.load_state_dict(a.state_dict())
    x = torch.randn(2, 8, 5, 7)
    assert torch.equal(a(x), b(x))
    assert torch.equal(a(x, "sigmoid"), torch.sigmoid(b(x)))
    assert torch.equal(a(x, "tanh"), torch.tanh(b(x)))
    cell = ConvGRUCell2(4, 8, 3)
    assert [k for k in cell.state_dict() if "norm" in k] == ["reset_gate_norm.weight", "reset_gate_norm.bias", "update_gate_norm.weight",
                                                             "update_gate_norm.bias", "output_norm.weight", "output_norm.bias"]


def test_graphed_train_step_host_logic():
    """satmvs_amd.train_graph: argument signatures (what triggers a re-capture), nested copies into the static buffers, and the
    capturable-optimizer requirement -- the capture itself needs a GPU (tests/test_hip_end_to_end.py)."""
    from satmvs_amd import train_graph as tg
    a = {"stage1": torch.zeros(1, 2, 3), "stage2": [torch.zeros(4), torch.zeros(5, dtype=torch.float64)]}
    b = {"stage1": torch.ones(1, 2, 3), "stage2": [torch.ones(4), torch.ones(5, dtype=torch.float64)]}
    assert tg._signature((a, 3)) == tg._signature((b, 3))
    assert tg._signature((a,)) != tg._signature(({"stage1": torch.zeros(1, 2, 4), "stage2": a["stage2"]},))
    assert tg._signature((a,)) != tg._signature(({"stage1": a["stage1"], "stage2": [torch.zeros(4), torch.zeros(5)]},))
    static = tg._map(a, lambda t: t.clone())
    tg._zip_copy(static, b)
    assert float(static["stage1"].sum()) == 6.0 and float(static["stage2"][1].sum()) == 5.0 and float(a["stage1"].sum()) == 0.0
    lin = torch.nn.Linear(2, 2)
    with pytest.raises(ValueError, match="capturable"):
        tg.GraphedTrainStep(lin, torch.optim.RMSprop(lin.parameters(), lr=1e-3), lambda out: out.sum())
