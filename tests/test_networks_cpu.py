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
