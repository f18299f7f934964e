"""Training path of the 3-D regulariser CostRegNet (--model casmvs / ucs): the native weight gradient of its 3x3x3 layers
(smvs_conv3d_wgrad, csrc/conv_wgrad.hip) behind satmvs_amd.modules.train_fns._conv3d, against float64 evaluations of the same
layers and against torch autograd of the whole module.  Reference: /root/reference/modules/module.py:324-410 (Conv3d / Deconv3d),
:546-577 (CostRegNet) under loss.backward() (/root/reference/train.py:284)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _layer(kind, cin, cout):
    if kind == "conv_s1":
        return torch.nn.Conv3d(cin, cout, 3, stride=1, padding=1, bias=False)
    if kind == "conv_s2":
        return torch.nn.Conv3d(cin, cout, 3, stride=2, padding=1, bias=False)
    return torch.nn.ConvTranspose3d(cin, cout, 3, stride=2, padding=1, output_padding=1, bias=False)


@pytest.mark.parametrize("kind,B,cin,cout,D,H,W", [
    ("conv_s1", 1, 32, 8, 8, 24, 48), ("conv_s1", 1, 8, 1, 8, 16, 72), ("conv_s1", 2, 5, 3, 3, 7, 33), ("conv_s1", 1, 16, 16, 4, 12, 130),
    ("conv_s1", 1, 64, 64, 2, 6, 12), ("conv_s1", 1, 1, 9, 1, 1, 1), ("conv_s1", 1, 8, 8, 5, 40, 64),
    ("conv_s2", 1, 8, 16, 8, 24, 48), ("conv_s2", 2, 3, 5, 2, 6, 66), ("conv_s2", 1, 32, 64, 4, 8, 16), ("conv_s2", 1, 16, 32, 6, 34, 130),
    ("convT_s2", 1, 64, 32, 2, 4, 8), ("convT_s2", 1, 16, 8, 4, 12, 24), ("convT_s2", 2, 3, 5, 1, 3, 33), ("convT_s2", 1, 32, 16, 3, 17, 65)])
def test_conv3d_native_weight_gradient(dev, kind, B, cin, cout, D, H, W):
    """Weight gradient of every layer kind of CostRegNet through _conv3d against a float64 evaluation of the same layer on the CPU
    (2e-5 of the gradient's scale: float32 sums of up to a few 10^5 products, split over waves and added with float atomics);
    forward bit-identical to torch's own (it IS torch's), input gradient torch's own."""
    from satmvs_amd.modules import train_fns as T
    torch.manual_seed(B * 1000 + cin * 10 + cout + D)
    conv = _layer(kind, cin, cout).to(dev)
    x = torch.randn(B, cin, D, H, W, device=dev, requires_grad=True)
    y1 = T._conv3d(conv, x)
    assert y1.grad_fn is not None and "Conv3dWgrad" in type(y1.grad_fn).__name__
    with torch.no_grad():
        assert torch.equal(y1, conv(x))
    gy = torch.randn_like(y1)
    y1.backward(gy)
    conv64 = _layer(kind, cin, cout).double()
    conv64.weight.data.copy_(conv.weight.detach().double().cpu())
    x64 = x.detach().double().cpu().requires_grad_(True)
    conv64(x64).backward(gy.double().cpu())
    ref = conv64.weight.grad
    got = conv.weight.grad.double().cpu()
    assert got.shape == ref.shape
    scale = float(ref.abs().max())
    assert float((got - ref).abs().max()) <= 2e-5 * scale, (kind, float((got - ref).abs().max()), scale)
    assert float((x.grad.double().cpu() - x64.grad).abs().max()) <= 1e-4 * float(x64.grad.abs().max())


def test_conv3d_wgrad_accumulates_and_rejects_bad_arguments(dev):
    """The C entry point accumulates into dw (two calls = twice the gradient) and answers bad arguments with SMVS_ERR_ARG."""
    from satmvs_amd import _lib
    torch.manual_seed(3)
    x = torch.randn(1, 4, 4, 8, 16, device=dev)
    gy = torch.randn(1, 6, 4, 8, 16, device=dev)
    dw = torch.zeros(6, 4, 3, 3, 3, device=dev)
    for _ in range(2):
        _lib.call("smvs_conv3d_wgrad", _lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), 1, 4, 6, 4, 8, 16, 1, _lib.current_stream(dev))
    w = torch.zeros(6, 4, 3, 3, 3, dtype=torch.float64, requires_grad=True)
    torch.nn.functional.conv3d(x.double().cpu(), w, padding=1).backward(gy.double().cpu())
    assert float((dw.double().cpu() - 2 * w.grad).abs().max()) <= 4e-5 * float(w.grad.abs().max())
    with pytest.raises(_lib.SatMVSNativeError):
        _lib.call("smvs_conv3d_wgrad", _lib.ptr(x), _lib.ptr(gy), _lib.ptr(dw), 1, 4, 6, 4, 8, 16, 3, _lib.current_stream(dev))
    with pytest.raises(_lib.SatMVSNativeError):
        _lib.call("smvs_conv3d_wgrad", None, _lib.ptr(gy), _lib.ptr(dw), 1, 4, 6, 4, 8, 16, 1, _lib.current_stream(dev))


@pytest.mark.parametrize("cin,shape", [(32, (8, 16, 32)), (8, (8, 24, 72))])
def test_costreg_training_step_native_weight_gradients(dev, cin, shape):
    """CostRegNet.train() (batch-statistics BatchNorm3d, autograd): one forward + backward with the native weight gradients against the
    same module on torch's own operators (SMVS_TRAIN_COMPOSITE_MASK bit 64) -- same output bits, every parameter gradient within 2e-4 of
    its scale, the gradient with respect to the variance volume torch's own."""
    from satmvs_amd.modules import module as M
    torch.manual_seed(cin)
    net = M.CostRegNet(cin, 8).to(dev).train()
    vol0 = torch.rand(1, cin, *shape, device=dev)
    target = torch.randn(1, 1, *shape, device=dev)
    res = []
    saved = M.SW.train_composite_mask
    try:
        for mask in (0, 64):
            M.SW.train_composite_mask = mask
            net.zero_grad()
            vol = vol0.clone().requires_grad_(True)
            out = net(vol)
            ((out - target) ** 2).mean().backward()
            torch.cuda.synchronize()
            res.append((out.detach().clone(), vol.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters()}))
    finally:
        M.SW.train_composite_mask = saved
    (o1, gv1, g1), (o0, gv0, g0) = res
    assert torch.equal(o1, o0)
    assert torch.allclose(gv1, gv0, rtol=1e-4, atol=1e-5 * float(gv0.abs().max()))
    for name, ref in g0.items():
        scale = float(ref.abs().max())
        assert float((g1[name] - ref).abs().max()) <= 2e-4 * scale + 1e-12, (name, float((g1[name] - ref).abs().max()), scale)
