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


_CASES = [
    ("conv_s1", 1, 32, 8, 8, 24, 48), ("conv_s1", 1, 8, 1, 8, 16, 72), ("conv_s1", 2, 5, 3, 3, 7, 33), ("conv_s1", 1, 16, 16, 4, 12, 130),
    ("conv_s1", 1, 64, 64, 2, 6, 12), ("conv_s1", 1, 1, 9, 1, 1, 1), ("conv_s1", 1, 8, 8, 5, 40, 64), ("conv_s1", 1, 32, 32, 4, 10, 40),
    ("conv_s1", 1, 8, 32, 4, 8, 36),
    ("conv_s2", 1, 8, 16, 8, 24, 48), ("conv_s2", 2, 3, 5, 2, 6, 66), ("conv_s2", 1, 32, 64, 4, 8, 16), ("conv_s2", 1, 16, 32, 6, 34, 130),
    ("convT_s2", 1, 64, 32, 2, 4, 8), ("convT_s2", 1, 16, 8, 4, 12, 24), ("convT_s2", 2, 3, 5, 1, 3, 33), ("convT_s2", 1, 32, 16, 3, 17, 65)]


def _float64_layer(kind, cin, cout, conv, x, gy):
    conv64 = _layer(kind, cin, cout).double()
    conv64.weight.data.copy_(conv.weight.detach().double().cpu())
    x64 = x.detach().double().cpu().requires_grad_(True)
    y64 = conv64(x64)
    y64.backward(gy.double().cpu())
    return y64.detach(), x64.grad, conv64.weight.grad


@pytest.mark.parametrize("kind,B,cin,cout,D,H,W", _CASES)
def test_conv3d_native_layer(dev, kind, B, cin, cout, D, H, W):
    """Every layer kind of CostRegNet through train_fns._conv3d -- forward (smvs_conv3d_fwd without the folded BatchNorm), input
    gradient (the adjoint layer on the same kernels) and weight gradient (smvs_conv3d_wgrad) -- against a float64 evaluation of the
    same layer on the CPU: 1e-5 of each tensor's scale (2e-5 for the weight gradient: float32 sums of up to a few 10^5 products,
    split over waves and added with float atomics)."""
    from satmvs_amd.modules import train_fns as T
    torch.manual_seed(B * 1000 + cin * 10 + cout + D)
    conv = _layer(kind, cin, cout).to(dev)
    x = torch.randn(B, cin, D, H, W, device=dev, requires_grad=True)
    y1 = T._conv3d(conv, x)
    assert y1.grad_fn is not None and "Conv3dNative" in type(y1.grad_fn).__name__
    gy = torch.randn_like(y1)
    y1.backward(gy)
    y64, dx64, dw64 = _float64_layer(kind, cin, cout, conv, x, gy)
    for got, ref, tol, name in ((y1.detach(), y64, 1e-5, "forward"), (x.grad, dx64, 1e-5, "input gradient"), (conv.weight.grad, dw64, 2e-5, "weight gradient")):
        got = got.double().cpu()
        assert got.shape == ref.shape
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= tol * scale, (kind, name, float((got - ref).abs().max()), scale)


@pytest.mark.parametrize("kind,B,cin,cout,D,H,W", _CASES[::3])
def test_conv3d_native_weight_gradient_alone(dev, kind, B, cin, cout, D, H, W):
    """SMVS_TRAIN_COMPOSITE_MASK bit 128: torch's forward and input gradient (bit-identical output) with the native weight gradient."""
    from satmvs_amd.modules import train_fns as T
    torch.manual_seed(B * 1000 + cin * 10 + cout + D)
    conv = _layer(kind, cin, cout).to(dev)
    x = torch.randn(B, cin, D, H, W, device=dev, requires_grad=True)
    saved = T.SW.train_composite_mask
    try:
        T.SW.train_composite_mask = 128
        y1 = T._conv3d(conv, x)
    finally:
        T.SW.train_composite_mask = saved
    assert y1.grad_fn is not None and "Conv3dWgrad" in type(y1.grad_fn).__name__
    with torch.no_grad():
        assert torch.equal(y1, conv(x))
    gy = torch.randn_like(y1)
    y1.backward(gy)
    _, dx64, dw64 = _float64_layer(kind, cin, cout, conv, x, gy)
    scale = float(dw64.abs().max())
    assert float((conv.weight.grad.double().cpu() - dw64).abs().max()) <= 2e-5 * scale
    assert float((x.grad.double().cpu() - dx64).abs().max()) <= 1e-4 * float(dx64.abs().max())


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


@pytest.mark.parametrize("kind,B,C,dims,relu", [("plain", 1, 8, (8, 24, 48), True), ("plain", 2, 5, (3, 7, 33), True), ("plain", 1, 64, (2, 6, 12), False),
                                               ("plain", 3, 16, (4, 12, 30), True), ("big", 1, 8, (8, 96, 192), True)])
def test_batchnorm3d_train_relu_native(dev, kind, B, C, dims, relu):
    """[relu](nn.BatchNorm3d(x)) in training form on smvs_batchnorm_train_fwd / _bwd against a float64 evaluation: output, input gradient,
    dgamma, dbeta within 2e-5 of their scales (1e-5 for the output), running statistics and num_batches_tracked updated like torch's."""
    from satmvs_amd.modules import train_fns as T
    torch.manual_seed(C + dims[0])
    bn = torch.nn.BatchNorm3d(C, momentum=0.1).to(dev).train()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5)
        bn.running_mean.uniform_(-1, 1); bn.running_var.uniform_(0.5, 2.0)
    bn64 = torch.nn.BatchNorm3d(C, momentum=0.1).double().train()
    bn64.load_state_dict({k: v.detach().double().cpu() if v.dtype.is_floating_point else v.cpu() for k, v in bn.state_dict().items()})
    x = (torch.randn(B, C, *dims, device=dev) * 2.0 + 0.7).requires_grad_(True)
    y = T._bn3d_relu(bn, x, relu)
    assert y is not None and "BatchNormRelu" in type(y.grad_fn).__name__
    gy = torch.randn_like(y)
    y.backward(gy)
    x64 = x.detach().double().cpu().requires_grad_(True)
    y64 = bn64(x64)
    if relu:
        y64 = torch.relu(y64)
    y64.backward(gy.double().cpu())
    pairs = [(y.detach(), y64.detach(), 1e-5, "output"), (x.grad, x64.grad, 2e-5, "input gradient"), (bn.weight.grad, bn64.weight.grad, 2e-5, "dgamma"),
             (bn.bias.grad, bn64.bias.grad, 2e-5, "dbeta"), (bn.running_mean, bn64.running_mean, 1e-6, "running_mean"),
             (bn.running_var, bn64.running_var, 1e-6, "running_var")]
    for got, ref, tol, name in pairs:
        scale = max(float(ref.abs().max()), 1e-30)
        assert float((got.detach().double().cpu() - ref).abs().max()) <= tol * scale, (name, float((got.detach().double().cpu() - ref).abs().max()), scale)
    assert int(bn.num_batches_tracked) == 1
    # elements the float32 and the float64 forward disagree on the sign of (|pre-activation| ~ 1e-7) move the gradients by dy at those
    # elements only; the 2e-5 above holds because such elements are a handful per 10^5


@pytest.mark.parametrize("cin,shape", [(32, (8, 16, 32)), (8, (8, 24, 72)), (16, (16, 32, 64))])
def test_costreg_training_step_native_vs_torch(dev, cin, shape):
    """CostRegNet.train() (batch-statistics BatchNorm3d, autograd): one forward + backward
      mask 0    every convolution (forward, input gradient, weight gradient) and every BatchNorm3d + ReLU native,
      mask 256  the convolutions native, torch's BatchNorm3d / ReLU,
      mask 384  torch's forward and input gradient with the native weight gradient (output bit-identical to torch's),
      mask 320  torch's own operators,
    each against a float64 evaluation of the same module on the CPU.  The native configurations must be as close to float64 as torch's
    operators are: output within 2e-5 of its scale, every gradient no farther from float64 than twice torch's distance (+ 2e-4 of the
    gradient's scale; ten batch-normalised layers and their ReLU masks amplify float32 round-off of either implementation to ~1e-3 of
    a weight gradient's scale on the coarse levels)."""
    import copy
    from satmvs_amd.modules import module as M
    torch.manual_seed(cin)
    net = M.CostRegNet(cin, 8).to(dev).train()
    state = copy.deepcopy(net.state_dict())
    vol0 = torch.rand(1, cin, *shape, device=dev)
    target = torch.randn(1, 1, *shape, device=dev)
    res = {}
    saved = M.SW.train_composite_mask
    try:
        for mask in (0, 256, 384, 320):
            M.SW.train_composite_mask = mask
            net.load_state_dict(state)                       # (the running statistics move with every training forward)
            net.zero_grad()
            vol = vol0.clone().requires_grad_(True)
            out = net(vol)
            ((out - target) ** 2).mean().backward()
            torch.cuda.synchronize()
            res[mask] = (out.detach().double().cpu(), vol.grad.double().cpu(), {n: p.grad.double().cpu() for n, p in net.named_parameters()},
                         {n: b.detach().double().cpu() for n, b in net.named_buffers() if b.dtype.is_floating_point})
    finally:
        M.SW.train_composite_mask = saved
    net64 = M.CostRegNet(cin, 8).double().train()
    net64.load_state_dict({k: (v.double() if v.dtype.is_floating_point else v).cpu() for k, v in state.items()})
    vol64 = vol0.double().cpu().requires_grad_(True)
    out64 = net64(vol64)
    ((out64 - target.double().cpu()) ** 2).mean().backward()
    g64 = {n: p.grad for n, p in net64.named_parameters()}
    b64 = {n: b.detach() for n, b in net64.named_buffers() if b.dtype.is_floating_point}

    def dist(got, ref):
        return float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)

    assert torch.equal(res[384][0], res[320][0])
    torch_out = dist(res[320][0], out64.detach())
    torch_gv = dist(res[320][1], vol64.grad)
    for mask in (0, 256, 384):
        o1, gv1, g1, b1 = res[mask]
        assert dist(o1, out64.detach()) <= max(2 * torch_out, 2e-5), (mask, dist(o1, out64.detach()), torch_out)
        assert dist(gv1, vol64.grad) <= 2 * torch_gv + 2e-4, (mask, dist(gv1, vol64.grad), torch_gv)
        for name, ref in g64.items():
            assert dist(g1[name], ref) <= 2 * dist(res[320][2][name], ref) + 2e-4, (mask, name, dist(g1[name], ref), dist(res[320][2][name], ref))
        for name, ref in b64.items():
            assert dist(b1[name], ref) <= 1e-5, (mask, name, dist(b1[name], ref))
