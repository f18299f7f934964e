"""The reference's RPC warp + variance build written as a composite of stock PyTorch operators on device=cpu.

TEST INFRASTRUCTURE (like everything under oracle/): imported only by tests/ and by bench.py's cpu_baseline leg, never by
the product.  north_star asks for the MI355X throughput to be reported "next to the reference's CPU (torch, device=cpu)
path timed on the same box's host cores"; the reference's Python cannot travel to the GPU box, so this file restates its
operator sequence with the same torch calls on the same dtypes -- a (B, N, 20) float64 monomial tensor, two
`sum(coef * rpc[...])` quotients per projection, `F.grid_sample(bilinear, zeros)` on a grid normalised with the
align_corners=True formula, running sum / sum of squares over the views, variance -- so that what is timed
is the op mix the reference runs (mul 31 % / add 20 % / grid_sampler 16 % / pow 15 % per BASELINE.md section 2).

    /root/reference/modules/warping.py:183-207   monomials               -> _plh
    /root/reference/modules/warping.py:218-307   RPC_Obj2Photo / Photo2Obj -> _project
    /root/reference/modules/warping.py:310-365   rpc_warping             -> rpc_warping
    /root/reference/networks/casred.py:191-212   plane loop, variance    -> variance_planes

Checked against the C oracle by tests/test_oracle_golden.py::test_torch_composite_matches_oracle (bit-identical volumes).
"""
import torch
import torch.nn.functional as F

# offsets into the 170-vector (tools/RPCCore.py:8-28)
LINE_OFF, SAMP_OFF, LAT_OFF, LON_OFF, H_OFF, LINE_SCALE, SAMP_SCALE, LAT_SCALE, LON_SCALE, H_SCALE = range(10)
LNUM, LDEN, SNUM, SDEN, LATNUM, LATDEN, LONNUM, LONDEN = (10 + 20 * i for i in range(8))


def _plh(P, L, H):
    """(B,N) x 3 -> (B,N,20) monomials 1,L,P,H,LP,LH,PH,LL,PP,HH,PLH,LLL,LPP,LHH,LLP,PPP,PHH,LLH,PPH,HHH."""
    LP, LH, PH, LL, PP, HH = L * P, L * H, P * H, L * L, P * P, H * H
    return torch.stack([torch.ones_like(P), L, P, H, LP, LH, PH, LL, PP, HH, P * LH, L * LL, L * PP, L * HH, L * LP,
                        P * PP, P * HH, L * LH, P * PH, H * HH], dim=-1)


def _project(a, b, h, rpc, to_image):
    """to_image: (lat, lon, h) -> (samp, line); else (samp, line, h) -> (lat, lon).  a, b, h (B,N) float64; rpc (B,170)."""
    col = lambda i: rpc[:, i].view(-1, 1)  # noqa: E731
    if to_image:
        P, L = (a - col(LAT_OFF)) / col(LAT_SCALE), (b - col(LON_OFF)) / col(LON_SCALE)
        n0, d0, n1, d1, s0, o0, s1, o1 = SNUM, SDEN, LNUM, LDEN, SAMP_SCALE, SAMP_OFF, LINE_SCALE, LINE_OFF
    else:
        P, L = (a - col(SAMP_OFF)) / col(SAMP_SCALE), (b - col(LINE_OFF)) / col(LINE_SCALE)
        n0, d0, n1, d1, s0, o0, s1, o1 = LATNUM, LATDEN, LONNUM, LONDEN, LAT_SCALE, LAT_OFF, LON_SCALE, LON_OFF
    coef = _plh(P, L, (h - col(H_OFF)) / col(H_SCALE))
    poly = lambda k: torch.sum(coef * rpc[:, k:k + 20].view(-1, 1, 20), dim=-1)  # noqa: E731
    return poly(n0) / poly(d0) * col(s0) + col(o0), poly(n1) / poly(d1) * col(s1) + col(o1)


def rpc_warping(src_fea, src_rpc, ref_rpc, depth_values):
    """src_fea (B,C,H,W) float32, rpcs (B,170) float64, depth_values (B,D) or (B,D,H,W) -> (B,C,D,H,W)."""
    B, C, H, W = src_fea.shape
    D = depth_values.shape[1]
    with torch.no_grad():
        y, x = torch.meshgrid(torch.arange(H, dtype=torch.double), torch.arange(W, dtype=torch.double), indexing="ij")
        x = x.reshape(1, 1, H, W).expand(B, D, H, W).reshape(B, -1)
        y = y.reshape(1, 1, H, W).expand(B, D, H, W).reshape(B, -1)
        h = depth_values.view(B, D, 1, 1).expand(B, D, H, W) if depth_values.dim() == 2 else depth_values
        h = h.reshape(B, -1).double()
        lat, lon = _project(x, y, h, ref_rpc, False)
        samp, line = _project(lat, lon, h, src_rpc, True)
        gx = samp.float() / ((W - 1) / 2) - 1
        gy = line.float() / ((H - 1) / 2) - 1
        grid = torch.stack((gx.view(B, D, H * W), gy.view(B, D, H * W)), dim=3)
    out = F.grid_sample(src_fea, grid.view(B, D * H, W, 2), mode="bilinear", padding_mode="zeros", align_corners=False)
    return out.view(B, C, D, H, W)


def variance_planes(features, rpc, depth_values, d_begin=0, d_end=None):
    """Plane-at-a-time build like the pred loop: features = [ref, src...] (B,C,H,W); rpc (B,V,170), view 0 = reference;
    -> (B,C,d_end-d_begin,H,W) variance."""
    ref = features[0]
    V = len(features)
    D = depth_values.shape[1]
    d_end = D if d_end is None else d_end
    planes = []
    for d in range(d_begin, d_end):
        dv = depth_values[:, d:d + 1]
        vsum = ref.unsqueeze(2).clone()
        vsq = ref.unsqueeze(2) ** 2
        for s in range(1, V):
            w = rpc_warping(features[s], rpc[:, s], rpc[:, 0], dv)
            vsum = vsum + w
            vsq = vsq + w ** 2
        planes.append(vsq.div_(V).sub_(vsum.div_(V).pow_(2)))
    return torch.cat(planes, dim=2)
