// costvol_bwd.hip -- backward of the fused variance volume (costvol.hip) for training:
// loss.backward() through /root/reference/networks/casred.py:22-53 (train.py:284), i.e. through
//   var = sq/V - (sum/V)^2,  sum = ref + sum_s warped_s,  sq = ref^2 + sum_s warped_s^2
// and through grid_sample w.r.t. the source features (the grid itself is built under no_grad,
// /root/reference/modules/warping.py:322).
//   d var / d warped_s = (2/V) (warped_s - sum/V)      d var / d ref = (2/V) (ref - sum/V)
// Taps are recomputed (float64 RPC chain) instead of saving 2.4 GB of warped volumes; gradients
// are scattered with hardware float32 atomics, like torch's grid_sampler backward.
#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

constexpr int MAX_SRC = 7;
constexpr int TILE_X = 64, TILE_Y = 4;

struct CostVolBwdParams {
    const float* grad_var;          // (B,C,D,H,W)
    const float* ref;               // (B,C,H,W)
    const float* src[MAX_SRC];
    float* grad_ref;                // (B,C,H,W), accumulated
    float* grad_src[MAX_SRC];       // (B,C,H,W), accumulated
    const double* geo;
    const float* depth;
    int B, V, C, D, H, W, depth_is_4d;
    int xt, yt, dct, dch;
};

template <int GEO, int NSRC>
__global__ __launch_bounds__(TILE_X * TILE_Y)
void costvol_bwd_kernel(const CostVolBwdParams p)
{
    uint32_t L = xcd_remap(blockIdx.x, gridDim.x);
    const int xtile = L % p.xt; L /= p.xt;
    const int dchunk = L % p.dct; L /= p.dct;
    const int ytile = L % p.yt;
    const int b = L / p.yt;
    const int x = xtile * TILE_X + threadIdx.x;
    const int y = ytile * TILE_Y + threadIdx.y;
    if (x >= p.W || y >= p.H) return;

    const int H = p.H, W = p.W, C = p.C, D = p.D;
    const int HW = H * W;
    const int pix = y * W + x;
    const int d0 = dchunk * p.dch, d1 = min(d0 + p.dch, D);
    const float half_wm1 = (float)((W - 1) * 0.5), half_hm1 = (float)((H - 1) * 0.5);
    const float fV = (float)p.V, rV = __fdiv_rn(1.0f, fV), two_over_v = 2.0f / fV;

    BufRsrc rs[NSRC];
#pragma unroll
    for (int s = 0; s < NSRC; ++s) rs[s] = make_rsrc(p.src[s] + (size_t)b * C * HW, (uint32_t)C * (uint32_t)HW * 4u);

    const cgeo_t geo_b = as_cgeo((GEO == 0) ? p.geo + (size_t)b * p.V * RPC_LEN : p.geo + (size_t)b * (p.V - 1) * 16);
    RpcInv ref_n, src_n[NSRC];
    if (GEO == 0) {
        ref_n = rpc_inv_image(geo_b);
#pragma unroll
        for (int s = 0; s < NSRC; ++s) src_n[s] = rpc_inv_ground(geo_b + (size_t)(s + 1) * RPC_LEN);
    }
    const double fx = (double)x, fy = (double)y;
    const float* refp = p.ref + (size_t)b * C * HW + pix;
    float* grefp = p.grad_ref + (size_t)b * C * HW + pix;

    for (int d = d0; d < d1; ++d) {
        const float hf = p.depth_is_4d ? p.depth[((size_t)b * D + d) * HW + pix] : p.depth[(size_t)b * D + d];
        const double h = (double)hf;
        const cgeo_t geo_d = launder(geo_b);
        Tap tap[NSRC];
        if (GEO == 0) {
            double lat, lon;
            rpc_photo2obj(geo_d, ref_n, fx, fy, h, lat, lon);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                double samp, line;
                rpc_obj2photo(geo_d + (size_t)(s + 1) * RPC_LEN, src_n[s], lat, lon, h, samp, line);
                tap[s] = tap_from_pixel((float)samp, (float)line, H, W, half_wm1, half_hm1);
            }
        } else {
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const cgeo_t P = geo_d + s * 16;
                const double rx = fma(P[1], fy, P[0] * fx) + P[2];
                const double ry = fma(P[5], fy, P[4] * fx) + P[6];
                const double rz = fma(P[9], fy, P[8] * fx) + P[10];
                const double X = fma(rx, h, P[3]), Y = fma(ry, h, P[7]), Z = fma(rz, h, P[11]);
                const float gx = (float)((X / Z) / ((W - 1) * 0.5) - 1.0);
                const float gy = (float)((Y / Z) / ((H - 1) * 0.5) - 1.0);
                tap[s] = tap_from_grid(gx, gy, H, W);
            }
        }
        const float* gp = p.grad_var + (((size_t)b * C) * D + d) * HW + pix;
        for (int c = 0; c < C; ++c) {
            const float g = gp[(size_t)c * D * HW] * two_over_v;
            const float r = refp[(size_t)c * HW];
            float wv[NSRC];
            float sum = r;
#pragma unroll
            for (int s = 0; s < NSRC; ++s) { wv[s] = tap_fetch(rs[s], tap[s], c * HW * 4); sum = sum + wv[s]; }
            const float m = div_by_views(sum, fV, rV);
            unsafeAtomicAdd(grefp + (size_t)c * HW, g * (r - m));
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const float gw = g * (wv[s] - m);
                float* plane = p.grad_src[s] + ((size_t)b * C + c) * HW;
                const Tap& t = tap[s];
                if (t.o_nw != SMVS_OOB) unsafeAtomicAdd(plane + (t.o_nw >> 2), gw * t.nw);
                if (t.o_ne != SMVS_OOB) unsafeAtomicAdd(plane + (t.o_ne >> 2), gw * t.ne);
                if (t.o_sw != SMVS_OOB) unsafeAtomicAdd(plane + (t.o_sw >> 2), gw * t.sw);
                if (t.o_se != SMVS_OOB) unsafeAtomicAdd(plane + (t.o_se >> 2), gw * t.se);
            }
        }
    }
}

template <int GEO>
static hipError_t launch_bwd(const CostVolBwdParams& p, unsigned nb, hipStream_t st)
{
    dim3 blk(TILE_X, TILE_Y), grd(nb);
    switch (p.V - 1) {
    case 1: hipLaunchKernelGGL((costvol_bwd_kernel<GEO, 1>), grd, blk, 0, st, p); break;
    case 2: hipLaunchKernelGGL((costvol_bwd_kernel<GEO, 2>), grd, blk, 0, st, p); break;
    case 3: hipLaunchKernelGGL((costvol_bwd_kernel<GEO, 3>), grd, blk, 0, st, p); break;
    case 4: hipLaunchKernelGGL((costvol_bwd_kernel<GEO, 4>), grd, blk, 0, st, p); break;
    case 5: hipLaunchKernelGGL((costvol_bwd_kernel<GEO, 5>), grd, blk, 0, st, p); break;
    case 6: hipLaunchKernelGGL((costvol_bwd_kernel<GEO, 6>), grd, blk, 0, st, p); break;
    case 7: hipLaunchKernelGGL((costvol_bwd_kernel<GEO, 7>), grd, blk, 0, st, p); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace smvs

extern "C" SMVS_EXPORT int smvs_costvol_bwd(int geo_kind, const float* grad_var, const float* ref_fea,
                                            const float* const* src_fea, int n_src, const double* geo,
                                            const float* depth, int depth_is_4d, float* grad_ref,
                                            float* const* grad_src, int B, int C, int D, int H, int W, void* stream)
{
    using namespace smvs;
    if (!grad_var || !ref_fea || !src_fea || !geo || !depth || !grad_ref || !grad_src) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (geo_kind != 0 && geo_kind != 1) return fail(SMVS_ERR_ARG, "geo_kind must be 0 (rpc) or 1 (homography)");
    if (n_src < 1 || n_src > MAX_SRC) return fail(SMVS_ERR_ARG, "n_src must be in [1,7], got %d", n_src);
    if (B < 1 || C < 1 || D < 1 || H < 1 || W < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if ((long long)C * H * W * 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "feature map larger than 2 GiB per batch item");
    CostVolBwdParams p{};
    p.grad_var = grad_var; p.ref = ref_fea; p.grad_ref = grad_ref; p.geo = geo; p.depth = depth;
    for (int s = 0; s < n_src; ++s) {
        if (!src_fea[s] || !grad_src[s]) return fail(SMVS_ERR_ARG, "null source pointer %d", s);
        p.src[s] = src_fea[s]; p.grad_src[s] = grad_src[s];
    }
    p.B = B; p.V = n_src + 1; p.C = C; p.D = D; p.H = H; p.W = W; p.depth_is_4d = depth_is_4d;
    p.xt = (W + TILE_X - 1) / TILE_X; p.yt = (H + TILE_Y - 1) / TILE_Y;
    p.dch = D < 8 ? D : 8; p.dct = (D + p.dch - 1) / p.dch;
    const long long nb = (long long)p.xt * p.yt * p.dct * B;
    if (nb >= (1ll << 31)) return fail(SMVS_ERR_ARG, "grid too large");
    hipError_t e = geo_kind == 0 ? launch_bwd<0>(p, (unsigned)nb, (hipStream_t)stream)
                                 : launch_bwd<1>(p, (unsigned)nb, (hipStream_t)stream);
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "costvol_bwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}
