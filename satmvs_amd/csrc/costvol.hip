// costvol.hip -- fused plane-sweep warp + per-channel variance cost volume (forward).
//
// Replaces, in one launch, the reference's per-source-view chain
//   rpc_warping / homo_warping  (/root/reference/modules/warping.py:310-365, :6-44)
//   + volume_sum / volume_sq_sum accumulation and the variance
//     (/root/reference/networks/casred.py:22-53 train, :191-212 pred; twins casmvs.py:26-59,
//      ucs.py:27-58)
// without ever materialising a warped volume or the (B, N, 20) float64 `coef` scratch.
//
// Work decomposition (B, D, H, W voxels; C channels; V views, view 0 = reference):
//   workgroup = 64 x 4 ref pixels (4 waves, one image row each) x a chunk of DCH planes;
//   a lane owns one ref pixel, loops over the chunk's planes, and for every plane
//     1. ref pixel + height -> ground (inverse RPC of the ref view), once            [float64]
//     2. ground -> each source image (direct RPC), float32 tap per source             [float64]
//     3. channel loop: 4 buffer loads per source (out-of-image taps return 0 from the
//        hardware range check), running sum / sum of squares, variance, one coalesced
//        256-B store per wave and channel.
//   RPC coefficients are wave-uniform and live in SGPRs (scalar loads), never in LDS or VGPRs.
//   Workgroup ids are remapped per XCD so that each XCD sweeps one band of ref rows: all plane
//   chunks of a band touch the same few source rows, which then stay in that XCD's 4 MiB L2.
//
// Algorithmic HBM bytes per voxel: 4*C (variance write) + 4 (height, if per-voxel) +
// 4*C*V/D (each feature map read once per D planes)  -- SURVEY.md section 8(d).
#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

constexpr int MAX_SRC = 7;          // V <= 8 views
constexpr int TILE_X = 64, TILE_Y = 4;

struct CostVolParams {
    const float* ref;               // (B,C,H,W)
    const float* src[MAX_SRC];      // (B,C,H,W) each
    const double* geo;              // rpc: (B,V,170); homography: (B,V-1,16) composed src @ inv(ref)
    const float* depth;             // (B,D,H,W) or (B,D)
    float* out;                     // (B,C,D_out,H,W)
    int B, V, C, D, H, W;
    int d_begin, d_end;             // planes built by this launch
    int D_out, d_out_off;           // plane d lands at index d - d_begin + d_out_off of `out`
    int depth_is_4d;
    int xt, yt, dct, dch;           // tiles in x, y; plane chunks; planes per chunk
};

template <int GEO, int NSRC, int CT>
__global__ __launch_bounds__(TILE_X * TILE_Y)
void costvol_fwd_kernel(const CostVolParams p)
{
    // ---- which tile am I (XCD-aware order: x tile fastest, then plane chunk, then row band) ------
    uint32_t L = xcd_remap(blockIdx.x, gridDim.x);
    const int xtile = L % p.xt; L /= p.xt;
    const int dchunk = L % p.dct; L /= p.dct;
    const int ytile = L % p.yt;
    const int b = L / p.yt;

    const int x = xtile * TILE_X + threadIdx.x;
    const int y = ytile * TILE_Y + threadIdx.y;
    if (x >= p.W || y >= p.H) return;

    const int H = p.H, W = p.W, C = (CT > 0) ? CT : p.C;
    const int HW = H * W;
    const int d0 = p.d_begin + dchunk * p.dch;
    const int d1 = min(d0 + p.dch, p.d_end);
    const int pix = y * W + x;

    BufRsrc rs[NSRC];
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
        rs[s] = make_rsrc(p.src[s] + (size_t)b * C * HW, (uint32_t)C * (uint32_t)HW * 4u);

    // ref feature of this pixel: plane-invariant, kept in registers when C is a compile-time size
    const float* refp = p.ref + (size_t)b * C * HW + pix;
    float refv[CT > 0 ? CT : 1];
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) refv[c] = refp[(size_t)c * HW];
    }

    const float fV = (float)p.V;
    const float rV = __fdiv_rn(1.0f, fV);
    const float half_wm1 = (float)((W - 1) * 0.5);
    const float half_hm1 = (float)((H - 1) * 0.5);

    // geometry constants
    const cgeo_t geo_b = as_cgeo((GEO == 0) ? p.geo + (size_t)b * p.V * RPC_LEN
                                            : p.geo + (size_t)b * (p.V - 1) * 16);
    RpcNorm ref_n;
    RpcNorm src_n[NSRC];
    if (GEO == 0) {
        ref_n = rpc_norm(geo_b);
#pragma unroll
        for (int s = 0; s < NSRC; ++s) src_n[s] = rpc_norm(geo_b + (size_t)(s + 1) * RPC_LEN);
    }
    const double fx = (double)x, fy = (double)y;

    float* outp = p.out + (size_t)b * C * p.D_out * HW + pix;

    for (int d = d0; d < d1; ++d) {
        const float hf = p.depth_is_4d ? p.depth[((size_t)b * p.D + d) * HW + pix]
                                       : p.depth[(size_t)b * p.D + d];
        const double h = (double)hf;

        // Launder the (wave-uniform) coefficient pointer once per plane: without this the compiler
        // hoists all 80*V loop-invariant scalar loads out of the plane loop and spills ~570 SGPRs
        // into VGPR lanes.  Re-issuing the s_loads per plane (scalar-cache hits) keeps the
        // coefficients in SGPRs only while they are used.
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
            // homo_warping, warping.py:28-38: rot.(x,y,1)*depth + trans, divide, normalise in float64
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const cgeo_t P = geo_d + s * 16;
                const double rx = fma(P[1], fy, P[0] * fx) + P[2];
                const double ry = fma(P[5], fy, P[4] * fx) + P[6];
                const double rz = fma(P[9], fy, P[8] * fx) + P[10];
                const double X = fma(rx, h, P[3]);
                const double Y = fma(ry, h, P[7]);
                const double Z = fma(rz, h, P[11]);
                const float gx = (float)((X / Z) / ((W - 1) * 0.5) - 1.0);
                const float gy = (float)((Y / Z) / ((H - 1) * 0.5) - 1.0);
                tap[s] = tap_from_grid(gx, gy, H, W);
            }
        }

        float* od = outp + (size_t)(d - p.d_begin + p.d_out_off) * HW;
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const float r = (CT > 0) ? refv[c] : refp[(size_t)c * HW];
            float sum = r;
            float sq = r * r;
            const int choff = c * HW * 4;
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const float wv = tap_fetch(rs[s], tap[s], choff);
                sum = sum + wv;
                sq = sq + wv * wv;
            }
            const float m = div_by_views(sum, fV, rV);
            const float q = div_by_views(sq, fV, rV);
            od[(size_t)c * p.D_out * HW] = q - m * m;
        }
    }
}

template <int GEO, int NSRC>
static hipError_t launch_ct(const CostVolParams& p, int nblocks, hipStream_t st)
{
    dim3 blk(TILE_X, TILE_Y);
    switch (p.C) {
    case 8:  hipLaunchKernelGGL((costvol_fwd_kernel<GEO, NSRC, 8>), dim3(nblocks), blk, 0, st, p); break;
    case 16: hipLaunchKernelGGL((costvol_fwd_kernel<GEO, NSRC, 16>), dim3(nblocks), blk, 0, st, p); break;
    case 32: hipLaunchKernelGGL((costvol_fwd_kernel<GEO, NSRC, 32>), dim3(nblocks), blk, 0, st, p); break;
    default: hipLaunchKernelGGL((costvol_fwd_kernel<GEO, NSRC, 0>), dim3(nblocks), blk, 0, st, p); break;
    }
    return hipGetLastError();
}

template <int GEO>
static hipError_t launch_nsrc(const CostVolParams& p, int nblocks, hipStream_t st)
{
    switch (p.V - 1) {
    case 1: return launch_ct<GEO, 1>(p, nblocks, st);
    case 2: return launch_ct<GEO, 2>(p, nblocks, st);
    case 3: return launch_ct<GEO, 3>(p, nblocks, st);
    case 4: return launch_ct<GEO, 4>(p, nblocks, st);
    case 5: return launch_ct<GEO, 5>(p, nblocks, st);
    case 6: return launch_ct<GEO, 6>(p, nblocks, st);
    case 7: return launch_ct<GEO, 7>(p, nblocks, st);
    default: return hipErrorInvalidValue;
    }
}

static int costvol_fwd(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                       const double* geo, const float* depth, int depth_is_4d, float* out,
                       int B, int C, int D, int H, int W, int d_begin, int d_end, int D_out, int d_out_off,
                       void* stream)
{
    if (!ref_fea || !src_fea || !geo || !depth || !out) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (n_src < 1 || n_src > MAX_SRC) return fail(SMVS_ERR_ARG, "n_src must be in [1,7] (2..8 views), got %d", n_src);
    if (B < 1 || C < 1 || D < 1 || H < 1 || W < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if (d_begin < 0 || d_end > D || d_begin > d_end) return fail(SMVS_ERR_ARG, "bad plane range [%d,%d) of %d", d_begin, d_end, D);
    if (d_out_off < 0 || d_out_off + (d_end - d_begin) > D_out) return fail(SMVS_ERR_ARG, "plane range does not fit the output (%d planes at %d of %d)", d_end - d_begin, d_out_off, D_out);
    if ((long long)C * H * W * 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "feature map larger than 2 GiB per batch item");
    for (int s = 0; s < n_src; ++s)
        if (!src_fea[s]) return fail(SMVS_ERR_ARG, "null source feature pointer %d", s);
    if (d_begin == d_end) return SMVS_OK;

    CostVolParams p{};
    p.ref = ref_fea;
    for (int s = 0; s < n_src; ++s) p.src[s] = src_fea[s];
    p.geo = geo; p.depth = depth; p.out = out;
    p.B = B; p.V = n_src + 1; p.C = C; p.D = D; p.H = H; p.W = W;
    p.d_begin = d_begin; p.d_end = d_end; p.D_out = D_out; p.d_out_off = d_out_off;
    p.depth_is_4d = depth_is_4d;
    p.xt = (W + TILE_X - 1) / TILE_X;
    p.yt = (H + TILE_Y - 1) / TILE_Y;
    const int nd = d_end - d_begin;
    p.dch = nd < 8 ? nd : 8;
    p.dct = (nd + p.dch - 1) / p.dch;
    const long long nblocks = (long long)p.xt * p.yt * p.dct * B;
    if (nblocks >= (1ll << 31)) return fail(SMVS_ERR_ARG, "grid too large");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = (geo_kind == 0) ? launch_nsrc<0>(p, (int)nblocks, st) : launch_nsrc<1>(p, (int)nblocks, st);
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "costvol_fwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

}  // namespace smvs

extern "C" {

SMVS_EXPORT int smvs_rpc_costvol_fwd(const float* ref_fea, const float* const* src_fea, int n_src,
                                     const double* rpc, const float* depth, int depth_is_4d, float* out_var,
                                     int B, int C, int D, int H, int W,
                                     int d_begin, int d_end, int D_out, int d_out_off, void* stream)
{
    return smvs::costvol_fwd(0, ref_fea, src_fea, n_src, rpc, depth, depth_is_4d, out_var,
                             B, C, D, H, W, d_begin, d_end, D_out, d_out_off, stream);
}

SMVS_EXPORT int smvs_homo_costvol_fwd(const float* ref_fea, const float* const* src_fea, int n_src,
                                      const double* proj, const float* depth, int depth_is_4d, float* out_var,
                                      int B, int C, int D, int H, int W,
                                      int d_begin, int d_end, int D_out, int d_out_off, void* stream)
{
    return smvs::costvol_fwd(1, ref_fea, src_fea, n_src, proj, depth, depth_is_4d, out_var,
                             B, C, D, H, W, d_begin, d_end, D_out, d_out_off, stream);
}

}  // extern "C"
