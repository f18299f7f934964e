// warp.hip -- the stand-alone warping operators and their autograd:
//   rpc_warping / rpc_warping_enisum   /root/reference/modules/warping.py:310-365, :139-178
//   homo_warping                       /root/reference/modules/warping.py:6-44
// plus the backward of the fused variance volume (costvol.hip) used by training.
// Same geometry, sampler and tile order as costvol.hip; one source view per launch.
#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

constexpr int TILE_X = 64, TILE_Y = 4;

struct WarpParams {
    const float* fea;        // fwd: src_fea (B,C,H,W)          bwd: grad_out (B,C,D,H,W)
    float* out;              // fwd: warped (B,C,D,H,W)          bwd: grad_src (B,C,H,W)
    const double* src_geo;   // rpc: src_rpc (B,170)             homography: proj (B,16)
    const double* ref_geo;   // rpc: ref_rpc (B,170)             homography: unused
    const float* depth;
    int B, C, D, H, W, depth_is_4d;
    int xt, yt, dct, dch;
};

template <int GEO>
__device__ __forceinline__ Tap voxel_tap(const WarpParams& p, cgeo_t sg, cgeo_t rg,
                                         const RpcInv& rn, const RpcInv& sn, int x, int y, double h,
                                         float half_wm1, float half_hm1)
{
    if (GEO == 0) {
        double lat, lon, samp, line;
        rpc_photo2obj(rg, rn, (double)x, (double)y, h, lat, lon);
        rpc_obj2photo(sg, sn, lat, lon, h, samp, line);
        return tap_from_pixel((float)samp, (float)line, p.H, p.W, half_wm1, half_hm1);
    } else {
        const double fx = (double)x, fy = (double)y;
        const double rx = fma(sg[1], fy, sg[0] * fx) + sg[2];
        const double ry = fma(sg[5], fy, sg[4] * fx) + sg[6];
        const double rz = fma(sg[9], fy, sg[8] * fx) + sg[10];
        const double X = fma(rx, h, sg[3]);
        const double Y = fma(ry, h, sg[7]);
        const double Z = fma(rz, h, sg[11]);
        const float gx = (float)((X / Z) / ((p.W - 1) * 0.5) - 1.0);
        const float gy = (float)((Y / Z) / ((p.H - 1) * 0.5) - 1.0);
        return tap_from_grid(gx, gy, p.H, p.W);
    }
}

template <int GEO, bool BWD>
__global__ __launch_bounds__(TILE_X * TILE_Y)
void warp_kernel(const WarpParams p)
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

    const cgeo_t sg0 = as_cgeo(p.src_geo + (size_t)b * (GEO == 0 ? RPC_LEN : 16));
    const cgeo_t rg0 = as_cgeo((GEO == 0) ? p.ref_geo + (size_t)b * RPC_LEN : p.src_geo);
    RpcInv rn, sn;
    if (GEO == 0) { rn = rpc_inv_image(rg0); sn = rpc_inv_ground(sg0); }

    BufRsrc rs;
    if (!BWD) rs = make_rsrc(p.fea + (size_t)b * C * HW, (uint32_t)C * (uint32_t)HW * 4u);

    for (int d = d0; d < d1; ++d) {
        const float hf = p.depth_is_4d ? p.depth[((size_t)b * D + d) * HW + pix] : p.depth[(size_t)b * D + d];
        const cgeo_t sg = launder(sg0), rg = launder(rg0);     // keep coefficient loads inside the plane loop
        const Tap t = voxel_tap<GEO>(p, sg, rg, rn, sn, x, y, (double)hf, half_wm1, half_hm1);
        if (!BWD) {
            float* o = p.out + (((size_t)b * C) * D + d) * HW + pix;
#pragma unroll 4
            for (int c = 0; c < C; ++c) o[(size_t)c * D * HW] = tap_fetch(rs, t, c * HW * 4);
        } else {
            // d warped / d src: the four weights, scattered (grid_sample backward w.r.t. input)
            const float* g = p.fea + (((size_t)b * C) * D + d) * HW + pix;
            float* gs = p.out + (size_t)b * C * HW;
            for (int c = 0; c < C; ++c) {
                const float gv = g[(size_t)c * D * HW];
                float* plane = gs + (size_t)c * HW;
                if (t.o_nw != SMVS_OOB) unsafeAtomicAdd(plane + (t.o_nw >> 2), gv * t.nw);
                if (t.o_ne != SMVS_OOB) unsafeAtomicAdd(plane + (t.o_ne >> 2), gv * t.ne);
                if (t.o_sw != SMVS_OOB) unsafeAtomicAdd(plane + (t.o_sw >> 2), gv * t.sw);
                if (t.o_se != SMVS_OOB) unsafeAtomicAdd(plane + (t.o_se >> 2), gv * t.se);
            }
        }
    }
}

static int warp_launch(int geo, bool bwd, const float* fea, float* out, const double* src_geo,
                       const double* ref_geo, const float* depth, int depth_is_4d,
                       int B, int C, int D, int H, int W, void* stream)
{
    if (!fea || !out || !src_geo || !depth || (geo == 0 && !ref_geo)) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || C < 1 || D < 1 || H < 1 || W < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if ((long long)C * H * W * 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "feature map larger than 2 GiB per batch item");
    WarpParams p{};
    p.fea = fea; p.out = out; p.src_geo = src_geo; p.ref_geo = ref_geo; p.depth = depth;
    p.B = B; p.C = C; p.D = D; p.H = H; p.W = W; p.depth_is_4d = (depth_is_4d & ~SMVS_CALL_ARITH_MASK) != 0;
    p.xt = (W + TILE_X - 1) / TILE_X;
    p.yt = (H + TILE_Y - 1) / TILE_Y;
    p.dch = D < 8 ? D : 8;
    p.dct = (D + p.dch - 1) / p.dch;
    const long long nb = (long long)p.xt * p.yt * p.dct * B;
    if (nb >= (1ll << 31)) return fail(SMVS_ERR_ARG, "grid too large");
    dim3 blk(TILE_X, TILE_Y), grd((unsigned)nb);
    hipStream_t st = (hipStream_t)stream;
    if (geo == 0 && !bwd) hipLaunchKernelGGL((warp_kernel<0, false>), grd, blk, 0, st, p);
    else if (geo == 0)    hipLaunchKernelGGL((warp_kernel<0, true>), grd, blk, 0, st, p);
    else if (!bwd)        hipLaunchKernelGGL((warp_kernel<1, false>), grd, blk, 0, st, p);
    else                  hipLaunchKernelGGL((warp_kernel<1, true>), grd, blk, 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "warp launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

// ---- src_proj @ inverse(ref_proj), warping.py:19 ------------------------------------------------------
__global__ void homo_compose_kernel(const double* __restrict__ src, const double* __restrict__ ref,
                                    double* __restrict__ out, int n)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            a[i][j] = ref[b * 16 + i * 4 + j];
            a[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    // Gauss-Jordan, partial pivoting (same elimination order as oracle/oracle.c)
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        for (int r = col + 1; r < 4; ++r)
            if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
        for (int j = 0; j < 8; ++j) { const double t = a[col][j]; a[col][j] = a[piv][j]; a[piv][j] = t; }
        const double inv = 1.0 / a[col][col];      // singular ref_proj -> inf/nan propagate, like torch.inverse on GPU
        for (int j = 0; j < 8; ++j) a[col][j] = a[col][j] * inv;
        for (int r = 0; r < 4; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
            for (int j = 0; j < 8; ++j) a[r][j] = a[r][j] - f * a[col][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s = s + src[b * 16 + i * 4 + k] * a[k][4 + j];
            out[b * 16 + i * 4 + j] = s;
        }
}

}  // namespace smvs

extern "C" {

SMVS_EXPORT int smvs_rpc_warp_fwd(const float* src_fea, const double* src_rpc, const double* ref_rpc,
                                  const float* depth, int depth_is_4d, float* out,
                                  int B, int C, int D, int H, int W, void* stream)
{
    return smvs::warp_launch(0, false, src_fea, out, src_rpc, ref_rpc, depth, depth_is_4d, B, C, D, H, W, stream);
}

SMVS_EXPORT int smvs_rpc_warp_bwd(const float* grad_out, const double* src_rpc, const double* ref_rpc,
                                  const float* depth, int depth_is_4d, float* grad_src,
                                  int B, int C, int D, int H, int W, void* stream)
{
    return smvs::warp_launch(0, true, grad_out, grad_src, src_rpc, ref_rpc, depth, depth_is_4d, B, C, D, H, W, stream);
}

SMVS_EXPORT int smvs_homo_warp_fwd(const float* src_fea, const double* proj, const float* depth, int depth_is_4d,
                                   float* out, int B, int C, int D, int H, int W, void* stream)
{
    return smvs::warp_launch(1, false, src_fea, out, proj, nullptr, depth, depth_is_4d, B, C, D, H, W, stream);
}

SMVS_EXPORT int smvs_homo_warp_bwd(const float* grad_out, const double* proj, const float* depth, int depth_is_4d,
                                   float* grad_src, int B, int C, int D, int H, int W, void* stream)
{
    return smvs::warp_launch(1, true, grad_out, grad_src, proj, nullptr, depth, depth_is_4d, B, C, D, H, W, stream);
}

SMVS_EXPORT int smvs_homo_compose(const double* src_proj, const double* ref_proj, double* out, int n, void* stream)
{
    if (!src_proj || !ref_proj || !out) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    if (n < 1) return smvs::fail(SMVS_ERR_ARG, "non-positive matrix count");
    hipLaunchKernelGGL(smvs::homo_compose_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       src_proj, ref_proj, out, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return smvs::fail(SMVS_ERR_LAUNCH, "homo_compose launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

}  // extern "C"
