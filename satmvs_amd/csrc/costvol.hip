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
#include <limits.h>
#include <stdlib.h>

#include <type_traits>

#include "smvs_device.h"
#include "smvs_host.h"

#include <math.h>

#include <atomic>

#include "costvol_kernels.h"

namespace smvs {

// DEFAULT arithmetic of the variance build (smvs_set_arith, include/satmvs.h), for calls that do not carry their own
// (SMVS_CALL_ARITH_* in depth_is_4d / smvs_height_gen.arith); read once per call.
static std::atomic<int> g_arith{SMVS_ARITH_FUSED};

hipError_t launch_costvol_fused(int geo_kind, const CostVolParams& p, hipStream_t st);     // costvol_fused.hip

// Folds every source view's four direct cubics at the normalised height of every plane (smvs_device.h, "plane-constant
// heights"): one thread per (batch item, plane, source, cubic).  The plane's height is depth[b, d] or, for a (B,D,H,W)
// tensor, the height of the plane's first pixel -- the consuming kernel checks its own heights against record[0].
__global__ __launch_bounds__(64)
void rpc_plane_coef_kernel(const double* __restrict__ rpc, const float* __restrict__ depth, int is4d, double* __restrict__ pc,
                           int B, int n_src, int D, size_t HW, int d_begin, int nd)
{
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= B * nd * n_src * 4) return;
    const int i = idx & 3, s = (idx >> 2) % n_src, bp = idx / (4 * n_src), b = bp / nd, bd = b * D + d_begin + (bp - b * nd);
    const float hf = is4d ? depth[(size_t)bd * HW] : depth[bd];
    const double* r = rpc + ((size_t)b * (n_src + 1) + s + 1) * RPC_LEN;
    const double Hn = ((double)hf - r[I_H_OFF]) * recip_scale(r[I_H_SCALE]);       // as o2p_xn normalises it (the kernel's own reciprocal)
    const int base = i == 0 ? I_SNUM : i == 1 ? I_SDEN : i == 2 ? I_LNUM : I_LDEN;
    const double* c = r + base;
    double* o = pc + pc_header_doubles((size_t)B * D, B) + pc_offset(b, s, i, bd - b * D, n_src, D);
    o[0] = fma(Hn, fma(Hn, fma(Hn, c[19], c[9]), c[3]), c[0]);
    o[1] = fma(Hn, fma(Hn, c[13], c[5]), c[1]);
    o[2] = fma(Hn, fma(Hn, c[16], c[6]), c[2]);
    o[3] = fma(Hn, c[10], c[4]);
    o[4] = fma(Hn, c[17], c[7]);
    o[5] = fma(Hn, c[18], c[8]);
    if (s == 0 && i == 0) pc[bd] = (double)hf;
    // the views' reciprocal scales, once per batch item (by the threads of the window's first plane): source view s by cubic lane 0, the
    // ref view by (s, i) = (0, 1)
    if (bp - b * nd == 0 && (i == 0 || (s == 0 && i == 1))) {
        const bool refv = i == 1;
        const double* rv = refv ? rpc + (size_t)b * (n_src + 1) * RPC_LEN : r;
        double* sc = pc + pc_heights_doubles((size_t)B * D) + (size_t)b * PC_SCALES + (refv ? 0 : 3 * (s + 1));
        sc[0] = recip_scale(rv[refv ? I_SAMP_SCALE : I_LAT_SCALE]);
        sc[1] = recip_scale(rv[refv ? I_LINE_SCALE : I_LON_SCALE]);
        sc[2] = recip_scale(rv[I_H_SCALE]);
    }
}

static int costvol_fwd(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                       const double* geo, const float* depth, int depth_is_4d, const smvs_height_gen* gen, float* out,
                       int B, int C, int D, int H, int W, int d_begin, int d_end, int D_out, int d_out_off,
                       void* stream, const double* plane_coef = nullptr)
{
    if (!ref_fea || !src_fea || !geo || (!depth && !gen) || !out) return fail(SMVS_ERR_ARG, "null pointer argument");
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
    p.geo = geo; p.depth = depth; p.out = out; p.pc = plane_coef;
    p.B = B; p.V = n_src + 1; p.C = C; p.D = D; p.H = H; p.W = W;
    p.d_begin = d_begin; p.d_end = d_end; p.D_out = D_out; p.d_out_off = d_out_off;
    // arithmetic of this call: the bits the call carries (depth_is_4d / gen->arith), else the process default
    const int call = (gen ? gen->arith : depth_is_4d) & SMVS_CALL_ARITH_MASK;
    if (call == SMVS_CALL_ARITH_MASK) return fail(SMVS_ERR_ARG, "both SMVS_CALL_ARITH_EXACT and SMVS_CALL_ARITH_FUSED set");
    const int arith = call == SMVS_CALL_ARITH_EXACT ? SMVS_ARITH_EXACT : call == SMVS_CALL_ARITH_FUSED ? SMVS_ARITH_FUSED
                      : g_arith.load(std::memory_order_relaxed);
    p.depth_is_4d = (depth_is_4d & ~SMVS_CALL_ARITH_MASK) ? HEIGHT_TENSOR : HEIGHT_PLANES;
    if (gen) {
        HeightGenHost hh;
        if (const char* msg = height_gen_check(gen, D, H, W, hh)) return fail(SMVS_ERR_ARG, "%s", msg);
        p.depth_is_4d = HEIGHT_GENERATED;
        p.hg.prev = hh.prev; p.hg.hp = hh.hp; p.hg.wp = hh.wp; p.hg.ih = hh.ih; p.hg.iw = hh.iw; p.hg.scale = hh.scale;
        p.hg.c = hh.c; p.hg.ndm1 = hh.ndm1; p.hg.var = hh.var; p.hg.rmin = hh.rmin; p.hg.rmax = hh.rmax;
    }
    p.rV = 1.0f / (float)(n_src + 1);
    p.kw = n_src == 1 ? 0.5f : n_src == 2 ? (float)(sqrt(2.0) / 3.0) : (float)(1.0 / sqrt((double)(n_src + 1)));
    p.r_half_wm1 = 1.0f / (float)((W - 1) * 0.5);
    p.r_half_hm1 = 1.0f / (float)((H - 1) * 0.5);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = arith == SMVS_ARITH_FUSED ? launch_costvol_fused(geo_kind, p, st)
                   : (geo_kind == 0) ? launch_nsrc<0, AR_EXACT>(p, st) : launch_nsrc<1, AR_EXACT>(p, st);
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "costvol_fwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

}  // namespace smvs

extern "C" {

#ifdef SMVS_TIMING
SMVS_EXPORT int smvs_debug_timing(unsigned long long* out12, int reset)
{
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(smvs::smvs_timing), 96) != hipSuccess) return 1;
    if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(smvs::smvs_timing), z, 96) != hipSuccess) return 1; }
    return 0;
}
#endif

SMVS_EXPORT int smvs_set_arith(int mode)
{
    if (mode != SMVS_ARITH_EXACT && mode != SMVS_ARITH_FUSED) return -1;
    return smvs::g_arith.exchange(mode);
}

SMVS_EXPORT int smvs_get_arith(void) { return smvs::g_arith.load(); }

SMVS_EXPORT int smvs_rpc_costvol_fwd(const float* ref_fea, const float* const* src_fea, int n_src,
                                     const double* rpc, const float* depth, int depth_is_4d, float* out_var,
                                     int B, int C, int D, int H, int W,
                                     int d_begin, int d_end, int D_out, int d_out_off, void* stream)
{
    return smvs::costvol_fwd(0, ref_fea, src_fea, n_src, rpc, depth, depth_is_4d, nullptr, out_var,
                             B, C, D, H, W, d_begin, d_end, D_out, d_out_off, stream);
}

SMVS_EXPORT size_t smvs_rpc_plane_coef_bytes(int B, int n_src, int D)
{
    if (B < 1 || D < 1 || n_src < 1 || n_src > smvs::MAX_SRC) return 0;
    return smvs::pc_total_doubles(B, n_src, D) * sizeof(double);
}

SMVS_EXPORT int smvs_rpc_plane_coef(const double* rpc, const float* depth, int depth_is_4d, double* plane_coef,
                                    int B, int n_src, int D, int H, int W, int d_begin, int d_end, void* stream)
{
    using namespace smvs;
    if (!rpc || !depth || !plane_coef) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (n_src < 1 || n_src > MAX_SRC) return fail(SMVS_ERR_ARG, "n_src must be in [1,7] (2..8 views), got %d", n_src);
    if (B < 1 || D < 1 || H < 1 || W < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if (d_begin < 0 || d_end > D || d_begin > d_end) return fail(SMVS_ERR_ARG, "bad plane range [%d,%d) of %d", d_begin, d_end, D);
    if ((long long)B * D * n_src * 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "too many planes");
    if (d_begin == d_end) return SMVS_OK;
    const int n = B * (d_end - d_begin) * n_src * 4;
    hipLaunchKernelGGL(rpc_plane_coef_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, rpc, depth,
                       (depth_is_4d & ~SMVS_CALL_ARITH_MASK) ? 1 : 0, plane_coef, B, n_src, D, (size_t)H * W, d_begin, d_end - d_begin);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "rpc_plane_coef launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

SMVS_EXPORT int smvs_rpc_costvol_fwd_pc(const float* ref_fea, const float* const* src_fea, int n_src,
                                        const double* rpc, const float* depth, int depth_is_4d, const double* plane_coef,
                                        float* out_var, int B, int C, int D, int H, int W,
                                        int d_begin, int d_end, int D_out, int d_out_off, void* stream)
{
    return smvs::costvol_fwd(0, ref_fea, src_fea, n_src, rpc, depth, depth_is_4d, nullptr, out_var,
                             B, C, D, H, W, d_begin, d_end, D_out, d_out_off, stream, plane_coef);
}

SMVS_EXPORT int smvs_rpc_costvol_fwd_gen(const float* ref_fea, const float* const* src_fea, int n_src,
                                         const double* rpc, const smvs_height_gen* gen, float* out_var,
                                         int B, int C, int D, int H, int W,
                                         int d_begin, int d_end, int D_out, int d_out_off, void* stream)
{
    return smvs::costvol_fwd(0, ref_fea, src_fea, n_src, rpc, nullptr, 0, gen, out_var,
                             B, C, D, H, W, d_begin, d_end, D_out, d_out_off, stream);
}

SMVS_EXPORT int smvs_homo_costvol_fwd(const float* ref_fea, const float* const* src_fea, int n_src,
                                      const double* proj, const float* depth, int depth_is_4d, float* out_var,
                                      int B, int C, int D, int H, int W,
                                      int d_begin, int d_end, int D_out, int d_out_off, void* stream)
{
    return smvs::costvol_fwd(1, ref_fea, src_fea, n_src, proj, depth, depth_is_4d, nullptr, out_var,
                             B, C, D, H, W, d_begin, d_end, D_out, d_out_off, stream);
}

SMVS_EXPORT int smvs_homo_costvol_fwd_gen(const float* ref_fea, const float* const* src_fea, int n_src,
                                          const double* proj, const smvs_height_gen* gen, float* out_var,
                                          int B, int C, int D, int H, int W,
                                          int d_begin, int d_end, int D_out, int d_out_off, void* stream)
{
    return smvs::costvol_fwd(1, ref_fea, src_fea, n_src, proj, nullptr, 0, gen, out_var,
                             B, C, D, H, W, d_begin, d_end, D_out, d_out_off, stream);
}

}  // extern "C"
