// regress.hip -- height regression and the flat-array RPC projectors.
//   softmax + depth_regression + max-probability     /root/reference/networks/casred.py:58-62,
//                                                    /root/reference/modules/module.py:433-439
//   streaming (plane-at-a-time) softmax regression   /root/reference/networks/casred.py:182-184,218-236
//   RPC_Photo2Obj / RPC_Obj2Photo on flat arrays     /root/reference/modules/warping.py:255-307,218-252,
//                                                    /root/reference/tools/RPCCore.py:424-489
// All three are streaming, HBM-bound kernels: one lane per pixel / point, coalesced along x.
#include <string.h>

#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

// ---- train path: p = softmax_D(reg); depth = sum_D p*h; conf = max_D p -----------------------
// Three sweeps over the D values of a pixel (max, sum of exp, normalised accumulate) -- the same
// arithmetic as torch's softmax followed by the reference's two reductions; the 2nd/3rd sweep
// hit L2 (D*4 bytes per pixel, 256 B at D=64).
__global__ __launch_bounds__(256)
void softmax_regress_kernel(const float* __restrict__ reg, const float* __restrict__ depth, int depth_is_4d, HeightGen hg,
                            float* __restrict__ out_depth, float* __restrict__ out_conf,
                            int B, int D, int HW, int W)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const int b = (int)(i / HW);
    const int pix = (int)(i % HW);
    HeightPix hpx;
    if (depth_is_4d == HEIGHT_GENERATED) hg_prepare(hg, b, pix / W, pix % W, hpx);
    const float* r = reg + (size_t)b * D * HW + pix;
    float mx = r[0];
    for (int d = 1; d < D; ++d) mx = fmaxf(mx, r[(size_t)d * HW]);
    float den = 0.0f;
    for (int d = 0; d < D; ++d) den = den + expf(r[(size_t)d * HW] - mx);
    float acc = 0.0f, best = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float pr = __fdiv_rn(expf(r[(size_t)d * HW] - mx), den);
        const float hv = depth_is_4d == HEIGHT_GENERATED ? hg_height(hg, hpx, d)
                         : depth_is_4d ? depth[((size_t)b * D + d) * HW + pix] : depth[(size_t)b * D + d];
        acc = acc + pr * hv;
        best = (d == 0 || pr > best) ? pr : best;
    }
    out_depth[i] = acc;
    out_conf[i] = best;
}

// ---- casmvs / ucs flavour: softmax + expected height + window-4 confidence (+ ucs standard deviation) ------
// networks/casmvs.py:66-74 and networks/ucs.py:60-74 in one pass structure over (B,D,H,W):
//   p = softmax_D(reg); depth = sum p*h; idx = clamp(trunc(sum p*d), 0, D-1);
//   conf = p[idx-1] + p[idx] + p[idx+1] + p[idx+2] (planes outside [0,D) count 0: F.pad(.,(1,2)) + 4*avg_pool3d(4,1,1));
//   variance = lamb * sqrt(sum p * (h - depth)^2)                      (ucs only, out_var may be null)
// One lane per pixel; the D regulariser values of a pixel are re-read per pass (they sit in L2: a stage volume is
// 3.5 - 9.4 MB).  float32 throughout, same operation order as the torch composite the reference runs.
__global__ __launch_bounds__(256)
void window_regress_kernel(const float* __restrict__ reg, const float* __restrict__ depth, int depth_is_4d, HeightGen hg,
                           float* __restrict__ out_depth, float* __restrict__ out_conf, float* __restrict__ out_var,
                           float lamb, int B, int D, int HW, int W)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const int b = (int)(i / HW);
    const int pix = (int)(i % HW);
    const float* r = reg + (size_t)b * D * HW + pix;
    const bool gen = depth_is_4d == HEIGHT_GENERATED;
    HeightPix hpx;
    if (gen) hg_prepare(hg, b, pix / W, pix % W, hpx);
    const float* hp = gen ? r : (depth_is_4d ? depth + (size_t)b * D * HW + pix : depth + (size_t)b * D);
    const size_t hs = depth_is_4d ? (size_t)HW : 1;
    auto height = [&](int d) { return gen ? hg_height(hg, hpx, d) : hp[d * hs]; };
    float mx = r[0];
    for (int d = 1; d < D; ++d) mx = fmaxf(mx, r[(size_t)d * HW]);
    float den = 0.0f;
    for (int d = 0; d < D; ++d) den = den + expf(r[(size_t)d * HW] - mx);
    float acc = 0.0f, fidx = 0.0f;
    for (int d = 0; d < D; ++d) {
        const float pr = __fdiv_rn(expf(r[(size_t)d * HW] - mx), den);
        acc = acc + pr * height(d);
        fidx = fidx + pr * (float)d;
    }
    int idx = (int)fidx;                                   // .long(): truncation (the value is >= 0)
    idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
    float conf = 0.0f;
    for (int k = -1; k <= 2; ++k) {                        // pooling window [idx-1, idx+2], summed front to back
        const int d = idx + k;
        const float pr = (d >= 0 && d < D) ? __fdiv_rn(expf(r[(size_t)d * HW] - mx), den) : 0.0f;
        conf = conf + pr;
    }
    out_depth[i] = acc;
    out_conf[i] = conf;
    if (out_var) {
        float v = 0.0f;
        for (int d = 0; d < D; ++d) {
            const float pr = __fdiv_rn(expf(r[(size_t)d * HW] - mx), den);
            const float dh = height(d) - acc;
            v = v + (dh * dh) * pr;
        }
        out_var[i] = lamb * sqrtf(v);
    }
}

// ---- the generated hypotheses as a (B,D,H,W) tensor (training path, tests) ------------------------------------
__global__ __launch_bounds__(256)
void height_hypotheses_kernel(HeightGen hg, float* __restrict__ out, int B, int D, int HW, int W)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const int b = (int)(i / HW);
    const int pix = (int)(i % HW);
    HeightPix hpx;
    hg_prepare(hg, b, pix / W, pix % W, hpx);
    for (int d = 0; d < D; ++d) out[((size_t)b * D + d) * HW + pix] = hg_height(hg, hpx, d);
}

// ---- pred path, one plane: float64 accumulators, no max-subtraction (casred.py:218-231) -----------
__global__ __launch_bounds__(256)
void stream_regress_step_kernel(const float* __restrict__ reg_plane, const float* __restrict__ depth,
                                int depth_is_4d, HeightGen hg, double* __restrict__ exp_sum, double* __restrict__ depth_img,
                                double* __restrict__ max_prob, int B, int D, int HW, int W, int d)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * HW) return;
    const int b = (int)(i / HW);
    const int pix = (int)(i % HW);
    const double pr = exp((double)reg_plane[i]);
    double hv;
    if (depth_is_4d == HEIGHT_GENERATED) {
        HeightPix hpx;
        hg_prepare(hg, b, pix / W, pix % W, hpx);
        hv = (double)hg_height(hg, hpx, d);
    } else {
        hv = depth_is_4d ? (double)depth[((size_t)b * D + d) * HW + pix] : (double)depth[(size_t)b * D + d];
    }
    const double m = max_prob[i], di = depth_img[i], es = exp_sum[i];
    max_prob[i] = (m < pr) ? pr : m;
    depth_img[i] = fma(hv, pr, di);
    exp_sum[i] = es + pr;
}

__global__ __launch_bounds__(256)
void stream_regress_final_kernel(const double* __restrict__ exp_sum, const double* __restrict__ depth_img,
                                 const double* __restrict__ max_prob, float* __restrict__ out_depth,
                                 float* __restrict__ out_conf, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double den = exp_sum[i] + 1e-10;
    out_depth[i] = (float)(depth_img[i] / den);
    out_conf[i] = (float)(max_prob[i] / den);
}

// ---- flat projectors ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void rpc_project_kernel(const double* __restrict__ rpc, const double* __restrict__ a, const double* __restrict__ b,
                        const double* __restrict__ h, double* __restrict__ o0, double* __restrict__ o1,
                        size_t n, int dir)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const cgeo_t r = as_cgeo(rpc);
    double u, v;
    if (dir == 0) rpc_photo2obj(r, rpc_inv_image(r), a[i], b[i], h[i], u, v);
    else          rpc_obj2photo(r, rpc_inv_ground(r), a[i], b[i], h[i], u, v);
    o0[i] = u;
    o1[i] = v;
}

char* last_error_buf()
{
    static thread_local char buf[512] = {0};
    return buf;
}

static int check_launch(const char* what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "%s launch: %s", what, hipGetErrorString(e));
    return SMVS_OK;
}

}  // namespace smvs

extern "C" {

SMVS_EXPORT const char* smvs_version(void) { return "satmvs-hip 0.1.0 gfx950"; }

SMVS_EXPORT const char* smvs_last_error(void) { return smvs::last_error_buf(); }

}  // extern "C"

namespace smvs {

// heights: exactly one of (depth, gen)
int resolve_heights(const float* depth, int depth_is_4d, const smvs_height_gen* gen, int D, int H, int W,
                           int& mode, HeightGen& hg)
{
    hg = HeightGen{};
    if (gen) {
        HeightGenHost hh;
        if (const char* msg = height_gen_check(gen, D, H, W, hh)) return fail(SMVS_ERR_ARG, "%s", msg);
        hg.prev = hh.prev; hg.hp = hh.hp; hg.wp = hh.wp; hg.ih = hh.ih; hg.iw = hh.iw; hg.scale = hh.scale; hg.c = hh.c; hg.ndm1 = hh.ndm1; hg.var = hh.var; hg.rmin = hh.rmin; hg.rmax = hh.rmax;
        mode = HEIGHT_GENERATED;
        return SMVS_OK;
    }
    if (!depth) return fail(SMVS_ERR_ARG, "null pointer argument");
    mode = (depth_is_4d & ~SMVS_CALL_ARITH_MASK) ? HEIGHT_TENSOR : HEIGHT_PLANES;
    return SMVS_OK;
}

static int softmax_regress(const float* reg, const float* depth, int depth_is_4d, const smvs_height_gen* gen,
                           float* out_depth, float* out_conf, int B, int D, int H, int W, void* stream)
{
    if (!reg || !out_depth || !out_conf) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || D < 1 || H < 1 || W < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    int mode; HeightGen hg;
    if (int rc = resolve_heights(depth, depth_is_4d, gen, D, H, W, mode, hg)) return rc;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(softmax_regress_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, reg, depth, mode, hg, out_depth, out_conf, B, D, H * W, W);
    return check_launch("softmax_regress");
}

static int window_regress(const float* reg, const float* depth, int depth_is_4d, const smvs_height_gen* gen,
                          float* out_depth, float* out_conf, float* out_var, float lamb,
                          int B, int D, int H, int W, void* stream)
{
    if (!reg || !out_depth || !out_conf) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || D < 1 || H < 1 || W < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    int mode; HeightGen hg;
    if (int rc = resolve_heights(depth, depth_is_4d, gen, D, H, W, mode, hg)) return rc;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(window_regress_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, reg, depth, mode, hg, out_depth, out_conf, out_var, lamb, B, D, H * W, W);
    return check_launch("window_regress");
}

int stream_regress_step(const float* reg_plane, const float* depth, int depth_is_4d, const smvs_height_gen* gen,
                        double* exp_sum, double* depth_img, double* max_prob,
                        int B, int D, int H, int W, int d, void* stream)
{
    if (!reg_plane || !exp_sum || !depth_img || !max_prob) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || D < 1 || H < 1 || W < 1 || d < 0 || d >= D) return fail(SMVS_ERR_ARG, "bad dimension or plane index %d of %d", d, D);
    int mode; HeightGen hg;
    if (int rc = resolve_heights(depth, depth_is_4d, gen, D, H, W, mode, hg)) return rc;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(stream_regress_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, reg_plane, depth, mode, hg, exp_sum, depth_img, max_prob, B, D, H * W, W, d);
    return check_launch("stream_regress_step");
}

// The reduce step of the sharded regression's exchange (satmvs_amd/shard.py): rank r receives every rank's copy of chunk r of the
// flattened (3,B,H,W) float64 accumulators [exp_sum | depth_img | max_prob] and folds them in rank order -- sum for the first two
// rows, max for the third; element i of the chunk is element first + i of the flattened slab, its row is (first + i) / row_len --
// straight into the slab position the all-gather sends from.  One launch instead of two torch reductions and two layout copies;
// the association is fixed (rank 0, 1, 2, ...), so every rank and every run produce the same bits.
__global__ void regress_fold_kernel(const double* __restrict__ recv, double* __restrict__ out, int world, size_t chunk, size_t first, size_t row_len)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chunk) return;
    const bool is_max = (first + i) / row_len >= 2;
    double v = recv[i];
    for (int r = 1; r < world; ++r) {
        const double x = recv[(size_t)r * chunk + i];
        v = is_max ? (v < x ? x : v) : v + x;
    }
    out[i] = v;
}

}  // namespace smvs

extern "C" {

SMVS_EXPORT int smvs_height_hypotheses(const smvs_height_gen* gen, float* out, int B, int H, int W, void* stream)
{
    if (!gen || !out) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || H < 1 || W < 1) return smvs::fail(SMVS_ERR_ARG, "non-positive dimension");
    int mode; smvs::HeightGen hg;
    if (int rc = smvs::resolve_heights(nullptr, 0, gen, gen->ndepth, H, W, mode, hg)) return rc;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(smvs::height_hypotheses_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, hg, out, B, gen->ndepth, H * W, W);
    return smvs::check_launch("height_hypotheses");
}

SMVS_EXPORT int smvs_softmax_regress_fwd(const float* reg, const float* depth, int depth_is_4d,
                                         float* out_depth, float* out_conf, int B, int D, int H, int W, void* stream)
{
    return smvs::softmax_regress(reg, depth, depth_is_4d, nullptr, out_depth, out_conf, B, D, H, W, stream);
}

SMVS_EXPORT int smvs_softmax_regress_fwd_gen(const float* reg, const smvs_height_gen* gen,
                                             float* out_depth, float* out_conf, int B, int D, int H, int W, void* stream)
{
    if (!gen) return smvs::fail(SMVS_ERR_ARG, "null height generator");
    return smvs::softmax_regress(reg, nullptr, 0, gen, out_depth, out_conf, B, D, H, W, stream);
}

SMVS_EXPORT int smvs_window_regress_fwd(const float* reg, const float* depth, int depth_is_4d,
                                        float* out_depth, float* out_conf, float* out_var, float lamb,
                                        int B, int D, int H, int W, void* stream)
{
    return smvs::window_regress(reg, depth, depth_is_4d, nullptr, out_depth, out_conf, out_var, lamb, B, D, H, W, stream);
}

SMVS_EXPORT int smvs_window_regress_fwd_gen(const float* reg, const smvs_height_gen* gen,
                                            float* out_depth, float* out_conf, float* out_var, float lamb,
                                            int B, int D, int H, int W, void* stream)
{
    if (!gen) return smvs::fail(SMVS_ERR_ARG, "null height generator");
    return smvs::window_regress(reg, nullptr, 0, gen, out_depth, out_conf, out_var, lamb, B, D, H, W, stream);
}

SMVS_EXPORT int smvs_stream_regress_step(const float* reg_plane, const float* depth, int depth_is_4d,
                                         double* exp_sum, double* depth_img, double* max_prob,
                                         int B, int D, int H, int W, int d, void* stream)
{
    return smvs::stream_regress_step(reg_plane, depth, depth_is_4d, nullptr, exp_sum, depth_img, max_prob, B, D, H, W, d, stream);
}

SMVS_EXPORT int smvs_stream_regress_final(const double* exp_sum, const double* depth_img, const double* max_prob,
                                          float* out_depth, float* out_conf, size_t n, void* stream)
{
    if (!exp_sum || !depth_img || !max_prob || !out_depth || !out_conf) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    if (n == 0) return SMVS_OK;
    hipLaunchKernelGGL(smvs::stream_regress_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, exp_sum, depth_img, max_prob, out_depth, out_conf, n);
    return smvs::check_launch("stream_regress_final");
}

SMVS_EXPORT int smvs_regress_fold(const double* recv, double* out, int world, size_t chunk, size_t first, size_t row_len, void* stream)
{
    if (!recv || !out) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    if (world < 1 || row_len == 0) return smvs::fail(SMVS_ERR_ARG, "bad world size / row length");
    if (chunk == 0) return SMVS_OK;
    hipLaunchKernelGGL(smvs::regress_fold_kernel, dim3((unsigned)((chunk + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       recv, out, world, chunk, first, row_len);
    return smvs::check_launch("regress_fold");
}

SMVS_EXPORT int smvs_rpc_project(const double* rpc170, const double* a, const double* b, const double* h,
                                 double* o0, double* o1, size_t n, int dir, void* stream)
{
    if (!rpc170 || !a || !b || !h || !o0 || !o1) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    if (dir != 0 && dir != 1) return smvs::fail(SMVS_ERR_ARG, "dir must be 0 (photo->object) or 1 (object->photo)");
    if (n == 0) return SMVS_OK;
    hipLaunchKernelGGL(smvs::rpc_project_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, rpc170, a, b, h, o0, o1, n, dir);
    return smvs::check_launch("rpc_project");
}

}  // extern "C"
