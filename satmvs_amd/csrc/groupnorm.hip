// groupnorm.hip -- GroupNorm(1, C) (+ the gate's sigmoid / tanh) forward and backward for the TRAINING path of the
// recurrent regulariser: /root/reference/modules/module.py:15-20 (three nn.GroupNorm(1, C, 1e-5) per ConvGRU cell),
// :38-52 (sigmoid(norm(r)), sigmoid(norm(u)), tanh(norm(candidate))), differentiated by train.py:284.
//
// Why it exists: with ONE group a sample is one "row" of C*H*W values, and torch's RowwiseMomentsCUDAKernel gives a row
// to ONE workgroup: 152 us per call at the cascade's sizes, 5280 calls per training step = 45 % of the step's kernel
// time (profiles/r03_train_step.txt), with ComputeInternalGradients the same way in the backward.  Here every pass is
// spread over the whole chip:
//   forward   stats: (chunks, B) workgroups sum x and x^2 of a 4096-element chunk in float64 and write the pair to the scratch
//             apply: every workgroup folds its sample's chunk pairs (a few hundred doubles from L2), then
//                    y = act((x - mean) * rstd * gamma_c + beta_c), one (b, c) row segment per workgroup
//   backward  rows:  per (b, c, segment): S1 = sum dz, S2 = sum dz * xhat   (dz = dy * act'(y)), written to the scratch
//             dx:    every workgroup folds A = sum_c gamma_c S1, Q = sum_c gamma_c S2 of its sample, then
//                    dx = rstd * (dz * gamma_c - A/N - xhat * Q/N); workgroup 0 also writes dgamma_c = sum_b S2, dbeta_c = sum_b S1
// No atomics and no clearing of the scratch: every partial sum has one writer, the folds run in a fixed order (deterministic).
// The inference pipelines do not come here: red.hip folds the statistics into the producing convolution.
#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

constexpr int GN_THREADS = 256;
constexpr int GN_ELEMS = 4096;                     // elements of one workgroup's segment (16 per thread)

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// sum of (a, b) over the workgroup; valid in thread 0
__device__ __forceinline__ void block_sum2(double& a, double& b)
{
    __shared__ double part[2][GN_THREADS / 64];
    a = wave_sum(a); b = wave_sum(b);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { part[0][wave] = a; part[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = part[0][0]; b = part[1][0];
#pragma unroll
        for (int w = 1; w < GN_THREADS / 64; ++w) { a += part[0][w]; b += part[1][w]; }
    }
}

__device__ __forceinline__ float act_fwd(float v, int act)
{
    if (act == 1) return 1.0f / (1.0f + expf(-v));
    if (act == 2) return tanhf(v);
    return v;
}
__device__ __forceinline__ float act_bwd(float dy, float y, int act)
{
    if (act == 1) return dy * (y * (1.0f - y));
    if (act == 2) return dy * (1.0f - y * y);
    return dy;
}

// x: (B, C, HW) with batch stride xbs (elements) -- the gate halves of a (B, 2C, H, W) tensor are normalised in place
__global__ __launch_bounds__(GN_THREADS)
void gn1_stats_kernel(const float* __restrict__ x, long long xbs, long long n, double* __restrict__ partial)
{
    const int b = blockIdx.y;
    const float* xp = x + (size_t)b * xbs;
    const long long i0 = (long long)blockIdx.x * GN_ELEMS;
    const long long i1 = min(i0 + GN_ELEMS, n);
    double s = 0.0, q = 0.0;
    if ((((uintptr_t)xp) & 15) == 0 && ((i1 - i0) & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(xp + i0);
        for (int i = threadIdx.x; i < (int)((i1 - i0) >> 2); i += GN_THREADS) {
            const float4 v = x4[i];
            s += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
            q += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += GN_THREADS) {
            const double v = xp[i];
            s += v; q += v * v;
        }
    }
    block_sum2(s, q);
    if (threadIdx.x == 0) { double* o = partial + 2 * ((size_t)b * gridDim.x + blockIdx.x); o[0] = s; o[1] = q; }
}

// every thread of the workgroup gets the sample's mean and 1/sqrt(var + eps) from the nblk chunk pairs of the stats pass
__device__ __forceinline__ void mean_rstd_of(const double* partial, int b, int nblk, long long n, float eps, float& mean, float& rstd)
{
    __shared__ float mr[2];
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < nblk; i += GN_THREADS) { s += partial[2 * ((size_t)b * nblk + i)]; q += partial[2 * ((size_t)b * nblk + i) + 1]; }
    block_sum2(s, q);
    if (threadIdx.x == 0) {
        const double m = s / (double)n;
        const double var = fmax(q / (double)n - m * m, 0.0);
        mr[0] = (float)m;
        mr[1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    mean = mr[0]; rstd = mr[1];
}

// What the ConvGRU cell does with a norm's output right away, folded into the apply pass (round 6: one launch per step and cell less, each way
// a ~4 us link of a launch-bound chain):
//   blend (single form, module.py:57):  out = u * h + (1 - u) * y     u (B,C,HW) at batch stride ubs (the u half of the gate tensor), h, out (B,C,HW)
//   mul   (pair form,  module.py:43):   rh  = y * h for the FIRST half (the r gate) of every sample pair; h, rh (B,C,HW)
struct GnEpilogue {
    const float* u; long long ubs; const float* h; float* out;
    const float* mh; float* rh;
};

// grid (segments of HW, B*C)
__global__ __launch_bounds__(GN_THREADS)
void gn1_apply_kernel(const float* __restrict__ x, long long xbs, const float* __restrict__ gamma, const float* __restrict__ beta,
                      const float* __restrict__ gamma2, const float* __restrict__ beta2,
                      const double* __restrict__ partial, int nblk, float eps, int act, float* __restrict__ y, float* __restrict__ mean_rstd,
                      int C, int HW, const GnEpilogue ep)
{
    const int row = blockIdx.y, b = row / C, c = row - b * C;
    float mean, rstd;
    mean_rstd_of(partial, b, nblk, (long long)C * HW, eps, mean, rstd);
    if (c == 0 && blockIdx.x == 0 && threadIdx.x == 0) { mean_rstd[2 * b] = mean; mean_rstd[2 * b + 1] = rstd; }
    const bool second = gamma2 != nullptr && (b & 1);                   // pair form: odd samples are the second half of a (B, 2C, HW) tensor
    const float g = (second ? gamma2 : gamma)[c] * rstd, o = (second ? beta2 : beta)[c] - mean * g;            // y = x * g + o
    const float* xp = x + (size_t)b * xbs + (size_t)c * HW;
    float* yp = y + ((size_t)b * C + c) * HW;
    const int i0 = blockIdx.x * GN_ELEMS, i1 = min(i0 + GN_ELEMS, HW);
    // epilogue operands of this row (null: none)
    const float* up = ep.u ? ep.u + (size_t)b * ep.ubs + (size_t)c * HW : nullptr;
    const float* hp = ep.u ? ep.h + ((size_t)b * C + c) * HW : nullptr;
    float* bo = ep.u ? ep.out + ((size_t)b * C + c) * HW : nullptr;
    const bool mul = ep.mh != nullptr && !(b & 1);
    const float* mp = mul ? ep.mh + ((size_t)(b >> 1) * C + c) * HW : nullptr;
    float* mo = mul ? ep.rh + ((size_t)(b >> 1) * C + c) * HW : nullptr;
    const uintptr_t al = ((uintptr_t)xp) | ((uintptr_t)yp) | ((uintptr_t)up) | ((uintptr_t)hp) | ((uintptr_t)bo) | ((uintptr_t)mp) | ((uintptr_t)mo);
    if ((al & 15) == 0 && ((i1 - i0) & 3) == 0 && (i0 & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(xp + i0);
        float4* y4 = reinterpret_cast<float4*>(yp + i0);
        for (int i = threadIdx.x; i < ((i1 - i0) >> 2); i += GN_THREADS) {
            const float4 v = x4[i];
            float4 r;
            r.x = act_fwd(fmaf(v.x, g, o), act); r.y = act_fwd(fmaf(v.y, g, o), act);
            r.z = act_fwd(fmaf(v.z, g, o), act); r.w = act_fwd(fmaf(v.w, g, o), act);
            y4[i] = r;
            if (up) {
                // u * h + (1 - u) * y with the products rounded one by one, as smvs_gru_blend_fwd (and torch) forms them
                const float4 uu = reinterpret_cast<const float4*>(up + i0)[i], hh = reinterpret_cast<const float4*>(hp + i0)[i];
                float4 q;
                q.x = uu.x * hh.x + (1.0f - uu.x) * r.x; q.y = uu.y * hh.y + (1.0f - uu.y) * r.y;
                q.z = uu.z * hh.z + (1.0f - uu.z) * r.z; q.w = uu.w * hh.w + (1.0f - uu.w) * r.w;
                reinterpret_cast<float4*>(bo + i0)[i] = q;
            }
            if (mp) {
                const float4 hh = reinterpret_cast<const float4*>(mp + i0)[i];
                float4 q;
                q.x = r.x * hh.x; q.y = r.y * hh.y; q.z = r.z * hh.z; q.w = r.w * hh.w;
                reinterpret_cast<float4*>(mo + i0)[i] = q;
            }
        }
    } else {
        for (int i = i0 + threadIdx.x; i < i1; i += GN_THREADS) {
            const float r = act_fwd(fmaf(xp[i], g, o), act);
            yp[i] = r;
            if (up) bo[i] = up[i] * hp[i] + (1.0f - up[i]) * r;
            if (mp) mo[i] = r * mp[i];
        }
    }
}

// grid (segments of HW, B*C): S1 = sum dz, S2 = sum dz * xhat of one (b, c) row
__global__ __launch_bounds__(GN_THREADS)
void gn1_bwd_rows_kernel(const float* __restrict__ dy, const float* __restrict__ x, long long xbs, const float* __restrict__ y,
                         const float* __restrict__ mean_rstd, int act, double* __restrict__ rows, int C, int HW)
{
    const int row = blockIdx.y, b = row / C, c = row - b * C;
    const float mean = mean_rstd[2 * b], rstd = mean_rstd[2 * b + 1];
    const float* xp = x + (size_t)b * xbs + (size_t)c * HW;
    const float* dp = dy + ((size_t)b * C + c) * HW;
    const float* yp = y + ((size_t)b * C + c) * HW;
    const int i0 = blockIdx.x * GN_ELEMS, i1 = min(i0 + GN_ELEMS, HW);
    double s1 = 0.0, s2 = 0.0;
    if (((((uintptr_t)xp) | ((uintptr_t)dp) | ((uintptr_t)yp)) & 15) == 0 && ((i1 - i0) & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(xp + i0);
        const float4* d4 = reinterpret_cast<const float4*>(dp + i0);
        const float4* y4 = reinterpret_cast<const float4*>(yp + i0);
        for (int i = threadIdx.x; i < ((i1 - i0) >> 2); i += GN_THREADS) {
            const float4 xv = x4[i], dv = d4[i];
            float4 yv = dv;
            if (act) yv = y4[i];
            const float z0 = act_bwd(dv.x, yv.x, act), z1 = act_bwd(dv.y, yv.y, act), z2 = act_bwd(dv.z, yv.z, act), z3 = act_bwd(dv.w, yv.w, act);
            s1 += (double)((z0 + z1) + (z2 + z3));
            s2 += (double)((z0 * ((xv.x - mean) * rstd) + z1 * ((xv.y - mean) * rstd)) + (z2 * ((xv.z - mean) * rstd) + z3 * ((xv.w - mean) * rstd)));
        }
    } else
    for (int i = i0 + threadIdx.x; i < i1; i += GN_THREADS) {
        const float dz = act_bwd(dp[i], act ? yp[i] : 0.0f, act);
        s1 += (double)dz;
        s2 += (double)dz * (double)((xp[i] - mean) * rstd);
    }
    block_sum2(s1, s2);
    if (threadIdx.x == 0) { double* o = rows + 2 * ((size_t)row * gridDim.x + blockIdx.x); o[0] = s1; o[1] = s2; }
}

// grid (segments of HW, B*C)
__global__ __launch_bounds__(GN_THREADS)
void gn1_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x, long long xbs, const float* __restrict__ y,
                       const float* __restrict__ gamma, const float* __restrict__ gamma2, const float* __restrict__ mean_rstd,
                       const double* __restrict__ rows, int act, float* __restrict__ dx, long long dxbs, float* __restrict__ dgamma,
                       float* __restrict__ dbeta, float* __restrict__ dgamma2, float* __restrict__ dbeta2, int B, int C, int HW)
{
    const int row = blockIdx.y, b = row / C, c = row - b * C;
    // every workgroup folds the sample's C row sums itself (C <= a few hundred doubles from L2): no launch in between
    const bool pair = gamma2 != nullptr;                     // odd samples: second parameter set
    const float* gm = (pair && (b & 1)) ? gamma2 : gamma;
    const int nseg = gridDim.x;                              // partial pairs per (b, c) row, in the order the rows pass wrote them
    double a_ = 0.0, q_ = 0.0;
    for (int j = threadIdx.x; j < C * nseg; j += GN_THREADS) {
        const double g_ = (double)gm[j / nseg];
        a_ += g_ * rows[2 * ((size_t)b * C * nseg + j)];
        q_ += g_ * rows[2 * ((size_t)b * C * nseg + j) + 1];
    }
    __shared__ float coef[2];
    block_sum2(a_, q_);
    if (threadIdx.x == 0) { coef[0] = (float)(a_ / ((double)C * HW)); coef[1] = (float)(q_ / ((double)C * HW)); }
    __syncthreads();
    if (row == 0 && blockIdx.x == 0) {                       // parameter gradients: sums over the batch, written once
        for (int k = threadIdx.x; k < C; k += GN_THREADS) {
            double s1 = 0.0, s2 = 0.0, t1 = 0.0, t2 = 0.0;
            for (int bb = 0; bb < B; ++bb)
                for (int sg = 0; sg < nseg; ++sg) {
                    const double* r_ = rows + 2 * (((size_t)bb * C + k) * nseg + sg);
                    if (pair && (bb & 1)) { t1 += r_[0]; t2 += r_[1]; }
                    else { s1 += r_[0]; s2 += r_[1]; }
                }
            dbeta[k] = (float)s1; dgamma[k] = (float)s2;
            if (pair) { dbeta2[k] = (float)t1; dgamma2[k] = (float)t2; }
        }
    }
    const float mean = mean_rstd[2 * b], rstd = mean_rstd[2 * b + 1];
    const float a = coef[0], q = coef[1], g = gm[c];
    const float* xp = x + (size_t)b * xbs + (size_t)c * HW;
    const float* dp = dy + ((size_t)b * C + c) * HW;
    const float* yp = y + ((size_t)b * C + c) * HW;
    float* op = dx + (size_t)b * dxbs + (size_t)c * HW;
    const int i0 = blockIdx.x * GN_ELEMS, i1 = min(i0 + GN_ELEMS, HW);
    if (((((uintptr_t)xp) | ((uintptr_t)dp) | ((uintptr_t)yp) | ((uintptr_t)op)) & 15) == 0 && ((i1 - i0) & 3) == 0) {
        const float4* x4 = reinterpret_cast<const float4*>(xp + i0);
        const float4* d4 = reinterpret_cast<const float4*>(dp + i0);
        const float4* y4 = reinterpret_cast<const float4*>(yp + i0);
        float4* o4 = reinterpret_cast<float4*>(op + i0);
        for (int i = threadIdx.x; i < ((i1 - i0) >> 2); i += GN_THREADS) {
            const float4 xv = x4[i], dv = d4[i];
            float4 yv = dv, r;
            if (act) yv = y4[i];
            r.x = rstd * (act_bwd(dv.x, yv.x, act) * g - a - ((xv.x - mean) * rstd) * q);
            r.y = rstd * (act_bwd(dv.y, yv.y, act) * g - a - ((xv.y - mean) * rstd) * q);
            r.z = rstd * (act_bwd(dv.z, yv.z, act) * g - a - ((xv.z - mean) * rstd) * q);
            r.w = rstd * (act_bwd(dv.w, yv.w, act) * g - a - ((xv.w - mean) * rstd) * q);
            o4[i] = r;
        }
    } else
    for (int i = i0 + threadIdx.x; i < i1; i += GN_THREADS) {
        const float dz = act_bwd(dp[i], act ? yp[i] : 0.0f, act);
        const float xh = (xp[i] - mean) * rstd;
        op[i] = rstd * (dz * g - a - xh * q);
    }
}

}  // namespace smvs

namespace smvs {

static int gn_fwd(const float* x, long long xbs, const float* gamma, const float* beta, const float* gamma2, const float* beta2, float eps, int act,
                  float* y, float* mean_rstd, double* workspace, int B, int C, int HW, void* stream, const GnEpilogue& ep = GnEpilogue{})
{
    if (!x || !gamma || !beta || !y || !mean_rstd || !workspace) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || C < 1 || HW < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if (act < 0 || act > 2) return fail(SMVS_ERR_ARG, "act must be 0 (none), 1 (sigmoid) or 2 (tanh)");
    if (xbs < (long long)C * HW) return fail(SMVS_ERR_ARG, "batch stride smaller than one sample");
    if ((long long)B * C > 65535 || B > 65535) return fail(SMVS_ERR_ARG, "B*C exceeds the grid limit 65535");
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)C * HW;
    const int nblk = (int)((n + GN_ELEMS - 1) / GN_ELEMS);
    hipLaunchKernelGGL(gn1_stats_kernel, dim3((unsigned)nblk, B), dim3(GN_THREADS), 0, st, x, xbs, n, workspace);
    hipLaunchKernelGGL(gn1_apply_kernel, dim3((HW + GN_ELEMS - 1) / GN_ELEMS, B * C), dim3(GN_THREADS), 0, st, x, xbs, gamma, beta, gamma2, beta2,
                       workspace, nblk, eps, act, y, mean_rstd, C, HW, ep);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "groupnorm1_fwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

static int gn_bwd(const float* dy, const float* x, long long xbs, const float* y, const float* gamma, const float* gamma2, const float* mean_rstd, int act,
                  float* dx, long long dxbs, float* dgamma, float* dbeta, float* dgamma2, float* dbeta2, double* workspace, int B, int C, int HW,
                  void* stream)
{
    if (!dy || !x || !gamma || !mean_rstd || !dx || !dgamma || !dbeta || !workspace) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (gamma2 && (!dgamma2 || !dbeta2)) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (act != 0 && !y) return fail(SMVS_ERR_ARG, "the activation's backward needs the forward output y");
    if (B < 1 || C < 1 || HW < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if (act < 0 || act > 2) return fail(SMVS_ERR_ARG, "act must be 0 (none), 1 (sigmoid) or 2 (tanh)");
    if (xbs < (long long)C * HW || dxbs < (long long)C * HW) return fail(SMVS_ERR_ARG, "batch stride smaller than one sample");
    if ((long long)B * C > 65535) return fail(SMVS_ERR_ARG, "B*C exceeds the grid limit 65535");
    hipStream_t st = (hipStream_t)stream;
    double* rows = workspace;                                // (B*C, segments, 2)
    const float* yy = y ? y : dy;
    const dim3 grid((HW + GN_ELEMS - 1) / GN_ELEMS, B * C);
    hipLaunchKernelGGL(gn1_bwd_rows_kernel, grid, dim3(GN_THREADS), 0, st, dy, x, xbs, yy, mean_rstd, act, rows, C, HW);
    hipLaunchKernelGGL(gn1_bwd_dx_kernel, grid, dim3(GN_THREADS), 0, st, dy, x, xbs, yy, gamma, gamma2, mean_rstd, rows, act, dx, dxbs, dgamma, dbeta,
                       dgamma2, dbeta2, B, C, HW);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "groupnorm1_bwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

}  // namespace smvs

extern "C" SMVS_EXPORT int smvs_groupnorm1_fwd(const float* x, long long x_batch_stride, const float* gamma, const float* beta, float eps,
                                               int act, float* y, float* mean_rstd, double* workspace, int B, int C, int HW, void* stream)
{
    return smvs::gn_fwd(x, x_batch_stride, gamma, beta, nullptr, nullptr, eps, act, y, mean_rstd, workspace, B, C, HW, stream);
}

extern "C" SMVS_EXPORT int smvs_groupnorm1_bwd(const float* dy, const float* x, long long x_batch_stride, const float* y, const float* gamma,
                                               const float* mean_rstd, int act, float* dx, long long dx_batch_stride, float* dgamma,
                                               float* dbeta, double* workspace, int B, int C, int HW, void* stream)
{
    return smvs::gn_bwd(dy, x, x_batch_stride, y, gamma, nullptr, mean_rstd, act, dx, dx_batch_stride, dgamma, dbeta, nullptr, nullptr, workspace, B, C,
                        HW, stream);
}

// The gate pair of a ConvGRU cell in one call: x (B, 2C, HW) contiguous = the gate convolution's output; channels [0, C) are normalised with
// (gamma, beta), channels [C, 2C) with (gamma2, beta2), each half over its own C*HW values (two nn.GroupNorm(1, C): module.py:15-16, :37-40).
// Internally 2B samples of C channels at stride C*HW.  mean_rstd (2B, 2); workspace as smvs_groupnorm1_* with 2B samples.
extern "C" SMVS_EXPORT int smvs_groupnorm1_pair_fwd(const float* x, const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                                                    float eps, int act, float* y, float* mean_rstd, double* workspace, int B, int C, int HW,
                                                    void* stream)
{
    if (!gamma2 || !beta2) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    return smvs::gn_fwd(x, (long long)C * HW, gamma, beta, gamma2, beta2, eps, act, y, mean_rstd, workspace, 2 * B, C, HW, stream);
}

// The two norms of a ConvGRU cell with the step that follows them folded into the apply pass (struct GnEpilogue above):
//   smvs_groupnorm1_fwd_blend      = smvs_groupnorm1_fwd, then out = u * h + (1 - u) * y (module.py:57); u (B,C,HW) at batch stride
//                                    u_batch_stride (the u half of the gate tensor), h and out (B,C,HW) contiguous; y is written too
//   smvs_groupnorm1_pair_fwd_mul   = smvs_groupnorm1_pair_fwd, then rh = y[:, :C] * h (module.py:43, the second operand of the candidate
//                                    convolution, which takes (x, rh) as two tensors: no concatenation); h, rh (B,C,HW) contiguous
extern "C" SMVS_EXPORT int smvs_groupnorm1_fwd_blend(const float* x, long long x_batch_stride, const float* gamma, const float* beta, float eps,
                                                     int act, float* y, float* mean_rstd, double* workspace, const float* u,
                                                     long long u_batch_stride, const float* h, float* out, int B, int C, int HW, void* stream)
{
    if (!u || !h || !out) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    if (u_batch_stride < (long long)C * HW) return smvs::fail(SMVS_ERR_ARG, "batch stride smaller than one sample");
    smvs::GnEpilogue ep{};
    ep.u = u; ep.ubs = u_batch_stride; ep.h = h; ep.out = out;
    return smvs::gn_fwd(x, x_batch_stride, gamma, beta, nullptr, nullptr, eps, act, y, mean_rstd, workspace, B, C, HW, stream, ep);
}

extern "C" SMVS_EXPORT int smvs_groupnorm1_pair_fwd_mul(const float* x, const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                                                        float eps, int act, float* y, float* mean_rstd, double* workspace, const float* h, float* rh,
                                                        int B, int C, int HW, void* stream)
{
    if (!gamma2 || !beta2 || !h || !rh) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    smvs::GnEpilogue ep{};
    ep.mh = h; ep.rh = rh;
    return smvs::gn_fwd(x, (long long)C * HW, gamma, beta, gamma2, beta2, eps, act, y, mean_rstd, workspace, 2 * B, C, HW, stream, ep);
}

extern "C" SMVS_EXPORT int smvs_groupnorm1_pair_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* gamma2,
                                                    const float* mean_rstd, int act, float* dx, float* dgamma, float* dbeta, float* dgamma2,
                                                    float* dbeta2, double* workspace, int B, int C, int HW, void* stream)
{
    if (!gamma2) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    return smvs::gn_bwd(dy, x, (long long)C * HW, y, gamma, gamma2, mean_rstd, act, dx, (long long)C * HW, dgamma, dbeta, dgamma2, dbeta2, workspace,
                        2 * B, C, HW, stream);
}

// ---- the ConvGRU cell's element-wise steps, training path -----------------------------------------------------------
// /root/reference/modules/module.py:43-44  f = cat((x, r * h), 1)   and  :57  output = u * h + (1 - u) * y.
// As torch operators these are 2 + 4 launches forward and ~10 backward per cell and plane -- of a training step whose cost IS
// its number of launches (32 000 per step, profiles/r03_train_step.txt); here one launch each way.
namespace smvs {

constexpr int GE_THREADS = 256;

__global__ __launch_bounds__(GE_THREADS)
void gru_blend_fwd_kernel(const float* __restrict__ u, const float* __restrict__ h, const float* __restrict__ y, float* __restrict__ out, long long n4, long long n)
{
    const long long i = (long long)blockIdx.x * GE_THREADS + threadIdx.x;
    if (i < n4) {
        const float4 a = reinterpret_cast<const float4*>(u)[i], b = reinterpret_cast<const float4*>(h)[i], c = reinterpret_cast<const float4*>(y)[i];
        float4 r;
        r.x = a.x * b.x + (1.0f - a.x) * c.x; r.y = a.y * b.y + (1.0f - a.y) * c.y;
        r.z = a.z * b.z + (1.0f - a.z) * c.z; r.w = a.w * b.w + (1.0f - a.w) * c.w;
        reinterpret_cast<float4*>(out)[i] = r;
    }
    const long long t = 4 * n4 + i;                          // tail (n not a multiple of 4), first workgroup only
    if (blockIdx.x == 0 && t < n) out[t] = u[t] * h[t] + (1.0f - u[t]) * y[t];
}

__global__ __launch_bounds__(GE_THREADS)
void gru_blend_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ u, const float* __restrict__ h, const float* __restrict__ y,
                          float* __restrict__ du, float* __restrict__ dh, float* __restrict__ dc, long long n4, long long n)
{
    const long long i = (long long)blockIdx.x * GE_THREADS + threadIdx.x;
    if (i < n4) {
        const float4 g = reinterpret_cast<const float4*>(dy)[i], a = reinterpret_cast<const float4*>(u)[i];
        const float4 b = reinterpret_cast<const float4*>(h)[i], c = reinterpret_cast<const float4*>(y)[i];
        float4 r;
        r.x = g.x * (b.x - c.x); r.y = g.y * (b.y - c.y); r.z = g.z * (b.z - c.z); r.w = g.w * (b.w - c.w);
        reinterpret_cast<float4*>(du)[i] = r;
        r.x = g.x * a.x; r.y = g.y * a.y; r.z = g.z * a.z; r.w = g.w * a.w;
        reinterpret_cast<float4*>(dh)[i] = r;
        r.x = g.x * (1.0f - a.x); r.y = g.y * (1.0f - a.y); r.z = g.z * (1.0f - a.z); r.w = g.w * (1.0f - a.w);
        reinterpret_cast<float4*>(dc)[i] = r;
    }
    const long long t = 4 * n4 + i;
    if (blockIdx.x == 0 && t < n) { du[t] = dy[t] * (h[t] - y[t]); dh[t] = dy[t] * u[t]; dc[t] = dy[t] * (1.0f - u[t]); }
}

// out (B, Cx + Ch, HW) = cat(x (B, Cx, HW), r * h (B, Ch, HW)); grid (segments of a sample, B)
__global__ __launch_bounds__(GE_THREADS)
void gru_mul_cat_fwd_kernel(const float* __restrict__ x, const float* __restrict__ r, const float* __restrict__ h, float* __restrict__ out,
                            long long nx, long long nh)
{
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * GE_THREADS + threadIdx.x;
    float* o = out + (size_t)b * (nx + nh);
    if (i < nx) o[i] = x[(size_t)b * nx + i];
    else if (i < nx + nh) { const size_t k = (size_t)b * nh + (i - nx); o[i] = r[k] * h[k]; }
}

// dr = dcat[:, Cx:] * h, dh = dcat[:, Cx:] * r
__global__ __launch_bounds__(GE_THREADS)
void gru_mul_cat_bwd_kernel(const float* __restrict__ dcat, const float* __restrict__ r, const float* __restrict__ h, float* __restrict__ dr,
                            float* __restrict__ dh, long long nx, long long nh)
{
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * GE_THREADS + threadIdx.x;
    if (i < nh) {
        const size_t k = (size_t)b * nh + i;
        const float g = dcat[(size_t)b * (nx + nh) + nx + i];
        dr[k] = g * h[k]; dh[k] = g * r[k];
    }
}

// the same with dh accumulated and stored IN PLACE over dcat's second half: dcat[:, Cx:] <- dcat[:, Cx:] * r + dh_acc  (the whole-cell
// backward, modules/module.py:_ConvGRUCellFn: dcat then is [dx | dh] ready to seed the gate convolution's input gradient)
__global__ __launch_bounds__(GE_THREADS)
void gru_mul_cat_bwd_acc_kernel(float* __restrict__ dcat, const float* __restrict__ r, const float* __restrict__ h, const float* __restrict__ dh_acc,
                                float* __restrict__ dr, long long nx, long long nh)
{
    const int b = blockIdx.y;
    const long long i = (long long)blockIdx.x * GE_THREADS + threadIdx.x;
    if (i < nh) {
        const size_t k = (size_t)b * nh + i, c = (size_t)b * (nx + nh) + nx + i;
        const float g = dcat[c];
        dr[k] = g * h[k];
        dcat[c] = fmaf(g, r[k], dh_acc[k]);
    }
}

}  // namespace smvs

extern "C" SMVS_EXPORT int smvs_gru_mul_cat_bwd_acc(float* dcat, const float* r, const float* h, const float* dh_acc, float* dr, int B, int Cx, int Ch,
                                                    int HW, void* stream)
{
    using namespace smvs;
    if (!dcat || !r || !h || !dr || !dh_acc) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || Cx < 0 || Ch < 1 || HW < 1 || B > 65535) return fail(SMVS_ERR_ARG, "bad dimension");
    const long long nx = (long long)Cx * HW, nh = (long long)Ch * HW;
    hipLaunchKernelGGL(gru_mul_cat_bwd_acc_kernel, dim3((unsigned)((nh + GE_THREADS - 1) / GE_THREADS), B), dim3(GE_THREADS), 0, (hipStream_t)stream,
                       dcat, r, h, dh_acc, dr, nx, nh);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "gru_mul_cat_bwd_acc launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

extern "C" SMVS_EXPORT int smvs_gru_blend_fwd(const float* u, const float* h, const float* y, float* out, long long n, void* stream)
{
    using namespace smvs;
    if (!u || !h || !y || !out) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (n < 1) return fail(SMVS_ERR_ARG, "non-positive size");
    if ((((uintptr_t)u) | ((uintptr_t)h) | ((uintptr_t)y) | ((uintptr_t)out)) & 15) return fail(SMVS_ERR_ARG, "pointers must be 16-byte aligned");
    const long long n4 = n / 4, nb = (n4 + GE_THREADS - 1) / GE_THREADS;
    hipLaunchKernelGGL(gru_blend_fwd_kernel, dim3((unsigned)(nb > 0 ? nb : 1)), dim3(GE_THREADS), 0, (hipStream_t)stream, u, h, y, out, n4, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "gru_blend_fwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

extern "C" SMVS_EXPORT int smvs_gru_blend_bwd(const float* dy, const float* u, const float* h, const float* y, float* du, float* dh, float* dcand,
                                              long long n, void* stream)
{
    using namespace smvs;
    if (!dy || !u || !h || !y || !du || !dh || !dcand) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (n < 1) return fail(SMVS_ERR_ARG, "non-positive size");
    if ((((uintptr_t)dy) | ((uintptr_t)u) | ((uintptr_t)h) | ((uintptr_t)y) | ((uintptr_t)du) | ((uintptr_t)dh) | ((uintptr_t)dcand)) & 15)
        return fail(SMVS_ERR_ARG, "pointers must be 16-byte aligned");
    const long long n4 = n / 4, nb = (n4 + GE_THREADS - 1) / GE_THREADS;
    hipLaunchKernelGGL(gru_blend_bwd_kernel, dim3((unsigned)(nb > 0 ? nb : 1)), dim3(GE_THREADS), 0, (hipStream_t)stream, dy, u, h, y, du, dh, dcand, n4, n);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "gru_blend_bwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

extern "C" SMVS_EXPORT int smvs_gru_mul_cat_fwd(const float* x, const float* r, const float* h, float* out, int B, int Cx, int Ch, int HW, void* stream)
{
    using namespace smvs;
    if (!x || !r || !h || !out) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || Cx < 0 || Ch < 1 || HW < 1 || B > 65535) return fail(SMVS_ERR_ARG, "bad dimension");
    const long long nx = (long long)Cx * HW, nh = (long long)Ch * HW;
    hipLaunchKernelGGL(gru_mul_cat_fwd_kernel, dim3((unsigned)((nx + nh + GE_THREADS - 1) / GE_THREADS), B), dim3(GE_THREADS), 0, (hipStream_t)stream,
                       x, r, h, out, nx, nh);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "gru_mul_cat_fwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

extern "C" SMVS_EXPORT int smvs_gru_mul_cat_bwd(const float* dcat, const float* r, const float* h, float* dr, float* dh, int B, int Cx, int Ch, int HW,
                                                void* stream)
{
    using namespace smvs;
    if (!dcat || !r || !h || !dr || !dh) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || Cx < 0 || Ch < 1 || HW < 1 || B > 65535) return fail(SMVS_ERR_ARG, "bad dimension");
    const long long nx = (long long)Cx * HW, nh = (long long)Ch * HW;
    hipLaunchKernelGGL(gru_mul_cat_bwd_kernel, dim3((unsigned)((nh + GE_THREADS - 1) / GE_THREADS), B), dim3(GE_THREADS), 0, (hipStream_t)stream,
                       dcat, r, h, dr, dh, nx, nh);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "gru_mul_cat_bwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}
