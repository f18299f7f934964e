// batchnorm.hip -- BatchNorm3d in TRAINING form (batch statistics) with the block's ReLU, forward and backward: the normalisation of
// every Conv3d / Deconv3d block of CostRegNet (/root/reference/modules/module.py:324-410: conv -> nn.BatchNorm3d -> F.relu) under
// autograd (/root/reference/train.py:279-285 with --model casmvs / ucs).
//
// On this image torch hands these to MIOpenBatchNormFwdTrainSpatial / MIOpenBatchNormBwdSpatial + a threshold kernel each way:
// 11.5 ms of the 42 ms training step of the 48/32/8 cascade at the 768x384 tile for ~0.5 GB of activations per pass
// (profiles/r05_train_step_casmvs.txt) -- an HBM stream that should take ~1 ms.  Four plain streaming kernels, float4 accesses,
// per-thread float partial sums (<= 64 elements) folded in float64 over the block and added to a per-channel float64 pair with one
// atomic per block:
//   forward   1. bn_stats:  S1 = sum x, S2 = sum x^2 per channel over (B, D*H*W)
//             2. bn_apply:  y = [relu]((x - mean) * scale + beta), scale = gamma * rstd; one thread per channel
//                           stores (mean, rstd) for the backward and updates running_mean / running_var (unbiased, momentum) like
//                           torch.nn.functional.batch_norm(training=True)
//   backward  3. bn_bwd_stats:  with dy' = dy where the forward's output was positive (the SAME fma recomputed from x: no mask tensor,
//                               no saved output) -- T1 = sum dy', T2 = sum dy' * xhat
//             4. bn_bwd_dx:     dx = scale * (dy' - T1/N - xhat * T2/N);  dgamma = T2, dbeta = T1
#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

constexpr int BN_BLOCK = 256;
constexpr int BN_CHUNK = BN_BLOCK * 4 * 16;        // elements of one channel a block walks: 16 float4 per thread
constexpr int BN_UNROLL = 4;                        // ... four of them in flight at a time

__device__ __forceinline__ double bn_wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block-wide sum of two doubles; the result is valid in thread 0
__device__ __forceinline__ void bn_block_sum2(double& a, double& b)
{
    __shared__ double red[2][BN_BLOCK / 64];
    a = bn_wave_sum(a); b = bn_wave_sum(b);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = red[0][0]; b = red[1][0];
#pragma unroll
        for (int w = 1; w < BN_BLOCK / 64; ++w) { a += red[0][w]; b += red[1][w]; }
    }
}

struct BnArgs {
    const float* x; const float* dy; float* y; float* dx;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;       // updated by the forward, or null
    long long* num_batches;                         // nn.BatchNorm's num_batches_tracked, incremented by the forward, or null
    float* saved;                                   // (C, 2): mean of (x - pivot) with pivot = x[0, c, 0] (re-read by the backward), rstd
    double* sums;                                   // (C, 2) float64 scratch, zeroed by the caller's launch sequence
    float* dgamma; float* dbeta;
    int B, C; long long N;                          // N = D*H*W
    int nchunk;                                     // chunks per (batch item, channel)
    float eps, momentum; int relu, vec4;
};

// grid (nchunk * B, C): one block = one chunk of one (b, c) row
template <bool BWD>
__global__ __launch_bounds__(BN_BLOCK)
void bn_stats_kernel(const BnArgs a)
{
    const int c = blockIdx.y, b = blockIdx.x / a.nchunk, ch = blockIdx.x % a.nchunk;
    const size_t row = ((size_t)b * a.C + c) * (size_t)a.N;
    const long long i0 = (long long)ch * BN_CHUNK, i1 = min(a.N, i0 + BN_CHUNK);
    float mean = 0.0f, rstd = 0.0f, scale = 0.0f, shift = 0.0f;      // mean: of (x - pivot), see below
    if (BWD) {
        mean = a.saved[2 * c]; rstd = a.saved[2 * c + 1];
        scale = a.gamma[c] * rstd; shift = a.beta[c];      // y = (x - mean) * scale + beta: the centred form (x * scale + (beta - mean * scale) cancels for |mean| >> std)
    }
    // Forward: sums of (x - pivot) and (x - pivot)^2 with the channel's first element as the pivot (the same for every block of
    // the channel; bn_apply_kernel re-reads it).  E[x^2] - E[x]^2 on raw float32 partial sums cancels catastrophically for a
    // channel whose |mean| is much larger than its spread (relative error of the variance ~1e-6 (mean / std)^2: x = 100 + 0.1 n
    // loses everything); shifted by a sample of the channel the two terms are of the size of the variance itself.
    // The pivot stays part of the centring everywhere: x - mean is formed as (x - pivot) - mean(x - pivot), whose two terms are exact /
    // small, instead of against a float32 mean whose own rounding (1.5e-5 at 150) times gamma * rstd would show in y.
    const float pivot = a.x[(size_t)c * (size_t)a.N];
    float s1 = 0.0f, s2 = 0.0f;
    auto one = [&](float xv, float gv) {
        if (BWD) {
            const float xc = (xv - pivot) - mean;
            const float g = (!a.relu || fmaf(xc, scale, shift) > 0.0f) ? gv : 0.0f;
            s1 += g; s2 = fmaf(g, xc * rstd, s2);
        } else {
            const float dv = xv - pivot;
            s1 += dv; s2 = fmaf(dv, dv, s2);
        }
    };
    if (a.vec4) {
        // four independent 16-byte loads per stream in flight per thread (one at a time left the full-volume layers at ~1.5 TB/s)
        for (long long i = i0 + threadIdx.x * 4; i < i1; i += BN_BLOCK * 4 * BN_UNROLL) {
            float4 xv[BN_UNROLL], gv[BN_UNROLL];
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
                const long long iu = i + (long long)u * BN_BLOCK * 4;
                const bool in = iu < i1;
                xv[u] = in ? *reinterpret_cast<const float4*>(a.x + row + iu) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                gv[u] = (BWD && in) ? *reinterpret_cast<const float4*>(a.dy + row + iu) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u)
                if (i + (long long)u * BN_BLOCK * 4 < i1) { one(xv[u].x, gv[u].x); one(xv[u].y, gv[u].y); one(xv[u].z, gv[u].z); one(xv[u].w, gv[u].w); }
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += BN_BLOCK) one(a.x[row + i], BWD ? a.dy[row + i] : 0.0f);
    }
    double d1 = (double)s1, d2 = (double)s2;
    bn_block_sum2(d1, d2);
    if (threadIdx.x == 0) {
        atomicAdd(a.sums + 2 * c, d1);
        atomicAdd(a.sums + 2 * c + 1, d2);
    }
}

// forward apply; block (0, c) also publishes (mean, rstd) and the running statistics
__global__ __launch_bounds__(BN_BLOCK)
void bn_apply_kernel(const BnArgs a)
{
    const int c = blockIdx.y, b = blockIdx.x / a.nchunk, ch = blockIdx.x % a.nchunk;
    const size_t row = ((size_t)b * a.C + c) * (size_t)a.N;
    const long long i0 = (long long)ch * BN_CHUNK, i1 = min(a.N, i0 + BN_CHUNK);
    const double cnt = (double)a.B * (double)a.N;
    const double ms = a.sums[2 * c] / cnt;                        // mean of (x - pivot), see bn_stats_kernel
    const double m = (double)a.x[(size_t)c * (size_t)a.N] + ms;
    double var = a.sums[2 * c + 1] / cnt - ms * ms;
    var = var > 0.0 ? var : 0.0;
    const float pivot = a.x[(size_t)c * (size_t)a.N];
    const float mean = (float)m, mshift = (float)ms, rstd = (float)(1.0 / sqrt(var + (double)a.eps));
    const float scale = a.gamma[c] * rstd, shift = a.beta[c];      // y = (x - mean) * scale + beta, centred
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (c == 0 && a.num_batches) *a.num_batches += 1;
        a.saved[2 * c] = mshift; a.saved[2 * c + 1] = rstd;            // (mean of x - pivot: the backward kernels centre the same way)
        if (a.running_mean) {
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            a.running_mean[c] = (1.0f - a.momentum) * a.running_mean[c] + a.momentum * mean;
            a.running_var[c] = (1.0f - a.momentum) * a.running_var[c] + a.momentum * (float)unbiased;
        }
    }
    auto one = [&](float xv) { const float r = fmaf((xv - pivot) - mshift, scale, shift); return a.relu ? fmaxf(r, 0.0f) : r; };
    if (a.vec4) {
        for (long long i = i0 + threadIdx.x * 4; i < i1; i += BN_BLOCK * 4 * BN_UNROLL) {
            float4 xv[BN_UNROLL];
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
                const long long iu = i + (long long)u * BN_BLOCK * 4;
                xv[u] = iu < i1 ? *reinterpret_cast<const float4*>(a.x + row + iu) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
                const long long iu = i + (long long)u * BN_BLOCK * 4;
                if (iu < i1) *reinterpret_cast<float4*>(a.y + row + iu) = make_float4(one(xv[u].x), one(xv[u].y), one(xv[u].z), one(xv[u].w));
            }
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += BN_BLOCK) a.y[row + i] = one(a.x[row + i]);
    }
}

__global__ __launch_bounds__(BN_BLOCK)
void bn_bwd_dx_kernel(const BnArgs a)
{
    const int c = blockIdx.y, b = blockIdx.x / a.nchunk, ch = blockIdx.x % a.nchunk;
    const size_t row = ((size_t)b * a.C + c) * (size_t)a.N;
    const long long i0 = (long long)ch * BN_CHUNK, i1 = min(a.N, i0 + BN_CHUNK);
    const double cnt = (double)a.B * (double)a.N;
    const float mean = a.saved[2 * c], rstd = a.saved[2 * c + 1];
    const float scale = a.gamma[c] * rstd, shift = a.beta[c];      // y = (x - mean) * scale + beta, centred
    const float t1 = (float)(a.sums[2 * c] / cnt), t2 = (float)(a.sums[2 * c + 1] / cnt);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.dbeta[c] = (float)a.sums[2 * c];
        a.dgamma[c] = (float)a.sums[2 * c + 1];
    }
    const float pivot = a.x[(size_t)c * (size_t)a.N];
    auto one = [&](float xv, float gv) {
        const float xc = (xv - pivot) - mean;                          // mean: of (x - pivot), as saved by the forward
        const float g = (!a.relu || fmaf(xc, scale, shift) > 0.0f) ? gv : 0.0f;
        const float xh = xc * rstd;
        return scale * (g - t1 - xh * t2);
    };
    if (a.vec4) {
        for (long long i = i0 + threadIdx.x * 4; i < i1; i += BN_BLOCK * 4 * BN_UNROLL) {
            float4 xv[BN_UNROLL], gv[BN_UNROLL];
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
                const long long iu = i + (long long)u * BN_BLOCK * 4;
                const bool in = iu < i1;
                xv[u] = in ? *reinterpret_cast<const float4*>(a.x + row + iu) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                gv[u] = in ? *reinterpret_cast<const float4*>(a.dy + row + iu) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
#pragma unroll
            for (int u = 0; u < BN_UNROLL; ++u) {
                const long long iu = i + (long long)u * BN_BLOCK * 4;
                if (iu < i1) *reinterpret_cast<float4*>(a.dx + row + iu) = make_float4(one(xv[u].x, gv[u].x), one(xv[u].y, gv[u].y), one(xv[u].z, gv[u].z), one(xv[u].w, gv[u].w));
            }
        }
    } else {
        for (long long i = i0 + threadIdx.x; i < i1; i += BN_BLOCK) a.dx[row + i] = one(a.x[row + i], a.dy[row + i]);
    }
}

static int bn_geometry(BnArgs& a, int B, int C, long long N)
{
    if (B < 1 || C < 1 || N < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if (C > 65535) return fail(SMVS_ERR_ARG, "more than 65535 channels");
    a.B = B; a.C = C; a.N = N;
    const long long nchunk = (N + BN_CHUNK - 1) / BN_CHUNK;
    if (nchunk * B >= (1ll << 31)) return fail(SMVS_ERR_ARG, "tensor too large for one launch grid");
    a.nchunk = (int)nchunk;
    return SMVS_OK;
}

static bool bn_aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace smvs

extern "C" SMVS_EXPORT int smvs_batchnorm_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean,
                                                    float* running_var, long long* num_batches_tracked, float momentum, float eps, int relu,
                                                    float* y, float* saved_mean_rstd, double* workspace, int B, int C, long long N, void* stream)
{
    using namespace smvs;
    if (!x || !gamma || !beta || !y || !saved_mean_rstd || !workspace) return fail(SMVS_ERR_ARG, "null pointer argument");
    if ((running_mean == nullptr) != (running_var == nullptr)) return fail(SMVS_ERR_ARG, "running_mean and running_var go together");
    BnArgs a{};
    const int rc = bn_geometry(a, B, C, N);
    if (rc) return rc;
    a.x = x; a.y = y; a.gamma = gamma; a.beta = beta; a.running_mean = running_mean; a.running_var = running_var; a.num_batches = num_batches_tracked;
    a.saved = saved_mean_rstd; a.sums = workspace; a.eps = eps; a.momentum = momentum; a.relu = (relu & 1) != 0;
    a.vec4 = (N % 4 == 0) && bn_aligned16(x) && bn_aligned16(y);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = (relu & SMVS_BN_WORKSPACE_ZERO) ? hipSuccess : hipMemsetAsync(workspace, 0, sizeof(double) * 2 * C, st);
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "batchnorm workspace clear: %s", hipGetErrorString(e));
    const dim3 grid((unsigned)(a.nchunk * B), (unsigned)C);
    hipLaunchKernelGGL(bn_stats_kernel<false>, grid, dim3(BN_BLOCK), 0, st, a);
    hipLaunchKernelGGL(bn_apply_kernel, grid, dim3(BN_BLOCK), 0, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "batchnorm_train_fwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

extern "C" SMVS_EXPORT int smvs_batchnorm_train_bwd(const float* dy, const float* x, const float* gamma, const float* beta,
                                                    const float* saved_mean_rstd, int relu, float* dx, float* dgamma, float* dbeta,
                                                    double* workspace, int B, int C, long long N, void* stream)
{
    using namespace smvs;
    if (!dy || !x || !gamma || !beta || !saved_mean_rstd || !dx || !dgamma || !dbeta || !workspace) return fail(SMVS_ERR_ARG, "null pointer argument");
    BnArgs a{};
    const int rc = bn_geometry(a, B, C, N);
    if (rc) return rc;
    a.x = x; a.dy = dy; a.dx = dx; a.gamma = gamma; a.beta = beta; a.saved = const_cast<float*>(saved_mean_rstd); a.sums = workspace;
    a.dgamma = dgamma; a.dbeta = dbeta; a.relu = (relu & 1) != 0;
    a.vec4 = (N % 4 == 0) && bn_aligned16(x) && bn_aligned16(dy) && bn_aligned16(dx);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = (relu & SMVS_BN_WORKSPACE_ZERO) ? hipSuccess : hipMemsetAsync(workspace, 0, sizeof(double) * 2 * C, st);
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "batchnorm workspace clear: %s", hipGetErrorString(e));
    const dim3 grid((unsigned)(a.nchunk * B), (unsigned)C);
    hipLaunchKernelGGL(bn_stats_kernel<true>, grid, dim3(BN_BLOCK), 0, st, a);
    hipLaunchKernelGGL(bn_bwd_dx_kernel, grid, dim3(BN_BLOCK), 0, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "batchnorm_train_bwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}
