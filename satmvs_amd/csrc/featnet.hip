// featnet.hip -- native inference forward of the feature extractor.
//
// Replaces FeatureNet.forward (/root/reference/modules/module.py:442-543, 3 stages; arch_mode "unet" = arch 0,
// arch_mode "fpn" = arch 1: 1x1 lateral convolutions added to the nearest-upsampled coarser level) in
// eval mode: 3x3 / 5x5 / 1x1 convolutions with BatchNorm folded to a per-channel scale/shift + ReLU
// (Conv2d :19-60), two stride-2 transposed convolutions fused with their skip concatenation
// (DeConv2dFuse :117-140) and the three 1x1 output heads.  All views of a sample go through ONE call
// (the network is applied per view with shared weights, networks/casred.py:116-121): 15 launches for
// V views instead of ~45 MIOpen launches per view.
//
// The channel counts (8/16/32) are far below an MFMA tile and the planes are large, so these are direct
// float32 convolutions in the throughput regime: one lane per output pixel, 8 output channels per lane in
// registers, wave-uniform weights read with scalar loads from a pre-packed buffer ([cout/8][cin][tap][8]),
// inputs through the buffer range check (zero padding for free), the taps of channel c+1 in flight while
// channel c is multiplied, concatenated inputs read from two tensors without materialising the concat.
#include <stdlib.h>

#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

constexpr int FN_COT = 8;
constexpr int FN_NL = 15;                   // layers of the unet variant (the fpn variant has 13)

struct FnLayer { int cin, cout, k, stride, transposed, bn, relu, bias; };

// layer list in execution order; c = base channels.  Returns the number of layers.
static int fn_layers(int c, int arch, FnLayer* L)
{
    const FnLayer trunk[8] = {
        {3, c, 3, 1, 0, 1, 1, 0},          {c, c, 3, 1, 0, 1, 1, 0},                                         // conv0.0, conv0.1
        {c, 2 * c, 5, 2, 0, 1, 1, 0},      {2 * c, 2 * c, 3, 1, 0, 1, 1, 0}, {2 * c, 2 * c, 3, 1, 0, 1, 1, 0},  // conv1.0..2
        {2 * c, 4 * c, 5, 2, 0, 1, 1, 0},  {4 * c, 4 * c, 3, 1, 0, 1, 1, 0}, {4 * c, 4 * c, 3, 1, 0, 1, 1, 0}};  // conv2.0..2
    for (int i = 0; i < 8; ++i) L[i] = trunk[i];
    if (arch == 0) {
        const FnLayer head[7] = {
            {4 * c, 4 * c, 1, 1, 0, 0, 0, 0},                                                                   // out1
            {4 * c, 2 * c, 3, 2, 1, 1, 1, 0},  {4 * c, 2 * c, 3, 1, 0, 1, 1, 0},                                // deconv1.deconv, deconv1.conv
            {2 * c, 2 * c, 1, 1, 0, 0, 0, 0},                                                                   // out2
            {2 * c, c, 3, 2, 1, 1, 1, 0},      {2 * c, c, 3, 1, 0, 1, 1, 0},                                    // deconv2.deconv, deconv2.conv
            {c, c, 1, 1, 0, 0, 0, 0}};                                                                          // out3
        for (int i = 0; i < 7; ++i) L[8 + i] = head[i];
        return 15;
    }
    const FnLayer head[5] = {
        {4 * c, 4 * c, 1, 1, 0, 0, 0, 0},                                                                       // out1
        {2 * c, 4 * c, 1, 1, 0, 0, 0, 1},  {4 * c, 2 * c, 3, 1, 0, 0, 0, 0},                                    // inner1 (+ up(c2)), out2
        {c, 4 * c, 1, 1, 0, 0, 0, 1},      {4 * c, c, 3, 1, 0, 0, 0, 0}};                                       // inner2 (+ up(f1)), out3
    for (int i = 0; i < 5; ++i) L[8 + i] = head[i];
    return 13;
}

__host__ __device__ inline size_t fn_packed_conv(int cin, int cout, int k) { return (size_t)((cout + FN_COT - 1) / FN_COT) * cin * k * k * FN_COT; }

struct FnLayout { size_t w[FN_NL], scale[FN_NL], shift[FN_NL], total; };

static FnLayout fn_layout(int c, int arch)
{
    FnLayer L[FN_NL];
    const int nl = fn_layers(c, arch, L);
    FnLayout lay{};
    size_t o = 0;
    for (int i = 0; i < nl; ++i) {
        const size_t cp = (size_t)((L[i].cout + FN_COT - 1) / FN_COT) * FN_COT;
        lay.w[i] = o; o += fn_packed_conv(L[i].cin, L[i].cout, L[i].k);
        lay.scale[i] = o; o += cp;
        lay.shift[i] = o; o += cp;
    }
    lay.total = o;
    return lay;
}

// src: Conv2d weight (Cout,Cin,K,K) or ConvTranspose2d weight (Cin,Cout,3,3) -> dst [cog][cin][tap][8]
__global__ void fn_pack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int cin, int cout, int k, int transposed)
{
    const int kk = k * k;
    const int n = ((cout + FN_COT - 1) / FN_COT) * cin * kk * FN_COT;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int j = i % FN_COT, tap = (i / FN_COT) % kk, ci = (i / (FN_COT * kk)) % cin, cog = i / (FN_COT * kk * cin);
    const int co = cog * FN_COT + j;
    float v = 0.0f;
    if (co < cout) v = transposed ? src[((size_t)ci * cout + co) * kk + tap] : src[((size_t)co * cin + ci) * kk + tap];
    dst[i] = v;
}

// BatchNorm2d in eval mode: y = (x - mean) / sqrt(var + eps) * gamma + beta = x * scale + shift
__global__ void fn_pack_bn_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                  const float* __restrict__ var, float* __restrict__ scale, float* __restrict__ shift,
                                  int cout, int cp, int bn)
{
    // bn == 1: fold the BatchNorm; bn == 2: `gamma` is a plain convolution bias (scale 1); bn == 0: identity
    const int i = threadIdx.x;
    if (i >= cp) return;
    float s = 1.0f, t = 0.0f;
    if (bn == 2 && i < cout) t = gamma[i];
    if (bn == 1 && i < cout) {
        const double sd = (double)gamma[i] / sqrt((double)var[i] + 1e-5);
        s = (float)sd;
        t = (float)((double)beta[i] - (double)mean[i] * sd);
    }
    scale[i] = s;
    shift[i] = t;
}

struct FnConvArgs {
    const float* inA; int CA;                // first CA input channels
    const float* inB; int CB;                // next CB channels (skip connection), or null/0
    const float* w;                          // packed [cog][CA+CB][K*K][8]
    const float* scale; const float* shift;  // (>= Cout, padded to 8)
    float* out;                              // (N,Cout,Ho,Wo)
    const float* up;                         // (N,Cout,Ho/2,Wo/2) added after nearest x2 upsampling, or null (fpn lateral)
    int Cout, Hi, Wi, Ho, Wo, relu;
};

typedef const float __attribute__((address_space(4))) * fcw_t;

// K x K correlation, padding K/2, STRIDE 1 or 2.  Workgroup = 64 x 4 output pixels, lane = one pixel.
template <int K, int STRIDE>
__global__ __launch_bounds__(256)
void fn_conv_kernel(const FnConvArgs a)
{
    constexpr int KK = K * K, PAD = K / 2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ox = blockIdx.x * 64 + lane, oy = blockIdx.y * 4 + wave;
    const int ncog = (a.Cout + FN_COT - 1) / FN_COT;
    const int cog = blockIdx.z % ncog, n = blockIdx.z / ncog;
    const bool active = ox < a.Wo && oy < a.Ho;
    const int Cin = a.CA + a.CB;
    const int HWi = a.Hi * a.Wi;

    uint32_t off[KK];
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int iy = oy * STRIDE - PAD + ky, ix = ox * STRIDE - PAD + kx;
            off[ky * K + kx] = (active && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) ? (uint32_t)(iy * a.Wi + ix) * 4u : SMVS_OOB;
        }
    const BufRsrc rA = make_rsrc(a.inA + (size_t)n * a.CA * HWi, (uint32_t)a.CA * (uint32_t)HWi * 4u);
    const BufRsrc rB = make_rsrc(a.CB ? a.inB + (size_t)n * a.CB * HWi : a.inA, (uint32_t)a.CB * (uint32_t)HWi * 4u);

    float acc[FN_COT];
#pragma unroll
    for (int j = 0; j < FN_COT; ++j) acc[j] = 0.0f;
    const fcw_t wbase = (fcw_t)(uintptr_t)(a.w + (size_t)cog * Cin * KK * FN_COT);

#define SMVS_FN_LOAD(V, CC)                                                                            \
    {                                                                                                  \
        const bool fa_ = (CC) < a.CA;                              /* wave-uniform: scalar selects */ \
        i32x4 rx_;                                                                                     \
        rx_.x = fa_ ? rA.v.x : rB.v.x; rx_.y = fa_ ? rA.v.y : rB.v.y;                                  \
        rx_.z = fa_ ? rA.v.z : rB.v.z; rx_.w = rA.v.w;                                                 \
        const int co_ = (fa_ ? (CC) : (CC) - a.CA) * HWi * 4;                                          \
        _Pragma("unroll") for (int k_ = 0; k_ < KK; ++k_) V[k_] = llvm_raw_buffer_load_f32(rx_, (int)off[k_], co_, 0); \
    }
#define SMVS_FN_FMA(V, CC)                                                                             \
    {                                                                                                  \
        const fcw_t wc_ = wbase + (size_t)(CC) * KK * FN_COT;                                          \
        _Pragma("unroll") for (int k_ = 0; k_ < KK; ++k_)                                              \
            _Pragma("unroll") for (int j_ = 0; j_ < FN_COT; ++j_) acc[j_] = fmaf(V[k_], wc_[k_ * FN_COT + j_], acc[j_]); \
    }
    float v0[KK], v1[KK];
    // prefetch loads are unconditional (past the end: the last channel again): a load under a branch makes the compiler
    // wait with vmcnt(0) before every multiply block, i.e. also for the loads it has just issued (mfma_conv.h)
    SMVS_FN_LOAD(v0, 0)
    for (int cc = 0; cc < Cin; cc += 2) {
        SMVS_FN_LOAD(v1, min(cc + 1, Cin - 1))
        __builtin_amdgcn_sched_barrier(0);
        SMVS_FN_FMA(v0, cc)
        __builtin_amdgcn_sched_barrier(0);
        SMVS_FN_LOAD(v0, min(cc + 2, Cin - 1))
        __builtin_amdgcn_sched_barrier(0);
        if (cc + 1 < Cin) SMVS_FN_FMA(v1, cc + 1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef SMVS_FN_LOAD
#undef SMVS_FN_FMA

    if (!active) return;
    const int HWo = a.Ho * a.Wo;
    float upv[FN_COT];                           // lateral-add operands first, all in flight together (channels beyond Cout read channel 0)
#pragma unroll
    for (int j = 0; j < FN_COT; ++j) {
        const int co = cog * FN_COT + j;
        upv[j] = a.up ? a.up[((size_t)n * a.Cout + (co < a.Cout ? co : 0)) * (HWo / 4) + (size_t)(oy >> 1) * (a.Wo >> 1) + (ox >> 1)] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < FN_COT; ++j) {
        const int co = cog * FN_COT + j;
        if (co < a.Cout) {
            float r = fmaf(acc[j], a.scale[co], a.shift[co]);
            if (a.relu) r = fmaxf(r, 0.0f);
            if (a.up) r = upv[j] + r;
            a.out[((size_t)n * a.Cout + co) * HWo + (size_t)oy * a.Wo + ox] = r;
        }
    }
}

// ConvTranspose2d(k=3, stride=2, pad=1, output_padding=1) + scale/shift + ReLU: lane = one INPUT position
// (y,x) producing the 2x2 output quad (2y..2y+1, 2x..2x+1) from inputs (y,x),(y,x+1),(y+1,x),(y+1,x+1):
//   out(2y  ,2x  ) = in(y,x) w[1][1]
//   out(2y  ,2x+1) = in(y,x) w[1][2] + in(y,x+1) w[1][0]
//   out(2y+1,2x  ) = in(y,x) w[2][1] + in(y+1,x) w[0][1]
//   out(2y+1,2x+1) = in(y,x) w[2][2] + in(y,x+1) w[2][0] + in(y+1,x) w[0][2] + in(y+1,x+1) w[0][0]
__global__ __launch_bounds__(256)
void fn_convT_kernel(const FnConvArgs a)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * 4 + wave;
    const int ncog = (a.Cout + FN_COT - 1) / FN_COT;
    const int cog = blockIdx.z % ncog, n = blockIdx.z / ncog;
    const bool active = x < a.Wi && y < a.Hi;
    const int HWi = a.Hi * a.Wi;
    uint32_t off[4];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
            off[dy * 2 + dx] = (active && y + dy < a.Hi && x + dx < a.Wi) ? (uint32_t)((y + dy) * a.Wi + x + dx) * 4u : SMVS_OOB;
    const BufRsrc rA = make_rsrc(a.inA + (size_t)n * a.CA * HWi, (uint32_t)a.CA * (uint32_t)HWi * 4u);
    float acc[4][FN_COT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < FN_COT; ++j) acc[q][j] = 0.0f;
    const fcw_t wbase = (fcw_t)(uintptr_t)(a.w + (size_t)cog * a.CA * 9 * FN_COT);
#define SMVS_FT_LOAD(V, CC) \
    { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) V[k_] = llvm_raw_buffer_load_f32(rA.v, (int)off[k_], (CC) * HWi * 4, 0); }
#define SMVS_FT_FMA(V, CC)                                                                             \
    {                                                                                                  \
        const fcw_t wc = wbase + (size_t)(CC) * 9 * FN_COT;                                            \
        _Pragma("unroll") for (int j = 0; j < FN_COT; ++j) {                                           \
            acc[0][j] = fmaf(V[0], wc[4 * FN_COT + j], acc[0][j]);                                     \
            acc[1][j] = fmaf(V[0], wc[5 * FN_COT + j], fmaf(V[1], wc[3 * FN_COT + j], acc[1][j]));     \
            acc[2][j] = fmaf(V[0], wc[7 * FN_COT + j], fmaf(V[2], wc[1 * FN_COT + j], acc[2][j]));     \
            acc[3][j] = fmaf(V[0], wc[8 * FN_COT + j], fmaf(V[1], wc[6 * FN_COT + j],                  \
                        fmaf(V[2], wc[2 * FN_COT + j], fmaf(V[3], wc[0 * FN_COT + j], acc[3][j]))));   \
        }                                                                                              \
    }
    float v0[4], v1[4];
    SMVS_FT_LOAD(v0, 0)
    for (int cc = 0; cc < a.CA; cc += 2) {
        SMVS_FT_LOAD(v1, min(cc + 1, a.CA - 1))                  // unconditional: see fn_conv_kernel
        __builtin_amdgcn_sched_barrier(0);
        SMVS_FT_FMA(v0, cc)
        __builtin_amdgcn_sched_barrier(0);
        SMVS_FT_LOAD(v0, min(cc + 2, a.CA - 1))
        __builtin_amdgcn_sched_barrier(0);
        if (cc + 1 < a.CA) SMVS_FT_FMA(v1, cc + 1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef SMVS_FT_LOAD
#undef SMVS_FT_FMA
    if (!active) return;
    const int HWo = a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < FN_COT; ++j) {
        const int co = cog * FN_COT + j;
        if (co < a.Cout) {
            float* o = a.out + ((size_t)n * a.Cout + co) * HWo + (size_t)(2 * y) * a.Wo + 2 * x;
            const float sc = a.scale[co], sh = a.shift[co];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float r = fmaf(acc[q][j], sc, sh);
                if (a.relu) r = fmaxf(r, 0.0f);
                o[(q >> 1) * a.Wo + (q & 1)] = r;
            }
        }
    }
}

struct FnWorkspace { size_t t0, c0, t1a, t1b, c1, t2a, t2b, c2, up1, f1, up2, f2, total; };

static FnWorkspace fn_workspace(int N, int H, int W, int c, int arch)
{
    FnWorkspace w{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
    const size_t p0 = (size_t)N * H * W, p1 = p0 / 4, p2 = p0 / 16;
    w.t0 = take(c * p0); w.c0 = take(c * p0);
    w.t1a = take(2 * c * p1); w.t1b = take(2 * c * p1); w.c1 = take(2 * c * p1);
    w.t2a = take(4 * c * p2); w.t2b = take(4 * c * p2); w.c2 = take(4 * c * p2);
    if (arch == 0) {
        w.up1 = take(2 * c * p1); w.f1 = take(2 * c * p1);
        w.up2 = take(c * p0); w.f2 = take(c * p0);
    } else {                                               // fpn: the merged levels carry 4c channels
        w.f1 = take(4 * c * p1);
        w.f2 = take(4 * c * p0);
    }
    w.total = o;
    return w;
}

static void fn_launch(const FnLayer& l, const FnConvArgs& a, int N, hipStream_t st)
{
    const int ncog = (a.Cout + FN_COT - 1) / FN_COT;
    if (l.transposed) {
        hipLaunchKernelGGL(fn_convT_kernel, dim3((a.Wi + 63) / 64, (a.Hi + 3) / 4, N * ncog), dim3(256), 0, st, a);
        return;
    }
    const dim3 grd((a.Wo + 63) / 64, (a.Ho + 3) / 4, N * ncog), blk(256);
    if (l.k == 1)                       hipLaunchKernelGGL((fn_conv_kernel<1, 1>), grd, blk, 0, st, a);
    else if (l.k == 3)                  hipLaunchKernelGGL((fn_conv_kernel<3, 1>), grd, blk, 0, st, a);
    else                                hipLaunchKernelGGL((fn_conv_kernel<5, 2>), grd, blk, 0, st, a);
}

}  // namespace smvs

extern "C" {

SMVS_EXPORT size_t smvs_featnet_packed_floats(int base_channels, int arch)
{
    return base_channels > 0 && (arch == 0 || arch == 1) ? smvs::fn_layout(base_channels, arch).total : 0;
}

SMVS_EXPORT size_t smvs_featnet_workspace_bytes(int N, int H, int W, int base_channels, int arch)
{
    if (N < 1 || base_channels < 1 || H < 4 || W < 4 || (H % 4) || (W % 4) || (arch != 0 && arch != 1)) return 0;
    if ((long long)4 * base_channels * H * W * 4 >= (1ll << 31)) return 0;                       // unsupported sizes: 0
    if ((long long)N * ((4 * base_channels + smvs::FN_COT - 1) / smvs::FN_COT) > 65535) return 0;
    return smvs::fn_workspace(N, H, W, base_channels, arch).total * sizeof(float);
}

// params: HOST array of device pointers in the module's own order.  Both variants start with
//   for each of conv0.0, conv0.1, conv1.0, conv1.1, conv1.2, conv2.0, conv2.1, conv2.2:
//       conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var                       (8 x 5 = 40)
// arch 0 (unet, 63 pointers): the same 5 for deconv1.deconv, deconv1.conv, deconv2.deconv, deconv2.conv (+20),
//   then out1.weight, out2.weight, out3.weight.
// arch 1 (fpn, 47 pointers): out1.weight, inner1.weight, inner1.bias, out2.weight, inner2.weight, inner2.bias,
//   out3.weight.
SMVS_EXPORT int smvs_featnet_pack_weights(const float* const* params, int base_channels, int arch, float* packed, void* stream)
{
    using namespace smvs;
    if (!params || !packed) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (base_channels < 1) return fail(SMVS_ERR_ARG, "non-positive channel count");
    if (arch != 0 && arch != 1) return fail(SMVS_ERR_ARG, "arch must be 0 (unet) or 1 (fpn)");
    const int nparams = arch == 0 ? 63 : 47;
    for (int i = 0; i < nparams; ++i)
        if (!params[i]) return fail(SMVS_ERR_ARG, "null parameter pointer %d", i);
    FnLayer L[FN_NL];
    const int nl = fn_layers(base_channels, arch, L);
    const FnLayout lay = fn_layout(base_channels, arch);
    // execution-order layer -> index of its first pointer
    const int first_unet[15] = {0, 5, 10, 15, 20, 25, 30, 35, 60, 40, 45, 61, 50, 55, 62};
    const int first_fpn[13] = {0, 5, 10, 15, 20, 25, 30, 35, 40, 41, 43, 44, 46};
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < nl; ++i) {
        const float* const* q = params + (arch == 0 ? first_unet[i] : first_fpn[i]);
        if (L[i].cout > 64) return fail(SMVS_ERR_UNSUPPORTED, "base_channels %d too large for the BatchNorm fold kernel", base_channels);
        const int n = (int)fn_packed_conv(L[i].cin, L[i].cout, L[i].k);
        hipLaunchKernelGGL(fn_pack_conv_kernel, dim3((n + 255) / 256), dim3(256), 0, st, q[0], packed + lay.w[i],
                           L[i].cin, L[i].cout, L[i].k, L[i].transposed);
        const int cp = ((L[i].cout + FN_COT - 1) / FN_COT) * FN_COT;
        const int mode = L[i].bn ? 1 : L[i].bias ? 2 : 0;
        hipLaunchKernelGGL(fn_pack_bn_kernel, dim3(1), dim3(64), 0, st, mode ? q[1] : q[0], mode == 1 ? q[2] : q[0],
                           mode == 1 ? q[3] : q[0], mode == 1 ? q[4] : q[0], packed + lay.scale[i], packed + lay.shift[i],
                           L[i].cout, cp, mode);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "featnet_pack_weights launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

// imgs (N,3,H,W) -> stage1 (N,4c,H/4,W/4), stage2 (N,2c,H/2,W/2), stage3 (N,c,H,W).  N = samples x views;
// H and W multiples of 4.
SMVS_EXPORT int smvs_featnet_fwd(const float* packed, const float* imgs, float* stage1, float* stage2, float* stage3,
                                 void* workspace, size_t workspace_bytes, int N, int H, int W, int base_channels, int arch,
                                 void* stream)
{
    using namespace smvs;
    if (!packed || !imgs || !stage1 || !stage2 || !stage3 || !workspace) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (arch != 0 && arch != 1) return fail(SMVS_ERR_ARG, "arch must be 0 (unet) or 1 (fpn)");
    const size_t need = smvs_featnet_workspace_bytes(N, H, W, base_channels, arch);
    if (need == 0) return fail(SMVS_ERR_ARG, "image %dx%d must be a positive multiple of 4 in both dimensions", H, W);
    if (workspace_bytes < need) return fail(SMVS_ERR_ARG, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    const int c = base_channels;
    if ((long long)4 * c * H * W * 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "image too large");
    if ((long long)N * ((4 * c + FN_COT - 1) / FN_COT) > 65535) return fail(SMVS_ERR_ARG, "too many views for one launch grid");
    FnLayer L[FN_NL];
    const int nl = fn_layers(c, arch, L);
    const FnLayout lay = fn_layout(c, arch);
    const FnWorkspace ws = fn_workspace(N, H, W, c, arch);
    float* f = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    struct Step { const float* inA; const float* inB; int CB; float* out; int lin, lout; const float* up; };   // lin / lout: 0 full, 1 half, 2 quarter
    const Step trunk[8] = {
        {imgs, nullptr, 0, f + ws.t0, 0, 0, nullptr},          {f + ws.t0, nullptr, 0, f + ws.c0, 0, 0, nullptr},
        {f + ws.c0, nullptr, 0, f + ws.t1a, 0, 1, nullptr},    {f + ws.t1a, nullptr, 0, f + ws.t1b, 1, 1, nullptr},  {f + ws.t1b, nullptr, 0, f + ws.c1, 1, 1, nullptr},
        {f + ws.c1, nullptr, 0, f + ws.t2a, 1, 2, nullptr},    {f + ws.t2a, nullptr, 0, f + ws.t2b, 2, 2, nullptr},  {f + ws.t2b, nullptr, 0, f + ws.c2, 2, 2, nullptr}};
    const Step head_unet[7] = {
        {f + ws.c2, nullptr, 0, stage1, 2, 2, nullptr},
        {f + ws.c2, nullptr, 0, f + ws.up1, 2, 1, nullptr},    {f + ws.up1, f + ws.c1, 2 * c, f + ws.f1, 1, 1, nullptr},
        {f + ws.f1, nullptr, 0, stage2, 1, 1, nullptr},
        {f + ws.f1, nullptr, 0, f + ws.up2, 1, 0, nullptr},    {f + ws.up2, f + ws.c0, c, f + ws.f2, 0, 0, nullptr},
        {f + ws.f2, nullptr, 0, stage3, 0, 0, nullptr}};
    const Step head_fpn[5] = {
        {f + ws.c2, nullptr, 0, stage1, 2, 2, nullptr},
        {f + ws.c1, nullptr, 0, f + ws.f1, 1, 1, f + ws.c2},   {f + ws.f1, nullptr, 0, stage2, 1, 1, nullptr},
        {f + ws.c0, nullptr, 0, f + ws.f2, 0, 0, f + ws.f1},   {f + ws.f2, nullptr, 0, stage3, 0, 0, nullptr}};
    for (int i = 0; i < nl; ++i) {
        const Step& s = i < 8 ? trunk[i] : arch == 0 ? head_unet[i - 8] : head_fpn[i - 8];
        FnConvArgs a{};
        a.inA = s.inA; a.CA = L[i].cin - s.CB; a.inB = s.inB; a.CB = s.CB;
        a.w = packed + lay.w[i]; a.scale = packed + lay.scale[i]; a.shift = packed + lay.shift[i];
        a.out = s.out; a.up = s.up; a.Cout = L[i].cout; a.relu = L[i].relu;
        a.Hi = H >> s.lin; a.Wi = W >> s.lin; a.Ho = H >> s.lout; a.Wo = W >> s.lout;
        fn_launch(L[i], a, N, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "featnet_fwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

}  // extern "C"
