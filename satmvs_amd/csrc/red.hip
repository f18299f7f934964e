// red.hip -- native forward of the recurrent encoder-decoder regulariser, one height plane per call.
//
// Replaces slice_RED_Regularization.forward (/root/reference/modules/module.py:672-693) and, called
// once per plane, the loop body of RED_Regularization.forward (:625-644) with its ConvGRUCell2 cells
// (:6-58): three stride-2 3x3 convolutions (encoder), a 3x3 ConvGRU with GroupNorm(1,.) on every gate
// at each of the four scales, three stride-2 transposed convolutions with additive skips (decoder) and
// a final 3x3 transposed convolution to one channel.  Hidden sizes 8/16/32/64 as hard-coded in the
// reference (module.py:617-620).
//
// Round-1 implementation: direct float32 convolutions -- one lane per output
// pixel, 8 output channels per lane in registers, weights wave-uniform and read with scalar loads from a
// pre-packed buffer ([cout/8][cin][tap][8]), inputs through the buffer range check (zero padding for free),
// concat inputs (x,h) / (x,r*h) read from two tensors without materialising the concatenation, bias / ReLU /
// GroupNorm statistics (float64 atomics) fused in the epilogue.  24 launches per plane instead of ~60 in
// the stock PyTorch composite.  Layers with >= 32 output channels (gates of levels 2-4, candidates of levels
// 3-4, encoder conv2/conv3) run on the float32 MFMA implicit-GEMM kernel of mfma_conv.h.
#include <stdlib.h>

#include "smvs_device.h"
#include "smvs_host.h"
#include "mfma_conv.h"

namespace smvs {

constexpr int COT = 8;                       // output channels per lane
constexpr int NSLOT = 64;                    // GroupNorm statistics are accumulated in 64 partial slots per
                                             // (sample, norm group): ~36 same-address float64 atomics per slot
                                             // instead of ~18000 on one address (measured: 237 -> see profile)
constexpr int HID[4] = {8, 16, 32, 64};      // hidden sizes of conv_gru1..4

// ---- packed parameter buffer -------------------------------------------------------------------------------
// conv weights are stored [cog][cin][tap][COT] (cout padded to a multiple of COT with zeros); small
// vectors (bias, norm affine) are copied verbatim.  Offsets in floats.
struct RedLayout {
    int C;
    size_t conv_w[3];                        // conv1..3
    size_t gate_w[4], gate_b[4], rn_w[4], rn_b[4], un_w[4], un_b[4], out_w[4], out_b[4], on_w[4], on_b[4];
    size_t up_w[3];                          // upconv1..3 (index i -> upconv{i+1})
    size_t up2d_w, up2d_b;
    size_t conv_wm[3], gate_wm[4], out_wm[4]; // MFMA-order copies for the layers mfma_conv_ok() accepts
    size_t total;
};

__host__ __device__ inline size_t packed_conv_floats(int cin, int cout) { return (size_t)((cout + COT - 1) / COT) * cin * 9 * COT; }

static RedLayout red_layout(int C)
{
    RedLayout L{};
    L.C = C;
    size_t o = 0;
    const int enc_in[3] = {C, 16, 32}, enc_out[3] = {16, 32, 64};
    for (int i = 0; i < 3; ++i) { L.conv_w[i] = o; o += packed_conv_floats(enc_in[i], enc_out[i]); }
    const int xin[4] = {C, 16, 32, 64};
    for (int i = 0; i < 4; ++i) {
        const int hc = HID[i], cin = xin[i] + hc;
        L.gate_w[i] = o; o += packed_conv_floats(cin, 2 * hc);
        L.gate_b[i] = o; o += 2 * hc;
        L.rn_w[i] = o; o += hc;  L.rn_b[i] = o; o += hc;
        L.un_w[i] = o; o += hc;  L.un_b[i] = o; o += hc;
        L.out_w[i] = o; o += packed_conv_floats(cin, hc);
        L.out_b[i] = o; o += hc;
        L.on_w[i] = o; o += hc;  L.on_b[i] = o; o += hc;
    }
    const int up_in[3] = {16, 32, 64}, up_out[3] = {8, 16, 32};
    for (int i = 0; i < 3; ++i) { L.up_w[i] = o; o += packed_conv_floats(up_in[i], up_out[i]); }
    L.up2d_w = o; o += packed_conv_floats(8, 1);
    L.up2d_b = o; o += 8;
    for (int i = 0; i < 3; ++i) { L.conv_wm[i] = o; if (mfma_conv_ok(enc_in[i], 0, enc_out[i])) o += mfma_packed_floats(enc_in[i], enc_out[i], 9); }
    for (int i = 0; i < 4; ++i) {
        const int hc = HID[i], cx = xin[i];
        L.gate_wm[i] = o; if (mfma_conv_ok(cx, hc, 2 * hc)) o += mfma_packed_floats(cx + hc, 2 * hc, 9);
        L.out_wm[i] = o;  if (mfma_conv_ok(cx, hc, hc)) o += mfma_packed_floats(cx + hc, hc, 9);
    }
    L.total = o;
    return L;
}

// mode 0: nn.Conv2d weight (Cout,Cin,3,3); mode 1: nn.ConvTranspose2d weight (Cin,Cout,3,3) kept as
// scatter taps (stride-2 kernel); mode 2: ConvTranspose2d stride 1 -> equivalent correlation (taps flipped)
__global__ void pack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int cin, int cout, int mode)
{
    const int ncog = (cout + COT - 1) / COT;
    const int n = ncog * cin * 9 * COT;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = i % COT, k = (i / COT) % 9, ci = (i / (COT * 9)) % cin, cog = i / (COT * 9 * cin);
        const int co = cog * COT + j;
        float v = 0.0f;
        if (co < cout) {
            if (mode == 0) v = src[((size_t)co * cin + ci) * 9 + k];
            else if (mode == 1) v = src[((size_t)ci * cout + co) * 9 + k];
            else v = src[((size_t)ci * cout + co) * 9 + (8 - k)];
        }
        dst[i] = v;
    }
}

__global__ void copy_vec_kernel(const float* __restrict__ src, float* __restrict__ dst, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ---- convolution -----------------------------------------------------------------------------------------------
struct ConvArgs {
    const float* inA; int CA; float scaleA;  // first CA input channels, multiplied by scaleA (-1 feeds -cost)
    const float* inB; int CB;                // next CB input channels (hidden state or r*h); may be null
    const float* w;                          // packed [cog][CA+CB][9][COT]
    const float* bias;                       // (Cout) or null
    float* out;                              // (B,Cout,Ho,Wo)
    double* stats;                           // (B,ngroups,2) sum / sum of squares of the raw output, or null
    int ngroups;                             // 1, or 2 (gate conv: reset half / update half)
    int Cout, Hi, Wi, Ho, Wo, relu;
};

typedef const float __attribute__((address_space(4))) * cw_t;

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// STRIDE 1 or 2 correlation with pad 1; lane = one output pixel, COT output channels.
template <int STRIDE>
__global__ __launch_bounds__(256)
void conv3x3_kernel(const ConvArgs a)
{
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int oy = (blockIdx.y * 4 + (threadIdx.x >> 6));
    const int ncog = (a.Cout + COT - 1) / COT;
    const int cog = blockIdx.z % ncog, b = blockIdx.z / ncog;
    const bool active = ox < a.Wo && oy < a.Ho;
    const int Cin = a.CA + a.CB;
    const int HWi = a.Hi * a.Wi;

    // the 9 tap offsets of this lane inside one input plane (SMVS_OOB outside the image = zero padding)
    uint32_t off[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * STRIDE - 1 + ky, ix = ox * STRIDE - 1 + kx;
            off[ky * 3 + kx] = (active && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) ? (uint32_t)(iy * a.Wi + ix) * 4u : SMVS_OOB;
        }
    const BufRsrc rA = make_rsrc(a.inA + (size_t)b * a.CA * HWi, (uint32_t)a.CA * (uint32_t)HWi * 4u);
    const BufRsrc rB = make_rsrc(a.inB ? a.inB + (size_t)b * a.CB * HWi : a.inA, (uint32_t)(a.inB ? a.CB : 0) * (uint32_t)HWi * 4u);

    float acc[COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) acc[j] = 0.0f;
    const cw_t wbase = (cw_t)(uintptr_t)(a.w + (size_t)cog * Cin * 9 * COT);

    for (int ci = 0; ci < a.CA; ++ci) {
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = llvm_raw_buffer_load_f32(rA.v, (int)off[k], ci * HWi * 4, 0) * a.scaleA;
        const cw_t wc = wbase + (size_t)ci * 9 * COT;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int j = 0; j < COT; ++j) acc[j] = fmaf(v[k], wc[k * COT + j], acc[j]);
    }
    for (int ci = 0; ci < a.CB; ++ci) {
        float v[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = llvm_raw_buffer_load_f32(rB.v, (int)off[k], ci * HWi * 4, 0);
        const cw_t wc = wbase + (size_t)(a.CA + ci) * 9 * COT;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int j = 0; j < COT; ++j) acc[j] = fmaf(v[k], wc[k * COT + j], acc[j]);
    }

    float s1 = 0.0f, s2 = 0.0f;
    const int HWo = a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = cog * COT + j;
        if (co < a.Cout) {
            float r = acc[j] + (a.bias ? a.bias[co] : 0.0f);
            if (active) { s1 += r; s2 = fmaf(r, r, s2); }
            if (a.relu) r = fmaxf(r, 0.0f);
            if (active) a.out[((size_t)b * a.Cout + co) * HWo + (size_t)oy * a.Wo + ox] = r;
        }
    }
    if (a.stats) {
        // GroupNorm(1,.) statistics of the raw (pre-activation) output; a cout group of 8 lies inside one
        // norm group because every hidden size is a multiple of 8.  Wave reduce, workgroup reduce through
        // LDS, then ONE float64 atomic pair per workgroup into one of NSLOT partial slots.
        __shared__ float red[2][4];
        s1 = wave_sum_f(s1);
        s2 = wave_sum_f(s2);
        if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s1; red[1][threadIdx.x >> 6] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int grp = (a.ngroups == 2 && cog * COT >= a.Cout / 2) ? 1 : 0;
            const int slot = (blockIdx.x + blockIdx.y * 7 + cog * 13) % NSLOT;
            double* st = a.stats + (((size_t)b * a.ngroups + grp) * NSLOT + slot) * 2;
            atomicAdd(st, (double)red[0][0] + (double)red[0][1] + (double)red[0][2] + (double)red[0][3]);
            atomicAdd(st + 1, (double)red[1][0] + (double)red[1][1] + (double)red[1][2] + (double)red[1][3]);
        }
    }
}

// ConvTranspose2d(k=3, stride=2, pad=1, output_padding=1): lane = one INPUT position (y,x), producing the
// 2x2 output quad (2y..2y+1, 2x..2x+1) from inputs (y,x),(y,x+1),(y+1,x),(y+1,x+1):
//   out(2y  ,2x  ) = in(y,x) w[1][1]
//   out(2y  ,2x+1) = in(y,x) w[1][2] + in(y,x+1) w[1][0]
//   out(2y+1,2x  ) = in(y,x) w[2][1] + in(y+1,x) w[0][1]
//   out(2y+1,2x+1) = in(y,x) w[2][2] + in(y,x+1) w[2][0] + in(y+1,x) w[0][2] + in(y+1,x+1) w[0][0]
__global__ __launch_bounds__(256)
void convT3x3s2_kernel(const ConvArgs a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int ncog = (a.Cout + COT - 1) / COT;
    const int cog = blockIdx.z % ncog, b = blockIdx.z / ncog;
    const bool active = x < a.Wi && y < a.Hi;
    const int HWi = a.Hi * a.Wi;
    uint32_t off[4];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
            off[dy * 2 + dx] = (active && y + dy < a.Hi && x + dx < a.Wi) ? (uint32_t)((y + dy) * a.Wi + x + dx) * 4u : SMVS_OOB;
    const BufRsrc rA = make_rsrc(a.inA + (size_t)b * a.CA * HWi, (uint32_t)a.CA * (uint32_t)HWi * 4u);
    float acc[4][COT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[q][j] = 0.0f;
    const cw_t wbase = (cw_t)(uintptr_t)(a.w + (size_t)cog * a.CA * 9 * COT);
    for (int ci = 0; ci < a.CA; ++ci) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = llvm_raw_buffer_load_f32(rA.v, (int)off[k], ci * HWi * 4, 0);
        const cw_t wc = wbase + (size_t)ci * 9 * COT;
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            acc[0][j] = fmaf(v[0], wc[4 * COT + j], acc[0][j]);
            acc[1][j] = fmaf(v[0], wc[5 * COT + j], fmaf(v[1], wc[3 * COT + j], acc[1][j]));
            acc[2][j] = fmaf(v[0], wc[7 * COT + j], fmaf(v[2], wc[1 * COT + j], acc[2][j]));
            acc[3][j] = fmaf(v[0], wc[8 * COT + j], fmaf(v[1], wc[6 * COT + j],
                        fmaf(v[2], wc[2 * COT + j], fmaf(v[3], wc[0 * COT + j], acc[3][j]))));
        }
    }
    if (!active) return;
    const int HWo = a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = cog * COT + j;
        if (co < a.Cout) {
            float* o = a.out + ((size_t)b * a.Cout + co) * HWo + (size_t)(2 * y) * a.Wo + 2 * x;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float r = acc[q][j];
                if (a.relu) r = fmaxf(r, 0.0f);
                o[(q >> 1) * a.Wo + (q & 1)] = r;
            }
        }
    }
}

// ---- GRU element-wise stages -------------------------------------------------------------------------------------
// mean / rstd of one norm group from its NSLOT partial (sum, sumsq) slots.  Called by every thread of the
// workgroup with the same `st`: wave 0 reduces the slots, the result is broadcast through LDS.
__device__ __forceinline__ void gn_coeffs(const double* st, double n, float eps, float& mean, float& rstd, float* lds2)
{
    if (threadIdx.x < 64) {
        double a = st[threadIdx.x * 2], q = st[threadIdx.x * 2 + 1];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); q += __shfl_xor(q, m, 64); }
        if (threadIdx.x == 0) {
            const double mu = a / n;
            double var = q / n - mu * mu;
            var = var < 0.0 ? 0.0 : var;
            lds2[0] = (float)mu;
            lds2[1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    mean = lds2[0];
    rstd = lds2[1];
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// gates raw (B,2HC,h,w) -> rh = sigmoid(GN(r)) * h ; u (in place over the update half) = sigmoid(GN(u))
__global__ __launch_bounds__(256)
void gru_gate_apply_kernel(float* __restrict__ gates, const double* __restrict__ stats, const float* __restrict__ rn_w,
                           const float* __restrict__ rn_b, const float* __restrict__ un_w, const float* __restrict__ un_b,
                           const float* __restrict__ h, float* __restrict__ rh, int B, int HC, int HW)
{
    __shared__ float coef[2][2];
    const int b = blockIdx.y;
    float mr, sr, mu, su;
    gn_coeffs(stats + ((size_t)b * 2 + 0) * NSLOT * 2, (double)HC * HW, 1e-5f, mr, sr, coef[0]);
    gn_coeffs(stats + ((size_t)b * 2 + 1) * NSLOT * 2, (double)HC * HW, 1e-5f, mu, su, coef[1]);
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // index inside the sample
    if (j >= (size_t)HC * HW) return;
    const int c = (int)(j / HW), p = (int)(j % HW);
    const size_t i = (size_t)b * HC * HW + j;
    float* gr = gates + ((size_t)b * 2 * HC + c) * HW + p;
    float* gu = gates + ((size_t)b * 2 * HC + HC + c) * HW + p;
    const float r = sigmoidf_(fmaf((*gr - mr) * sr, rn_w[c], rn_b[c]));
    const float u = sigmoidf_(fmaf((*gu - mu) * su, un_w[c], un_b[c]));
    rh[i] = r * h[i];
    *gu = u;
}

// h' = u*h + (1-u)*tanh(GN(cand)); state <- h'; optionally sum_out = up + h' (decoder skip)
__global__ __launch_bounds__(256)
void gru_combine_kernel(const float* __restrict__ cand, const double* __restrict__ stats, const float* __restrict__ on_w,
                        const float* __restrict__ on_b, const float* __restrict__ gates, float* __restrict__ h,
                        const float* __restrict__ up, float* __restrict__ sum_out, int B, int HC, int HW)
{
    __shared__ float coef[2];
    const int b = blockIdx.y;
    float m, s;
    gn_coeffs(stats + (size_t)b * NSLOT * 2, (double)HC * HW, 1e-5f, m, s, coef);
    const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (size_t)HC * HW) return;
    const int c = (int)(j / HW), p = (int)(j % HW);
    const size_t i = (size_t)b * HC * HW + j;
    const float y = tanhf(fmaf((cand[i] - m) * s, on_w[c], on_b[c]));
    const float u = gates[((size_t)b * 2 * HC + HC + c) * HW + p];
    const float hn = u * h[i] + (1.0f - u) * y;
    h[i] = hn;
    if (sum_out) sum_out[i] = up[i] + hn;
}

// ---- host orchestration ----------------------------------------------------------------------------------------------
struct RedWorkspace {                        // offsets in floats into the caller's workspace
    size_t e[3], gates[4], rh[4], cand[4], up[3], sum[3], stats;   // stats: doubles, offset in floats (8-byte aligned)
    size_t total;
};

static RedWorkspace red_workspace(int B, int C, int H, int W)
{
    RedWorkspace w{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
    const int hs[4] = {H, H / 2, H / 4, H / 8}, ws[4] = {W, W / 2, W / 4, W / 8};
    const int ech[3] = {16, 32, 64};
    for (int i = 0; i < 3; ++i) w.e[i] = take((size_t)B * ech[i] * hs[i + 1] * ws[i + 1]);
    for (int i = 0; i < 4; ++i) {
        w.gates[i] = take((size_t)B * 2 * HID[i] * hs[i] * ws[i]);
        w.rh[i] = take((size_t)B * HID[i] * hs[i] * ws[i]);
        w.cand[i] = take((size_t)B * HID[i] * hs[i] * ws[i]);
    }
    for (int i = 0; i < 3; ++i) {            // up[i]: output of upconv{i+1} at level i ; sum[i] = up[i] + state{i+1}'
        w.up[i] = take((size_t)B * HID[i] * hs[i] * ws[i]);
        w.sum[i] = take((size_t)B * HID[i] * hs[i] * ws[i]);
    }
    w.stats = take((size_t)B * 4 * 3 * NSLOT * 2 * 2);   // 4 GRUs x (reset, update, output) x NSLOT x (sum, sumsq) doubles
    w.total = o;
    return w;
}

static bool g_red_direct_only()
{
    const char* e = getenv("SMVS_CONV_DIRECT");             // A/B switch: direct kernels only
    return e && e[0] == '1';
}

// `wm` = MFMA-order weights of the same layer (used when the layer qualifies)
static void launch_conv(int stride, const ConvArgs& a, int B, hipStream_t st, const float* wm = nullptr)
{
    if (wm && mfma_conv_ok(a.CA, a.CB, a.Cout) && !g_red_direct_only()) {
        MfmaConvArgs m{};
        m.inA = a.inA; m.CA = a.CA; m.inB = a.inB; m.CB = a.CB; m.scaleA = a.scaleA; m.w = wm; m.bias = a.bias;
        m.out = a.out; m.stats = a.stats; m.ngroups = a.ngroups; m.nslot = NSLOT;
        m.Cout = a.Cout; m.relu = a.relu; m.stride = stride;
        m.Di = m.Do = 1; m.Hi = a.Hi; m.Wi = a.Wi; m.Ho = a.Ho; m.Wo = a.Wo;
        mfma_conv_launch<9>(m, B, st);
        return;
    }
    const int ncog = (a.Cout + COT - 1) / COT;
    dim3 grd((a.Wo + 63) / 64, (a.Ho + 3) / 4, B * ncog), blk(256);
    if (stride == 1) hipLaunchKernelGGL(conv3x3_kernel<1>, grd, blk, 0, st, a);
    else             hipLaunchKernelGGL(conv3x3_kernel<2>, grd, blk, 0, st, a);
}

static void launch_convT(const ConvArgs& a, int B, hipStream_t st)
{
    const int ncog = (a.Cout + COT - 1) / COT;
    dim3 grd((a.Wi + 63) / 64, (a.Hi + 3) / 4, B * ncog), blk(256);
    hipLaunchKernelGGL(convT3x3s2_kernel, grd, blk, 0, st, a);
}

}  // namespace smvs

extern "C" {

SMVS_EXPORT size_t smvs_red_packed_floats(int C) { return C > 0 ? smvs::red_layout(C).total : 0; }

SMVS_EXPORT size_t smvs_red_workspace_bytes(int B, int C, int H, int W)
{
    if (B < 1 || C < 1 || H < 8 || W < 8) return 0;
    return smvs::red_workspace(B, C, H, W).total * sizeof(float);
}

// params: HOST array of 48 device pointers in this order (the reference's parameter names):
//   for g in conv_gru1..4: gate_conv.weight, gate_conv.bias, reset_gate_norm.weight, .bias,
//                          update_gate_norm.weight, .bias, output_conv.weight, .bias, output_norm.weight, .bias
//   conv1.conv.weight, conv2.conv.weight, conv3.conv.weight,
//   upconv1.conv.weight, upconv2.conv.weight, upconv3.conv.weight, upconv2d.weight, upconv2d.bias
SMVS_EXPORT int smvs_red_pack_weights(const float* const* params, int C, float* packed, void* stream)
{
    using namespace smvs;
    if (!params || !packed) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (C < 1) return fail(SMVS_ERR_ARG, "non-positive channel count");
    for (int i = 0; i < 48; ++i)
        if (!params[i]) return fail(SMVS_ERR_ARG, "null parameter pointer %d", i);
    const RedLayout L = red_layout(C);
    hipStream_t st = (hipStream_t)stream;
    auto pack = [&](const float* src, size_t dst, int cin, int cout, int mode) {
        const int n = (int)packed_conv_floats(cin, cout);
        hipLaunchKernelGGL(pack_conv_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, packed + dst, cin, cout, mode);
    };
    auto copy = [&](const float* src, size_t dst, int n) {
        hipLaunchKernelGGL(copy_vec_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, packed + dst, n);
    };
    auto packm = [&](const float* src, size_t dst, int cin, int cout) {
        const int n = (int)mfma_packed_floats(cin, cout, 9);
        hipLaunchKernelGGL(mfma_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, packed + dst, cin, cout, 9);
    };
    const int xin[4] = {C, 16, 32, 64};
    for (int g = 0; g < 4; ++g) {
        const float* const* q = params + g * 10;
        const int hc = HID[g], cin = xin[g] + hc;
        if (mfma_conv_ok(xin[g], hc, 2 * hc)) packm(q[0], L.gate_wm[g], cin, 2 * hc);
        if (mfma_conv_ok(xin[g], hc, hc)) packm(q[6], L.out_wm[g], cin, hc);
        pack(q[0], L.gate_w[g], cin, 2 * hc, 0);  copy(q[1], L.gate_b[g], 2 * hc);
        copy(q[2], L.rn_w[g], hc);  copy(q[3], L.rn_b[g], hc);
        copy(q[4], L.un_w[g], hc);  copy(q[5], L.un_b[g], hc);
        pack(q[6], L.out_w[g], cin, hc, 0);  copy(q[7], L.out_b[g], hc);
        copy(q[8], L.on_w[g], hc);  copy(q[9], L.on_b[g], hc);
    }
    const int enc_in[3] = {C, 16, 32}, enc_out[3] = {16, 32, 64};
    for (int i = 0; i < 3; ++i) {
        pack(params[40 + i], L.conv_w[i], enc_in[i], enc_out[i], 0);
        if (mfma_conv_ok(enc_in[i], 0, enc_out[i])) packm(params[40 + i], L.conv_wm[i], enc_in[i], enc_out[i]);
    }
    const int up_in[3] = {16, 32, 64}, up_out[3] = {8, 16, 32};
    for (int i = 0; i < 3; ++i) pack(params[43 + i], L.up_w[i], up_in[i], up_out[i], 1);
    pack(params[46], L.up2d_w, 8, 1, 2);
    copy(params[47], L.up2d_b, 1);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "red_pack_weights launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

// One plane of the recurrent regulariser.  cost (B,C,H,W) is the variance plane (the network consumes
// -cost, module.py:675); state1..4 (B,8,H,W) (B,16,H/2,W/2) (B,32,H/4,W/4) (B,64,H/8,W/8) are updated in
// place; reg_out (B,1,H,W).  H and W must be multiples of 8.  workspace: smvs_red_workspace_bytes bytes.
SMVS_EXPORT int smvs_red_step_fwd(const float* packed, const float* cost, float* state1, float* state2,
                                  float* state3, float* state4, float* reg_out, void* workspace, size_t workspace_bytes,
                                  int B, int C, int H, int W, void* stream)
{
    using namespace smvs;
    if (!packed || !cost || !state1 || !state2 || !state3 || !state4 || !reg_out || !workspace)
        return fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || C < 1 || H < 8 || W < 8 || (H % 8) || (W % 8))
        return fail(SMVS_ERR_ARG, "plane %dx%d must be a positive multiple of 8 in both dimensions", H, W);
    const RedLayout L = red_layout(C);
    const RedWorkspace ws = red_workspace(B, C, H, W);
    if (workspace_bytes < ws.total * sizeof(float)) return fail(SMVS_ERR_ARG, "workspace too small: %zu < %zu bytes", workspace_bytes, ws.total * sizeof(float));
    if ((long long)(C + 8) * H * W * 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "plane too large");
    hipStream_t st = (hipStream_t)stream;
    float* wsf = (float*)workspace;
    double* stats = (double*)(wsf + ws.stats);
    float* state[4] = {state1, state2, state3, state4};
    const int hs[4] = {H, H / 2, H / 4, H / 8}, wd[4] = {W, W / 2, W / 4, W / 8};
    hipMemsetAsync(stats, 0, (size_t)B * 4 * 3 * NSLOT * 2 * sizeof(double), st);

    // encoder: e1 = relu(conv1(-cost)), e2 = relu(conv2(e1)), e3 = relu(conv3(e2))
    const int enc_in[3] = {C, 16, 32}, enc_out[3] = {16, 32, 64};
    for (int i = 0; i < 3; ++i) {
        ConvArgs a{};
        a.inA = i == 0 ? cost : wsf + ws.e[i - 1]; a.CA = enc_in[i]; a.scaleA = i == 0 ? -1.0f : 1.0f;
        a.w = packed + L.conv_w[i]; a.out = wsf + ws.e[i]; a.Cout = enc_out[i];
        a.Hi = hs[i]; a.Wi = wd[i]; a.Ho = hs[i + 1]; a.Wo = wd[i + 1]; a.relu = 1;
        launch_conv(2, a, B, st, packed + L.conv_wm[i]);
    }
    // four GRU levels, coarse to fine, interleaved with the decoder
    for (int g = 3; g >= 0; --g) {
        const int hc = HID[g], hw = hs[g] * wd[g];
        const float* x = g == 0 ? cost : wsf + ws.e[g - 1];
        const int cx = g == 0 ? C : enc_out[g - 1];
        const float sx = g == 0 ? -1.0f : 1.0f;
        double* sg = stats + (size_t)g * B * 3 * NSLOT * 2;               // [b][reset,update][slot][2], then [b][slot][2] for the output norm
        double* so = sg + (size_t)B * 2 * NSLOT * 2;
        ConvArgs a{};
        a.inA = x; a.CA = cx; a.scaleA = sx; a.inB = state[g]; a.CB = hc;
        a.w = packed + L.gate_w[g]; a.bias = packed + L.gate_b[g]; a.out = wsf + ws.gates[g]; a.stats = sg; a.ngroups = 2;
        a.Cout = 2 * hc; a.Hi = a.Ho = hs[g]; a.Wi = a.Wo = wd[g];
        launch_conv(1, a, B, st, packed + L.gate_wm[g]);
        const size_t n = (size_t)hc * hw;                                 // per sample; blockIdx.y = sample
        hipLaunchKernelGGL(gru_gate_apply_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, wsf + ws.gates[g], sg,
                           packed + L.rn_w[g], packed + L.rn_b[g], packed + L.un_w[g], packed + L.un_b[g], state[g],
                           wsf + ws.rh[g], B, hc, hw);
        ConvArgs o{};
        o.inA = x; o.CA = cx; o.scaleA = sx; o.inB = wsf + ws.rh[g]; o.CB = hc;
        o.w = packed + L.out_w[g]; o.bias = packed + L.out_b[g]; o.out = wsf + ws.cand[g]; o.stats = so; o.ngroups = 1;
        o.Cout = hc; o.Hi = o.Ho = hs[g]; o.Wi = o.Wo = wd[g];
        launch_conv(1, o, B, st, packed + L.out_wm[g]);
        const bool skip = g < 3;                                          // levels 3,2,1 add the upsampled coarser level
        hipLaunchKernelGGL(gru_combine_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, wsf + ws.cand[g], so,
                           packed + L.on_w[g], packed + L.on_b[g], wsf + ws.gates[g], state[g],
                           skip ? wsf + ws.up[g] : nullptr, skip ? wsf + ws.sum[g] : nullptr, B, hc, hw);
        if (g > 0) {
            // decoder: up[g-1] = relu(upconv{g}(state4' or sum[g]))
            ConvArgs u{};
            u.inA = g == 3 ? state[3] : wsf + ws.sum[g]; u.CA = hc; u.scaleA = 1.0f;
            u.w = packed + L.up_w[g - 1]; u.out = wsf + ws.up[g - 1]; u.Cout = HID[g - 1];
            u.Hi = hs[g]; u.Wi = wd[g]; u.Ho = hs[g - 1]; u.Wo = wd[g - 1]; u.relu = 1;
            launch_convT(u, B, st);
        }
    }
    // reg = upconv2d(up1 + state1') : ConvTranspose2d stride 1 == correlation with flipped taps
    ConvArgs f{};
    f.inA = wsf + ws.sum[0]; f.CA = 8; f.scaleA = 1.0f; f.w = packed + L.up2d_w; f.bias = packed + L.up2d_b;
    f.out = reg_out; f.Cout = 1; f.Hi = f.Ho = H; f.Wi = f.Wo = W;
    launch_conv(1, f, B, st);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "red_step launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

// The whole plane loop of compute_depth_when_pred (networks/casred.py:191-231) for planes [d_begin,d_end)
// in ONE call: per plane the fused warp+variance build of that plane, the RED step and the float64
// streaming-regression update are enqueued back to back from C (about 27 launches per plane, no Python
// or allocator work in between).  acc = (3,B,H,W) float64 [exp_sum, depth_img, max_prob], zeroed by the
// caller before plane 0; states as in smvs_red_step_fwd.  geo_kind 0: rpc (B,V,170); 1: composed
// homographies (B,n_src,4,4).
SMVS_EXPORT size_t smvs_red_pred_workspace_bytes(int B, int C, int H, int W)
{
    const size_t r = smvs_red_workspace_bytes(B, C, H, W);
    if (r == 0) return 0;
    return r + ((size_t)B * C * H * W + (size_t)B * H * W) * sizeof(float) + 64;
}

SMVS_EXPORT int smvs_red_pred_planes(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                                     const double* geo, const float* depth, int depth_is_4d, const float* packed,
                                     float* state1, float* state2, float* state3, float* state4, double* acc,
                                     void* workspace, size_t workspace_bytes,
                                     int B, int C, int D, int H, int W, int d_begin, int d_end, void* stream)
{
    using namespace smvs;
    if (!acc || !workspace) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (geo_kind != 0 && geo_kind != 1) return fail(SMVS_ERR_ARG, "geo_kind must be 0 (rpc) or 1 (homography)");
    const size_t need = smvs_red_pred_workspace_bytes(B, C, H, W);
    if (need == 0) return fail(SMVS_ERR_ARG, "plane %dx%d must be a positive multiple of 8 in both dimensions", H, W);
    if (workspace_bytes < need) return fail(SMVS_ERR_ARG, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    if (d_begin < 0 || d_end > D || d_begin > d_end) return fail(SMVS_ERR_ARG, "bad plane range [%d,%d) of %d", d_begin, d_end, D);
    const size_t red_bytes = smvs_red_workspace_bytes(B, C, H, W);
    float* plane = (float*)((char*)workspace + ((red_bytes + 15) & ~(size_t)15));
    float* reg = plane + (size_t)B * C * H * W;
    const size_t n = (size_t)B * H * W;
    for (int d = d_begin; d < d_end; ++d) {
        int rc = geo_kind == 0
            ? smvs_rpc_costvol_fwd(ref_fea, src_fea, n_src, geo, depth, depth_is_4d, plane, B, C, D, H, W, d, d + 1, 1, 0, stream)
            : smvs_homo_costvol_fwd(ref_fea, src_fea, n_src, geo, depth, depth_is_4d, plane, B, C, D, H, W, d, d + 1, 1, 0, stream);
        if (rc) return rc;
        rc = smvs_red_step_fwd(packed, plane, state1, state2, state3, state4, reg, workspace, red_bytes, B, C, H, W, stream);
        if (rc) return rc;
        rc = smvs_stream_regress_step(reg, depth, depth_is_4d, acc, acc + n, acc + 2 * n, B, D, H, W, d, stream);
        if (rc) return rc;
    }
    return SMVS_OK;
}

}  // extern "C"
