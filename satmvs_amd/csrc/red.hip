// red.hip -- native inference forward of the recurrent encoder-decoder (RED) regulariser.
//
// Replaces slice_RED_Regularization.forward (/root/reference/modules/module.py:672-693) and, called
// once per plane, the loop body of RED_Regularization.forward (:625-644) with its ConvGRUCell2 cells
// (:6-58): three stride-2 3x3 convolutions (encoder), a 3x3 ConvGRU with GroupNorm(1,.) on every gate
// at each of the four scales, three stride-2 transposed convolutions with additive skips (decoder) and
// a final 3x3 transposed convolution to one channel.  Hidden sizes 8/16/32/64 as hard-coded in the
// reference (module.py:617-620).  SURVEY.md section 8 rows a10, a11, a14.
//
// Kernels: direct float32 convolutions for the < 32-channel layers -- one lane per output pixel, 8 output
// channels per lane in registers, weights wave-uniform and read with scalar loads from a pre-packed buffer
// ([cout/8][cin][tap][8]), inputs through the buffer range check (zero padding for free), concat inputs
// (x,h) / (x,r*h) read from two tensors without materialising the concatenation, bias / ReLU / GroupNorm
// statistics (float64 atomics into 64 slots) fused in the epilogue; a channel-split variant for the coarse
// planes.  Layers with >= 32 output channels (gates of levels 2-4, candidates of levels 3-4, encoder
// conv2/conv3) run on the float32 MFMA implicit-GEMM kernel of mfma_conv.h.  The same stage of the four ConvGRU
// levels is ONE launch (level-batched: conv_jobs_kernel, gru_gate_apply_kernel, gru_combine_kernel); the decoder fuses
// the skip add, the output convolution fuses the regression update: 8 launches per plane (+ 4 per chunk of planes for
// cost volume and encoder) instead of ~60 in the stock PyTorch composite.
//
// Entry points: smvs_red_step_fwd (one plane, caller's stream), smvs_red_pred_planes / smvs_red_volume_planes
// (the whole plane loop incl. the cost-volume planes, as a 3-stream pipeline: see red_run_planes).
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "smvs_device.h"
#include "smvs_host.h"
#include "mfma_conv.h"

namespace smvs {

constexpr int COT = 8;                       // output channels per lane
// Pipeline depths of the direct kernels.  Measured on the three cascade stages (tools/host_bound_probe.py, ms per
// stage): depth 2/3 + scalar weights 3.70 / 3.44 / 2.45; depth 4/6 3.58 / 3.63 / 2.57; depth 4/6 + weights through
// LDS 3.50 / 4.15 / 2.83 -- the deeper variants shave the 192x96 stage and cost more on the larger ones (registers,
// code size, an LDS copy per workgroup), so the shallow ones ship.
#ifndef SMVS_CONV_PREFETCH
#define SMVS_CONV_PREFETCH 2                 // channels of taps in flight per wave in the direct convolutions
#endif
#ifndef SMVS_WLDS
#define SMVS_WLDS 0                          // 1: channel-split direct kernels read their weights from LDS instead of scalar loads
#endif
#ifndef SMVS_CONVT_PREFETCH
#define SMVS_CONVT_PREFETCH 3
#endif
constexpr int NSLOT = 64;                    // GroupNorm statistics are accumulated in 64 partial slots per
                                             // (sample, norm group): ~36 same-address float64 atomics per slot
                                             // instead of ~18000 on one address (measured: 237 us -> 25 us for the full-resolution gate convolution)
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
    size_t out_bstride;                      // floats between samples of `out`; 0 = Cout*Ho*Wo (dense)
    const float* skip;                       // transposed convolution only: same shape as out, added after the ReLU (decoder skip), or null
    // strided view of inA (all 0 = dense (B,CA,Hi,Wi)): the pred loop reads variance planes straight out of a
    // (B,C,CH,H,W) chunk written by the cost-volume kernel.  Sample s of the launch = (batch s % inA_bmod, plane
    // s / inA_bmod) when inA_bmod > 0, else (batch s, plane 0); its first channel starts at
    // inA + batch*inA_bs + plane*inA_ps and channels are inA_cs floats apart.
    size_t inA_cs, inA_bs, inA_ps; int inA_bmod;
    // Partial convolution over a slice of the layer's input channels (the ConvGRU convolutions split into their state-independent
    // and state-dependent halves, see issue_front): the packed weights hold CinW input channels (0 = CA + CB), this launch uses
    // channels wc0 .. wc0 + CA + CB - 1 of them; `init` (same shape as out, or null) is added to the sums before the statistics.
    const float* init; int wc0, CinW;
};

typedef const float __attribute__((address_space(4))) * cw_t;

__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// STRIDE 1 or 2 correlation with pad 1; lane = one output pixel, COT output channels.
// SPLIT = false: workgroup = 64x4 output pixels, every wave walks all input channels (large planes:
//   throughput regime).
// SPLIT = true : workgroup = 64x1 output pixels, its 4 waves split the input channels and reduce through
//   LDS.  The coarse planes give a few dozen workgroups on a 256-CU part: there the time is one wave's
//   serial channel loop (a load round trip per channel), so 4x shorter chains = ~3x shorter kernels.
// Both: the taps of the next channels are in flight while channel c is multiplied.
// SPLIT with SMVS_WLDS stages the workgroup's weights ([Cin][9][COT], Cin <= WLDS_MAX_CIN) in LDS first (off: see above).
// (bx,by,bz) = the workgroup's grid coordinates (blockIdx of the plain launch; decoded from a flat index by the
// level-batched launch); smem: CONV_SMEM_FLOATS floats (SPLIT) / 8 floats.
constexpr int WLDS_MAX_CIN = 64;
constexpr int CONV_PART_FLOATS = 3 * COT * 64;
constexpr int WLDS_FLOATS = SMVS_WLDS ? WLDS_MAX_CIN * 9 * COT : 0;
constexpr int CONV_SMEM_FLOATS = CONV_PART_FLOATS + WLDS_FLOATS;
// FUSE (stride 1): the B operand is the element-wise ConvGRU stage `fz` (mfma_conv.h: FuseB), computed by the workgroup
// into `tile` (LDS, CB x CONV_FUSE_TH(SPLIT) x CONV_FUSE_TW floats) before the channel loop.
constexpr int CONV_FUSE_TW = 66;
constexpr int conv_fuse_th(bool split) { return split ? 3 : 6; }
template <int STRIDE, bool SPLIT, bool FUSE = false>
__device__ __forceinline__ void conv3x3_body(const ConvArgs& a, int bx, int by, int bz, float* smem, const FuseB* fz = nullptr, float* tile = nullptr)
{
    static_assert(!FUSE || STRIDE == 1, "fused element-wise stages feed the stride-1 ConvGRU convolutions");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ox = bx * 64 + lane;
    const int oy = SPLIT ? by : by * 4 + wave;
    const int ncog = (a.Cout + COT - 1) / COT;
    const int cog = bz % ncog, b = bz / ncog;
    const bool active = ox < a.Wo && oy < a.Ho;
    const int Cin = a.CA + a.CB;
    const int HWi = a.Hi * a.Wi;

    // Channel-split (latency) variant: the nine taps of a lane are three rows of three consecutive floats = one dwordx3
    // load per row (a row outside the image = out-of-range offset = zeros).  The loaded columns start at max(ix0, 0);
    // where the west or east tap is zero padding (image edge lanes only) the values are shifted / dropped with selects --
    // in the waves that have such lanes.  The throughput variant keeps nine dword taps (measured: row loads cost it 4 %).
    constexpr bool ROWS = SPLIT;
    uint32_t off[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * STRIDE - 1 + ky, ix = ox * STRIDE - 1 + kx;
            off[ky * 3 + kx] = (active && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) ? (uint32_t)(iy * a.Wi + ix) * 4u : SMVS_OOB;
        }
    const int ix0 = ox * STRIDE - 1;
    const bool padL = ix0 < 0, padR = ix0 + 2 >= a.Wi;
    const bool anypad = __builtin_amdgcn_ballot_w64(active && (padL || padR)) != 0;      // wave-uniform
    uint32_t roff[3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * STRIDE - 1 + ky;
        roff[ky] = (active && iy >= 0 && iy < a.Hi) ? (uint32_t)(iy * a.Wi + (padL ? 0 : ix0)) * 4u : SMVS_OOB;
    }
    const int csA = a.inA_cs ? (int)a.inA_cs : HWi;                 // floats between channels of inA
    const int bA = a.inA_bmod ? b % a.inA_bmod : b, pA = a.inA_bmod ? b / a.inA_bmod : 0;
    const BufRsrc rA = make_rsrc(a.inA + (a.inA_cs ? (size_t)bA * a.inA_bs + (size_t)pA * a.inA_ps : (size_t)b * a.CA * HWi),
                                 ((uint32_t)(a.CA - 1) * (uint32_t)csA + (uint32_t)HWi) * 4u);
    const BufRsrc rB = make_rsrc(a.inB ? a.inB + (size_t)b * a.CB * HWi : a.inA, (uint32_t)(a.inB ? a.CB : 0) * (uint32_t)HWi * 4u);

    float acc[COT], ini[COT];                // ini: the partial sums this launch continues (issued ahead of the channel loop)
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        acc[j] = 0.0f;
        const int co = cog * COT + j;
        ini[j] = (a.init && active && co < a.Cout && (!SPLIT || wave == 0))
                     ? a.init[(a.out_bstride ? (size_t)b * a.out_bstride + (size_t)co * a.Ho * a.Wo : ((size_t)b * a.Cout + co) * a.Ho * a.Wo) + (size_t)oy * a.Wo + ox] : 0.0f;
    }
    const int CinW = a.CinW ? a.CinW : Cin;
    const cw_t wbase = (cw_t)(uintptr_t)(a.w + ((size_t)cog * CinW + a.wc0) * 9 * COT);
    const float* wl = smem + CONV_PART_FLOATS;
    constexpr bool WL = SPLIT && SMVS_WLDS;
    if (WL) {
        const float* wg = a.w + ((size_t)cog * CinW + a.wc0) * 9 * COT;
        float* wd = smem + CONV_PART_FLOATS;
        for (int i = threadIdx.x * 4; i < Cin * 9 * COT; i += 256 * 4) *(float4*)(wd + i) = *(const float4*)(wg + i);
        __syncthreads();
    }

    const int per = SPLIT ? (Cin + 3) / 4 : Cin;
    const int c0 = SPLIT ? wave * per : 0, c1 = SPLIT ? min(Cin, c0 + per) : Cin;
    constexpr int FTH = conv_fuse_th(SPLIT);
    if constexpr (FUSE) {
        fuse_fill_tile<4, 8, FTH, CONV_FUSE_TW>(*fz, b, a.Hi, a.Wi, (SPLIT ? by : by * 4) - 1, bx * 64 - 1, tile, wave, cog == 0);
        if (fz->zero && bx == 0 && by == 0 && bz == 0)
            for (int i = threadIdx.x; i < fz->zero_n; i += 256) fz->zero[i] = 0.0;
        __syncthreads();
    }
    // NPF-1 channels of taps are in flight while one is multiplied.
    constexpr int NPF = SMVS_CONV_PREFETCH;
    struct Taps { mf32x3 r[ROWS ? 3 : 1]; float t[ROWS ? 1 : 9]; };   // one of the two is used (compile-time)
    Taps v[NPF];
    // taps of input channel cc: from memory (A half; B half when not fused) or from the LDS tile (fused B half)
    auto load = [&](Taps& V, int cc, auto lds_tag) {
        constexpr bool LDS = decltype(lds_tag)::value;
        if constexpr (LDS) {
            const float* tc = tile + (size_t)(cc - a.CA) * (FTH * CONV_FUSE_TW) + (SPLIT ? 0 : wave) * CONV_FUSE_TW + lane;
            if (ROWS) {
#pragma unroll
                for (int k = 0; k < 3; ++k) { V.r[k].x = tc[k * CONV_FUSE_TW]; V.r[k].y = tc[k * CONV_FUSE_TW + 1]; V.r[k].z = tc[k * CONV_FUSE_TW + 2]; }
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) V.t[k] = tc[(k / 3) * CONV_FUSE_TW + k % 3];
            }
        } else {
            const bool fa = cc < a.CA;                                 // wave-uniform: scalar selects
            i32x4 rx;
            rx.x = fa ? rA.v.x : rB.v.x; rx.y = fa ? rA.v.y : rB.v.y;
            rx.z = fa ? rA.v.z : rB.v.z; rx.w = rA.v.w;
            const int co = fa ? cc * csA * 4 : (cc - a.CA) * HWi * 4;
            if (ROWS) {
#pragma unroll
                for (int k = 0; k < 3; ++k) V.r[k] = llvm_raw_buffer_load_v3f32(rx, (int)roff[k], co, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 9; ++k) V.t[k] = llvm_raw_buffer_load_f32(rx, (int)off[k], co, 0);
            }
        }
    };
    auto fma = [&](const Taps& V, int cc, auto lds_tag) {
        constexpr bool LDS = decltype(lds_tag)::value;
        float t[9];
        const float sc = cc < a.CA ? a.scaleA : 1.0f;
        if (ROWS) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float l0 = V.r[k].x, l1 = V.r[k].y, l2 = V.r[k].z;
                if (!LDS && anypad) { l2 = padR ? 0.0f : (padL ? l1 : l2); l1 = padL ? l0 : l1; l0 = padL ? 0.0f : l0; }   // (the LDS tile carries its zero padding)
                t[3 * k] = l0 * sc; t[3 * k + 1] = l1 * sc; t[3 * k + 2] = l2 * sc;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 9; ++k) t[k] = V.t[k] * sc;
        }
        if (WL) {
            const float* wc = wl + cc * 9 * COT;
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int j = 0; j < COT; ++j) acc[j] = fmaf(t[k], wc[k * COT + j], acc[j]);
        } else {
            const cw_t wc = wbase + (size_t)cc * 9 * COT;
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int j = 0; j < COT; ++j) acc[j] = fmaf(t[k], wc[k * COT + j], acc[j]);
        }
    };
    // The loads are UNCONDITIONAL (past the end: the last channel again, unused) -- a load under a branch makes the
    // compiler wait with vmcnt(0) before every multiply block, and the prefetch hides nothing (mfma_conv.h).
    // Channels [lo, hi) of this wave in order (the accumulation order is the unfused kernel's).
    auto run = [&](int lo, int hi, auto lds_tag) {
        const int cl = max(hi - 1, lo);
#pragma unroll
        for (int i = 0; i < NPF - 1; ++i) load(v[i], min(lo + i, cl), lds_tag);
        for (int cc = lo; cc < hi; cc += NPF) {
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                load(v[(i + NPF - 1) % NPF], min(cc + i + NPF - 1, cl), lds_tag);
                __builtin_amdgcn_sched_barrier(0);
                if (cc + i < hi) fma(v[i], cc + i, lds_tag);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    if constexpr (FUSE) {
        if (c0 < min(c1, a.CA)) run(c0, min(c1, a.CA), std::false_type());
        if (max(c0, a.CA) < c1) run(max(c0, a.CA), c1, std::true_type());
    } else {
        run(c0, c1, std::false_type());
    }

    if (FUSE) __syncthreads();                // the B tile shares its LDS with the scratch below: every wave is out of its channel loop
    if (SPLIT) {
        float (*part)[COT][64] = (float (*)[COT][64])smem;
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < COT; ++j) part[wave - 1][j][lane] = acc[j];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int j = 0; j < COT; ++j) acc[j] += part[0][j][lane] + part[1][j][lane] + part[2][j][lane];
    }

    float s1 = 0.0f, s2 = 0.0f;
    const int HWo = a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = cog * COT + j;
        if (co < a.Cout) {
            float r = acc[j] + ini[j] + (a.bias ? a.bias[co] : 0.0f);
            if (active) { s1 += r; s2 = fmaf(r, r, s2); }
            if (a.relu) r = fmaxf(r, 0.0f);
            if (active) a.out[(a.out_bstride ? (size_t)b * a.out_bstride + (size_t)co * HWo : ((size_t)b * a.Cout + co) * HWo) + (size_t)oy * a.Wo + ox] = r;
        }
    }
    if (a.stats) {
        // GroupNorm(1,.) statistics of the raw (pre-activation) output; a cout group of 8 lies inside one
        // norm group because every hidden size is a multiple of 8.  Wave reduce, workgroup reduce through
        // LDS, then ONE float64 atomic pair per workgroup into one of NSLOT partial slots.
        float (*red)[4] = (float (*)[4])smem;
        s1 = wave_sum_f(s1);
        s2 = wave_sum_f(s2);
        const int grp = (a.ngroups == 2 && cog * COT >= a.Cout / 2) ? 1 : 0;
        const int slot = (bx + by * 7 + cog * 13) % NSLOT;
        double* st = a.stats + (((size_t)b * a.ngroups + grp) * NSLOT + slot) * 2;
        if (SPLIT) {
            if (lane == 0) { atomicAdd(st, (double)s1); atomicAdd(st + 1, (double)s2); }
        } else {
            if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                atomicAdd(st, (double)red[0][0] + (double)red[0][1] + (double)red[0][2] + (double)red[0][3]);
                atomicAdd(st + 1, (double)red[1][0] + (double)red[1][1] + (double)red[1][2] + (double)red[1][3]);
            }
        }
    }
}

// Stride-1 correlation, THROUGHPUT regime (the full-resolution levels of the late cascade stages: thousands of workgroups): lane =
// one output column, R vertically adjacent output rows, COT output channels.  The nine-tap variant above loads 9 dwords per input
// channel for 9*COT multiply-adds and is bound by the texture path (measured 30-47 us for the 768x384 level-1 gate convolution,
// 1.36 GFLOP: ~35 TFLOP/s); here the R+2 input rows a lane needs are loaded once and shared by its R outputs: 3(R+2) loads per
// 9*R*COT multiply-adds.  Same accumulation order per output (channel, tap) as the nine-tap variant: same bits.
// Workgroup = 64 columns x 4R rows (wave w: rows (4 by + w) R ..).
template <int R>
__device__ __forceinline__ void conv3x3_rows_body(const ConvArgs& a, int bx, int by, int bz, float* smem)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ox = bx * 64 + lane;
    const int oy0 = (by * 4 + wave) * R;
    const int ncog = (a.Cout + COT - 1) / COT;
    const int cog = bz % ncog, b = bz / ncog;
    const bool colok = ox < a.Wo;
    const int Cin = a.CA + a.CB;
    const int HWi = a.Hi * a.Wi;
    constexpr int NT = (R + 2) * 3;
    uint32_t off[NT];
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy0 - 1 + rr, ix = ox - 1 + kx;
            off[rr * 3 + kx] = (colok && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi) ? (uint32_t)(iy * a.Wi + ix) * 4u : SMVS_OOB;
        }
    const int csA = a.inA_cs ? (int)a.inA_cs : HWi;                 // floats between channels of inA
    const int bA = a.inA_bmod ? b % a.inA_bmod : b, pA = a.inA_bmod ? b / a.inA_bmod : 0;
    const BufRsrc rA = make_rsrc(a.inA + (a.inA_cs ? (size_t)bA * a.inA_bs + (size_t)pA * a.inA_ps : (size_t)b * a.CA * HWi),
                                 ((uint32_t)(a.CA - 1) * (uint32_t)csA + (uint32_t)HWi) * 4u);
    const BufRsrc rB = make_rsrc(a.inB ? a.inB + (size_t)b * a.CB * HWi : a.inA, (uint32_t)(a.inB ? a.CB : 0) * (uint32_t)HWi * 4u);
    const int HWo = a.Ho * a.Wo;
    float acc[R][COT], ini[R][COT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            acc[r][j] = 0.0f;
            const int co = cog * COT + j;
            ini[r][j] = (a.init && colok && oy0 + r < a.Ho && co < a.Cout)
                            ? a.init[(a.out_bstride ? (size_t)b * a.out_bstride + (size_t)co * HWo : ((size_t)b * a.Cout + co) * HWo) + (size_t)(oy0 + r) * a.Wo + ox] : 0.0f;
        }
    const int CinW = a.CinW ? a.CinW : Cin;
    const cw_t wbase = (cw_t)(uintptr_t)(a.w + ((size_t)cog * CinW + a.wc0) * 9 * COT);
    constexpr int NPF = 2;
    struct Taps { float t[NT]; };
    Taps v[NPF];
    auto load = [&](Taps& V, int cc) {
        const bool fa = cc < a.CA;                                 // wave-uniform: scalar selects
        i32x4 rx;
        rx.x = fa ? rA.v.x : rB.v.x; rx.y = fa ? rA.v.y : rB.v.y;
        rx.z = fa ? rA.v.z : rB.v.z; rx.w = rA.v.w;
        const int co = fa ? cc * csA * 4 : (cc - a.CA) * HWi * 4;
#pragma unroll
        for (int i = 0; i < NT; ++i) V.t[i] = llvm_raw_buffer_load_f32(rx, (int)off[i], co, 0);
    };
    auto fma = [&](const Taps& V, int cc) {
        const float sc = cc < a.CA ? a.scaleA : 1.0f;
        float t[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) t[i] = V.t[i] * sc;
        const cw_t wc = wbase + (size_t)cc * 9 * COT;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < COT; ++j) acc[r][j] = fmaf(t[(r + k / 3) * 3 + k % 3], wc[k * COT + j], acc[r][j]);
    };
    // unconditional loads (past the end: the last channel again, unused), see conv3x3_body
    const int cl = Cin - 1;
    load(v[0], 0);
    for (int cc = 0; cc < Cin; cc += NPF) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            load(v[(i + 1) % NPF], min(cc + i + 1, cl));
            __builtin_amdgcn_sched_barrier(0);
            if (cc + i < Cin) fma(v[i], cc + i);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int oy = oy0 + r;
        const bool active = colok && oy < a.Ho;
#pragma unroll
        for (int j = 0; j < COT; ++j) {
            const int co = cog * COT + j;
            if (co < a.Cout) {
                float q = acc[r][j] + ini[r][j] + (a.bias ? a.bias[co] : 0.0f);
                if (active) { s1 += q; s2 = fmaf(q, q, s2); }
                if (a.relu) q = fmaxf(q, 0.0f);
                if (active) a.out[(a.out_bstride ? (size_t)b * a.out_bstride + (size_t)co * HWo : ((size_t)b * a.Cout + co) * HWo) + (size_t)oy * a.Wo + ox] = q;
            }
        }
    }
    if (a.stats) {
        float (*red)[4] = (float (*)[4])smem;
        s1 = wave_sum_f(s1);
        s2 = wave_sum_f(s2);
        const int grp = (a.ngroups == 2 && cog * COT >= a.Cout / 2) ? 1 : 0;
        const int slot = (bx + by * 7 + cog * 13) % NSLOT;
        double* st = a.stats + (((size_t)b * a.ngroups + grp) * NSLOT + slot) * 2;
        if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(st, (double)red[0][0] + (double)red[0][1] + (double)red[0][2] + (double)red[0][3]);
            atomicAdd(st + 1, (double)red[1][0] + (double)red[1][1] + (double)red[1][2] + (double)red[1][3]);
        }
    }
}

template <int R>
__global__ __launch_bounds__(256)
void conv3x3_rows_kernel(const ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[8];
    conv3x3_rows_body<R>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

template <int STRIDE, bool SPLIT>
__global__ __launch_bounds__(256)
void conv3x3_kernel(const ConvArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[SPLIT ? CONV_SMEM_FLOATS : 8];
    conv3x3_body<STRIDE, SPLIT>(a, blockIdx.x, blockIdx.y, blockIdx.z, smem);
}

// ConvTranspose2d(k=3, stride=2, pad=1, output_padding=1): lane = one INPUT position (y,x), producing the
// 2x2 output quad (2y..2y+1, 2x..2x+1) from inputs (y,x),(y,x+1),(y+1,x),(y+1,x+1):
//   out(2y  ,2x  ) = in(y,x) w[1][1]
//   out(2y  ,2x+1) = in(y,x) w[1][2] + in(y,x+1) w[1][0]
//   out(2y+1,2x  ) = in(y,x) w[2][1] + in(y+1,x) w[0][1]
//   out(2y+1,2x+1) = in(y,x) w[2][2] + in(y,x+1) w[2][0] + in(y+1,x) w[0][2] + in(y+1,x+1) w[0][0]
// SPLIT as in conv3x3_kernel (64x1 input positions per workgroup, 4 waves split the input channels).
template <bool SPLIT>
__global__ __launch_bounds__(256)
void convT3x3s2_kernel(const ConvArgs a)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = blockIdx.x * 64 + lane;
    const int y = SPLIT ? (int)blockIdx.y : (int)blockIdx.y * 4 + wave;
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
    __shared__ __attribute__((aligned(16))) float ct_smem[SPLIT ? 3 * 4 * COT * 64 + WLDS_FLOATS : 4];
    const float* wl = ct_smem + 3 * 4 * COT * 64;
    constexpr bool WL = SPLIT && SMVS_WLDS;
    if (WL) {                                    // weights of this cout group through LDS (see conv3x3_body)
        const float* wg = a.w + (size_t)cog * a.CA * 9 * COT;
        float* wd = ct_smem + 3 * 4 * COT * 64;
        for (int i = threadIdx.x * 4; i < a.CA * 9 * COT; i += 256 * 4) *(float4*)(wd + i) = *(const float4*)(wg + i);
        __syncthreads();
    }
    const int per = SPLIT ? (a.CA + 3) / 4 : a.CA;
    const int c0 = SPLIT ? wave * per : 0, c1 = SPLIT ? min(a.CA, c0 + per) : a.CA;
#define SMVS_CT_LOAD(V, CC) \
    { _Pragma("unroll") for (int k_ = 0; k_ < 4; ++k_) V[k_] = llvm_raw_buffer_load_f32(rA.v, (int)off[k_], (CC) * HWi * 4, 0); }
#define SMVS_CT_FMA(V, CC)                                                                             \
    {                                                                                                  \
        if (WL) { const float* wc = wl + (CC) * 9 * COT; SMVS_CT_FMA_BODY(V, wc) }                     \
        else { const cw_t wc = wbase + (size_t)(CC) * 9 * COT; SMVS_CT_FMA_BODY(V, wc) }               \
    }
#define SMVS_CT_FMA_BODY(V, wc)                                                                        \
    {                                                                                                  \
        _Pragma("unroll") for (int j = 0; j < COT; ++j) {                                              \
            acc[0][j] = fmaf(V[0], wc[4 * COT + j], acc[0][j]);                                        \
            acc[1][j] = fmaf(V[0], wc[5 * COT + j], fmaf(V[1], wc[3 * COT + j], acc[1][j]));           \
            acc[2][j] = fmaf(V[0], wc[7 * COT + j], fmaf(V[2], wc[1 * COT + j], acc[2][j]));           \
            acc[3][j] = fmaf(V[0], wc[8 * COT + j], fmaf(V[1], wc[6 * COT + j],                        \
                        fmaf(V[2], wc[2 * COT + j], fmaf(V[3], wc[0 * COT + j], acc[3][j]))));         \
        }                                                                                              \
    }
    // NPF-1 channels of taps in flight ahead of the multiply (4 loads per channel only)
    constexpr int NPF = SMVS_CONVT_PREFETCH;
    float v[NPF][4];
    const int cl = max(c1 - 1, 0);               // unconditional loads (see conv3x3_body)
#pragma unroll
    for (int i = 0; i < NPF - 1; ++i) SMVS_CT_LOAD(v[i], min(c0 + i, cl))
    for (int cc = c0; cc < c1; cc += NPF) {
#pragma unroll
        for (int i = 0; i < NPF; ++i) {
            SMVS_CT_LOAD(v[(i + NPF - 1) % NPF], min(cc + i + NPF - 1, cl))
            __builtin_amdgcn_sched_barrier(0);
            if (cc + i < c1) SMVS_CT_FMA(v[i], cc + i)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#undef SMVS_CT_LOAD
#undef SMVS_CT_FMA
#undef SMVS_CT_FMA_BODY
    if (SPLIT) {
        float (*part)[4 * COT][64] = (float (*)[4 * COT][64])ct_smem;
        if (wave > 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < COT; ++j) part[wave - 1][q * COT + j][lane] = acc[q][j];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < COT; ++j)
                acc[q][j] += part[0][q * COT + j][lane] + part[1][q * COT + j][lane] + part[2][q * COT + j][lane];
    }
    if (!active) return;
    const int HWo = a.Ho * a.Wo;
    // The output quad is two float2 (columns 2x, 2x+1 of rows 2y, 2y+1; 8-byte aligned: Wo is even).  All skip loads are
    // issued before the first use -- a load -> add -> store chain per value is one memory round trip each on these
    // small grids.  Channels beyond Cout (cout group padding) read channel 0 and are not stored.
    float2 sk[2][COT];
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = cog * COT + j, cs = co < a.Cout ? co : 0;
        const size_t o0 = ((size_t)b * a.Cout + cs) * HWo + (size_t)(2 * y) * a.Wo + 2 * x;
#pragma unroll
        for (int ry = 0; ry < 2; ++ry)
            sk[ry][j] = a.skip ? *reinterpret_cast<const float2*>(a.skip + o0 + (size_t)ry * a.Wo) : make_float2(0.0f, 0.0f);
    }
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = cog * COT + j;
        if (co < a.Cout) {
            const size_t o0 = ((size_t)b * a.Cout + co) * HWo + (size_t)(2 * y) * a.Wo + 2 * x;
#pragma unroll
            for (int ry = 0; ry < 2; ++ry) {
                float r0 = acc[2 * ry][j], r1 = acc[2 * ry + 1][j];
                if (a.relu) { r0 = fmaxf(r0, 0.0f); r1 = fmaxf(r1, 0.0f); }
                if (a.skip) { r0 = r0 + sk[ry][j].x; r1 = r1 + sk[ry][j].y; }
                *reinterpret_cast<float2*>(a.out + o0 + (size_t)ry * a.Wo) = make_float2(r0, r1);
            }
        }
    }
}

// Output convolution (8 -> 1 channel, ConvTranspose2d stride 1 packed as a correlation) with the streaming
// regression update of the pred loop (networks/casred.py:218-231, float64 accumulators) in its epilogue: the plane
// never goes to memory unless the caller wants the regularised volume.  lane = one pixel, 72 taps in flight.
struct OutConvArgs {
    const float* in;                         // (NP*B,8,H,W): sample = plane * B + batch
    int NP;                                  // planes of the chunk (height hypotheses d .. d+NP-1), folded in plane order
    const float* w;                          // packed [8][9][COT], output channel 0
    const float* bias;
    float* reg; size_t reg_bstride, reg_pstride;   // regularised planes (floats between batch samples / between planes), or null
    double *exp_sum, *depth_img, *max_prob;  // (B,H,W) accumulators, or null (no regression update)
    const float* depth; int hmode; HeightGen hg; int D, d;
    int B, H, W;
};

__global__ __launch_bounds__(256)
void out_conv_regress_kernel(const OutConvArgs a)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ox = blockIdx.x * 64 + lane, oy = blockIdx.y * 4 + wave, b = blockIdx.z;
    const bool active = ox < a.W && oy < a.H;
    const int HW = a.H * a.W;
    uint32_t off[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int iy = oy - 1 + k / 3, ix = ox - 1 + k % 3;
        off[k] = (active && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) ? (uint32_t)(iy * a.W + ix) * 4u : SMVS_OOB;
    }
    const cw_t w = (cw_t)(uintptr_t)a.w;
    const int pix = oy * a.W + ox;
    const size_t i = (size_t)b * HW + (active ? pix : 0);
    // the regression accumulators are read once, folded over the chunk's planes in plane order, and written once
    double m = 0.0, di = 0.0, es = 0.0;
    if (a.exp_sum && active) { m = a.max_prob[i]; di = a.depth_img[i]; es = a.exp_sum[i]; }
    HeightPix hpx;
    if (a.exp_sum && a.hmode == HEIGHT_GENERATED && active) hg_prepare(a.hg, b, oy, ox, hpx);
    for (int pl = 0; pl < a.NP; ++pl) {
        const BufRsrc rA = make_rsrc(a.in + ((size_t)pl * a.B + b) * 8 * HW, (uint32_t)(8 * HW) * 4u);
        float v[8][9];
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c][k] = llvm_raw_buffer_load_f32(rA.v, (int)off[k], c * HW * 4, 0);
        float acc = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(v[c][k], w[(c * 9 + k) * COT], acc);
        if (!active) continue;
        const float r = acc + a.bias[0];
        const int d = a.d + pl;
        if (a.reg) a.reg[(size_t)b * a.reg_bstride + (size_t)pl * a.reg_pstride + pix] = r;
        if (a.exp_sum) {
            const double pr = exp((double)r);
            double hv;
            if (a.hmode == HEIGHT_GENERATED) hv = (double)hg_height(a.hg, hpx, d);
            else hv = a.hmode == HEIGHT_TENSOR ? (double)a.depth[((size_t)b * a.D + d) * HW + pix] : (double)a.depth[(size_t)b * a.D + d];
            m = (m < pr) ? pr : m;
            di = fma(hv, pr, di);
            es = es + pr;
        }
    }
    if (a.exp_sum && active) { a.max_prob[i] = m; a.depth_img[i] = di; a.exp_sum[i] = es; }
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

// two norm groups at once: wave 0 reduces st0, wave 1 reduces st1 (one barrier instead of two dependent ones)
__device__ __forceinline__ void gn_coeffs2(const double* st0, const double* st1, double n, float eps,
                                           float& m0, float& r0, float& m1, float& r1, float (*lds2)[2])
{
    if (threadIdx.x < 128) {
        const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
        const double* st = w ? st1 : st0;
        double a = st[l * 2], q = st[l * 2 + 1];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); q += __shfl_xor(q, m, 64); }
        if (l == 0) {
            const double mu = a / n;
            double var = q / n - mu * mu;
            var = var < 0.0 ? 0.0 : var;
            lds2[w][0] = (float)mu;
            lds2[w][1] = (float)(1.0 / sqrt(var + (double)eps));
        }
    }
    __syncthreads();
    m0 = lds2[0][0]; r0 = lds2[0][1]; m1 = lds2[1][0]; r1 = lds2[1][1];
}

// Gate activations of the shipped element-wise kernels: expf / tanhf and a true division (the fast forms of the fused
// tuning-build launches, mfma_conv.h -- rcp + __expf, tanh as 1 - 2 / (1 + e^2x) -- lose bits in every one of up to 384 recurrent steps
// and bought nothing on these memory-bound kernels).  SMVS_RED_FAST_ACT=1 (profiling builds) keeps the fast forms.
#ifndef SMVS_RED_FAST_ACT
#define SMVS_RED_FAST_ACT 0
#endif
__device__ __forceinline__ float sigmoidf_(float x) { return SMVS_RED_FAST_ACT ? fuse_sigmoid(x) : 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return SMVS_RED_FAST_ACT ? fuse_tanh(x) : tanhf(x); }

// ---- level-batched launches ------------------------------------------------------------------------------------------
// The four ConvGRU levels of a plane do not depend on each other (only the decoder crosses levels), and each of
// their kernels is a latency chain of a few dozen to a few hundred workgroups.  So one launch carries the same
// stage of ALL levels: a flat grid whose workgroups look up their job (level) and run the ordinary kernel body.
// Per plane: gate convolutions, gate apply, candidate convolutions, combine = 4 launches instead of 16, and the
// levels run side by side without streams or events.
struct ConvJob {
    ConvArgs a; MfmaConvArgs m;              // kind 0/1: a (direct, unsplit / channel-split); kind 2: m (MFMA, one cout tile per workgroup)
    int kind, gx, gy, blk0;                  // grid (gx, gy, rest) flattened; workgroups [blk0, next job's blk0)
    FuseB fuse;                              // fused launches: the element-wise stage this convolution computes for its B operand (direct kinds; MFMA kinds: m.fuse)
};
struct ConvJobs { ConvJob j[4]; int n; };

constexpr int JOBS_SMEM_FLOATS = CONV_SMEM_FLOATS > 3 * 16 * 64 ? CONV_SMEM_FLOATS : 3 * 16 * 64;   // direct split (6 KiB + weights) | MFMA split-K partials of 3 waves (12 KiB)

__global__ __launch_bounds__(256)
void conv_jobs_kernel(const ConvJobs J)
{
    __shared__ __attribute__((aligned(16))) float smem[JOBS_SMEM_FLOATS];
    int bid = blockIdx.x, l = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < J.n && bid >= J.j[i].blk0) l = i;
    const ConvJob& jb = J.j[l];
    bid -= jb.blk0;
    const int bx = bid % jb.gx, t = bid / jb.gx;
    if (jb.kind == 4) mfma_conv_body<9, 1, 4, 1>(jb.m, bid, 0, smem);        // MFMA, throughput regime: a (tile, cout tile) unit per wave
    else if (jb.kind == 3) mfma_conv_body<9, 1, 4, 2>(jb.m, bid, 0, smem);   // two units per workgroup, K split in two
    else if (jb.kind == 2) mfma_conv_body<9, 1, 4>(jb.m, bx, t, smem);
    else if (jb.kind == 1) conv3x3_body<1, true>(jb.a, bx, t % jb.gy, t / jb.gy, smem);
    else if (jb.kind == 6) conv3x3_rows_body<4>(jb.a, bx, t % jb.gy, t / jb.gy, smem);
    else if (jb.kind == 5) conv3x3_rows_body<2>(jb.a, bx, t % jb.gy, t / jb.gy, smem);
    else conv3x3_body<1, false>(jb.a, bx, t % jb.gy, t / jb.gy, smem);
}

// Fused launches (conv_jobs_fused_kernel): the convolution computes the element-wise ConvGRU stage that feeds its B operand
// into LDS first (mfma_conv.h: FuseB) -- 2 dependent launches per plane instead of 4.  LDS: the job bodies' own scratch +
// the largest B tile (MFMA, 64 channels x 3 x 34; direct 16 x 3 x 66 / 8 x 6 x 66).
constexpr int FUSE_TILE_FLOATS = 64 * MFMA_FUSE_TH * MFMA_FUSE_TW;
__host__ __device__ inline int fuse_tile_floats(int kind, int CB)
{
    return kind == 0 ? CB * conv_fuse_th(false) * CONV_FUSE_TW : kind == 1 ? CB * conv_fuse_th(true) * CONV_FUSE_TW
         : (kind == 3 ? 2 : 1) * CB * MFMA_FUSE_TH * MFMA_FUSE_TW;
}

__global__ __launch_bounds__(256)
void conv_jobs_fused_kernel(const ConvJobs J)
{
    // one allocation: the B tile is dead when the bodies' scratch (split-K partials, statistics) comes into use
    __shared__ __attribute__((aligned(16))) float smem[JOBS_SMEM_FLOATS > FUSE_TILE_FLOATS ? JOBS_SMEM_FLOATS : FUSE_TILE_FLOATS];
    float* tile = smem;
    int bid = blockIdx.x, l = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < J.n && bid >= J.j[i].blk0) l = i;
    const ConvJob& jb = J.j[l];
    bid -= jb.blk0;
    const int bx = bid % jb.gx, t = bid / jb.gx;
    if (jb.kind == 3) mfma_conv_body<9, 1, 4, 2, true>(jb.m, bid, 0, smem, tile);
    else if (jb.kind == 2) mfma_conv_body<9, 1, 4, 4, true>(jb.m, bx, t, smem, tile);
    else if (jb.kind == 1) conv3x3_body<1, true, true>(jb.a, bx, t % jb.gy, t / jb.gy, smem, &jb.fuse, tile);
    else conv3x3_body<1, false, true>(jb.a, bx, t % jb.gy, t / jb.gy, smem, &jb.fuse, tile);
}

struct GruJob {                              // the element-wise stages of one level
    const float* gates;                      // raw gate convolution (B,2HC,h,w): reset half read by the apply stage, update half by the combine stage
    const double *stats_g, *stats_o;         // [b][reset,update][NSLOT][2] ; [b][NSLOT][2]
    const float *rn_w, *rn_b, *un_w, *un_b, *on_w, *on_b;
    float* h;                                // hidden state (B,HC,h,w) read by both stages
    float* h_out;                            // where the combine stage stores the new state (h itself: in place)
    float* rh;                               // r*h for the candidate convolution
    const float* cand;                       // raw candidate convolution
    float* hsnap;                            // copy of the new state for the decoder (which runs while the next plane updates h)
    double* zero_next;                       // next plane's statistics of this level, cleared by the apply stage
    int HC, HW, gx, blk0;                    // gx workgroups per sample; workgroups [blk0, blk0 + gx*B)
    int vec;                                 // 1: four elements per thread (HW % 4 == 0 and 16-byte aligned tensors), 0: one
};
struct GruJobs { GruJob j[4]; int n, B; };

// E consecutive elements of one channel per thread (E = 4: 16-byte accesses; the job's `vec` says whether its plane size and
// pointers allow it).  One element per thread made every workgroup pay the statistics reduction (64 float64 slots, a division and
// a square root: ~1.5 us of latency) for 256 elements of work: at the full-resolution stage the two kernels ran at 1.6 / 2.3 TB/s
// (23 + 27 us per plane).  Same arithmetic per element: the bits do not depend on E.
template <int E> struct fvec;
template <> struct fvec<1> { float v[1]; };
template <> struct __attribute__((aligned(16))) fvec<4> { float v[4]; };

// gates raw -> rh = sigmoid(GN(r)) * h.  (The update gate is normalised by the combine stage straight from the raw
// convolution output: writing u here and reading it back there was 2 of this pass's 5 memory streams -- round 3.)
template <int E>
__device__ __forceinline__ void gru_gate_apply_body(const GruJob& g, int bx, int b, int nB, float* coef)
{
    const int HC = g.HC, HW = g.HW;
    // element operands first (they do not depend on the norm coefficients): their latency runs under the reduction
    const size_t jr = ((size_t)bx * blockDim.x + threadIdx.x) * E;     // index inside the sample
    const bool valid = jr < (size_t)HC * HW;
    const size_t j = valid ? jr : 0;
    const int c = (int)(j / HW), p = (int)(j % HW);
    const size_t i = (size_t)b * HC * HW + j;
    const fvec<E> vr = *reinterpret_cast<const fvec<E>*>(g.gates + ((size_t)b * 2 * HC + c) * HW + p);
    const fvec<E> vh = *reinterpret_cast<const fvec<E>*>(g.h + i);
    const float wr = g.rn_w[c], br = g.rn_b[c];
    float mr, sr;
    gn_coeffs(g.stats_g + ((size_t)b * 2 + 0) * NSLOT * 2, (double)HC * HW, 1e-5f, mr, sr, coef);
    if (!valid) return;
    fvec<E> o;
#pragma unroll
    for (int e = 0; e < E; ++e) o.v[e] = sigmoidf_(fmaf((vr.v[e] - mr) * sr, wr, br)) * vh.v[e];
    *reinterpret_cast<fvec<E>*>(g.rh + i) = o;
}

__global__ __launch_bounds__(256)
void gru_gate_apply_kernel(const GruJobs J)
{
    __shared__ float coef[2];
    int bid = blockIdx.x, l = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < J.n && bid >= J.j[i].blk0) l = i;
    const GruJob& g = J.j[l];
    bid -= g.blk0;
    const int bx = bid % g.gx, b = bid / g.gx;
    // the next plane's statistics of this level (other ring entry) are cleared here: every reader of that buffer
    // (an older plane's apply / combine kernels) precedes this kernel on the stream, every writer (the next plane's
    // convolutions) follows it.
    if (g.zero_next && bid == 0)
        for (int i = threadIdx.x; i < J.B * 3 * NSLOT * 2; i += blockDim.x) g.zero_next[i] = 0.0;
    if (g.vec) gru_gate_apply_body<4>(g, bx, b, J.B, coef);
    else       gru_gate_apply_body<1>(g, bx, b, J.B, coef);
}

// u = sigmoid(GN(update gate)); h' = u*h + (1-u)*tanh(GN(cand)); state <- h'; hsnap <- h'
template <int E>
__device__ __forceinline__ void gru_combine_body(const GruJob& g, int bx, int b, float (*coef)[2])
{
    const int HC = g.HC, HW = g.HW;
    const size_t jr = ((size_t)bx * blockDim.x + threadIdx.x) * E;
    const bool valid = jr < (size_t)HC * HW;
    const size_t j = valid ? jr : 0;
    const int c = (int)(j / HW), p = (int)(j % HW);
    const size_t i = (size_t)b * HC * HW + j;
    const fvec<E> vc = *reinterpret_cast<const fvec<E>*>(g.cand + i);            // before the reduction: see the apply kernel
    const fvec<E> vu = *reinterpret_cast<const fvec<E>*>(g.gates + ((size_t)b * 2 * HC + HC + c) * HW + p);
    const fvec<E> vh = *reinterpret_cast<const fvec<E>*>(g.h + i);
    const float wu = g.un_w[c], bu = g.un_b[c], wo = g.on_w[c], bo = g.on_b[c];
    float mu, su, m, s;
    gn_coeffs2(g.stats_g + ((size_t)b * 2 + 1) * NSLOT * 2, g.stats_o + (size_t)b * NSLOT * 2, (double)HC * HW, 1e-5f, mu, su, m, s, coef);
    if (!valid) return;
    fvec<E> o;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const float u = sigmoidf_(fmaf((vu.v[e] - mu) * su, wu, bu));
        const float y = tanhf_(fmaf((vc.v[e] - m) * s, wo, bo));
        o.v[e] = u * vh.v[e] + (1.0f - u) * y;
    }
    *reinterpret_cast<fvec<E>*>(g.h_out + i) = o;
    *reinterpret_cast<fvec<E>*>(g.hsnap + i) = o;
}

__global__ __launch_bounds__(256)
void gru_combine_kernel(const GruJobs J)
{
    __shared__ float coef[2][2];
    int bid = blockIdx.x, l = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (i < J.n && bid >= J.j[i].blk0) l = i;
    const GruJob& g = J.j[l];
    bid -= g.blk0;
    const int bx = bid % g.gx, b = bid / g.gx;
    if (g.vec) gru_combine_body<4>(g, bx, b, coef);
    else       gru_combine_body<1>(g, bx, b, coef);
}

// ---- host orchestration ----------------------------------------------------------------------------------------------
// The state-independent front of a plane (cost-volume plane + the three encoder convolutions) is issued for a CHUNK of
// CH planes per launch; buffers that cross streams exist NSL = 2*CH times (two chunks: the front of chunk j+1 runs
// under the recurrent levels of chunk j).  CH depends on the geometry only, so that every caller (workspace size,
// single step, plane loop, any shard of the plane range) sees the same layout and the same kernel variants.
constexpr int CH_MAX = 8, NSL_MAX = 2 * CH_MAX;
static int red_chunk(int B, int C, int H, int W)
{
    int ch = (size_t)B * H * W <= (size_t)tune_int("SMVS_RED_CHUNK8_BELOW", 131072) ? 8 : 4;   // small planes: the host's enqueue rate binds
    const int force = tune_int("SMVS_RED_CHUNK", 0);
    if (force == 1 || force == 2 || force == 4 || force == 8) ch = force;
    while (ch > 1 && (long long)C * ch * H * W * 4 >= (1ll << 30)) ch >>= 1;                // chunk addressed with 32-bit byte offsets
    return ch;
}

struct RedWorkspace {                        // offsets in floats into the caller's workspace
    // e / hsnap exist NSL times: they cross streams, so plane k+1.. may be written while plane k is still being
    // read (see red_run_planes).  gates / rh / cand (recurrent stream) and sum (decoder stream) live on one stream.
    // e[i] + slot * e_stride[i]: the slots of a chunk are adjacent = one dense batch of CH*B samples.
    int CH, NSL;
    size_t e[3], e_stride[3], gates[4], rh[4], cand[4], hsnap[4], hsnap_stride[4], sum[3], stats[2];   // stats: doubles, offset in floats (8-byte aligned)
    // hsnap[g] + slot * hsnap_stride[g]: like e, the slots of a chunk are adjacent = one dense batch of CH*B samples for the
    // decoder, which runs once per CHUNK of planes (planes = batch dimension); sum[] holds a chunk.
    size_t gates2[4], halt[4];               // fused plane loop: the gate convolution's second buffer, the state's second buffer
    // xg / xc [g] + slot * stride: the state-INDEPENDENT half of the gate / candidate convolutions of a plane (input channels of
    // the encoder level, bias included), computed by the front for a chunk of planes; the recurrent chain continues them.
    size_t xg[4], xg_stride[4], xc[4], xc_stride[4];
    size_t total;
};

static RedWorkspace red_workspace(int B, int C, int H, int W)
{
    RedWorkspace w{};
    w.CH = red_chunk(B, C, H, W);
    w.NSL = 2 * w.CH;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
    const int hs[4] = {H, H / 2, H / 4, H / 8}, ws[4] = {W, W / 2, W / 4, W / 8};
    const int ech[3] = {16, 32, 64};
    for (int i = 0; i < 3; ++i) {
        w.e_stride[i] = (size_t)B * ech[i] * hs[i + 1] * ws[i + 1];      // multiples of 4 floats: H, W are multiples of 8
        w.e[i] = take(w.e_stride[i] * w.NSL);
    }
    for (int i = 0; i < 4; ++i) {
        w.gates[i] = take((size_t)B * 2 * HID[i] * hs[i] * ws[i]);
        w.rh[i] = take((size_t)B * HID[i] * hs[i] * ws[i]);
        w.cand[i] = take((size_t)B * HID[i] * hs[i] * ws[i]);
        w.hsnap_stride[i] = (size_t)B * HID[i] * hs[i] * ws[i];
        w.hsnap[i] = take(w.hsnap_stride[i] * w.NSL);
    }
    for (int i = 0; i < 3; ++i)              // sum[i] = relu(upconv{i+1}(.)) + state{i+1}'  (decoder skip fused into the transposed convolution)
        w.sum[i] = take((size_t)w.CH * B * HID[i] * hs[i] * ws[i]);
    for (int p = 0; p < 2; ++p)
        w.stats[p] = take((size_t)B * 4 * 3 * NSLOT * 2 * 2);   // 4 GRUs x (reset, update, output) x NSLOT x (sum, sumsq) doubles
    for (int i = 0; i < 4; ++i) {
        w.gates2[i] = take((size_t)B * 2 * HID[i] * hs[i] * ws[i]);
        w.halt[i] = take((size_t)B * HID[i] * hs[i] * ws[i]);
    }
    for (int i = 0; i < 4; ++i) {
        w.xg_stride[i] = (size_t)B * 2 * HID[i] * hs[i] * ws[i];
        w.xg[i] = take(w.xg_stride[i] * w.NSL);
        w.xc_stride[i] = (size_t)B * HID[i] * hs[i] * ws[i];
        w.xc[i] = take(w.xc_stride[i] * w.NSL);
    }
    w.total = o;
    return w;
}

static bool g_red_direct_only()
{
    return tune_int("SMVS_CONV_DIRECT", 0) == 1;            // A/B switch (tuning builds): direct kernels only
}

#ifndef SMVS_MFMA_UNIT_WAVES
#define SMVS_MFMA_UNIT_WAVES 1               // level-batched RED launches: unit-per-wave MFMA jobs in the throughput regime (0: round 2, A/B)
#endif
static int g_mfma_units(int which)
{
    static const int v1 = tune_int("SMVS_MFMA_UNITS_KS1", 1024), v2 = tune_int("SMVS_MFMA_UNITS_KS2", 512);
    return which == 0 ? v1 : v2;                            // (tile, cout tile) units of a layer from which its MFMA job runs one / two waves per unit
}

static int g_split_below()
{
    static const int v = tune_int("SMVS_CONV_SPLIT_BELOW", 512);
    return v;                                               // workgroups (unsplit) below which the channel-split kernels run
                                                            // (1024 -> 512: the 384x192 level-1 gate convolution, 576 workgroups, is
                                                            //  24 -> ~16 us unsplit; stage 2 of the cascade 3.13 -> 3.01 ms)
}

static int g_rows_from()
{
    static const int v = tune_int("SMVS_CONV_ROWS4_FROM", 1024);
    return v;                                               // workgroups (nine-tap kernel, one sample) from which a lane computes 4 output rows
}

static MfmaConvArgs mfma_args(int stride, const ConvArgs& a, const float* wm)
{
    MfmaConvArgs m{};
    m.inA = a.inA; m.CA = a.CA; m.inB = a.inB; m.CB = a.CB; m.scaleA = a.scaleA; m.w = wm; m.bias = a.bias;
    m.out = a.out; m.stats = a.stats; m.ngroups = a.ngroups; m.nslot = NSLOT;
    m.Cout = a.Cout; m.relu = a.relu; m.stride = stride;
    m.Di = m.Do = 1; m.Hi = a.Hi; m.Wi = a.Wi; m.Ho = a.Ho; m.Wo = a.Wo;
    m.init = a.init; m.wcip0 = a.wc0 / 2; m.wcipN = a.CinW / 2;
    return m;
}

// `wm` = MFMA-order weights of the same layer (used when the layer qualifies; the MFMA kernel reads dense inputs)
// Bh = the batch size the kernel VARIANT is chosen for (0 = B): the last chunk of a plane range may be short, and a
// plane's bits must not depend on how the range was chunked or sharded.
static void launch_conv(int stride, const ConvArgs& a, int B, hipStream_t st, const float* wm = nullptr, int Bh = 0)
{
    if (Bh <= 0) Bh = B;
    if (wm && !a.inA_cs && mfma_conv_ok(a.CA, a.CB, a.Cout) && !g_red_direct_only()) {
        mfma_conv_launch<9>(mfma_args(stride, a, wm), B, st, Bh);
        return;
    }
    const int ncog = (a.Cout + COT - 1) / COT;
    const int wx = (a.Wo + 63) / 64;
    if (wx * ((a.Ho + 3) / 4) * Bh * ncog < g_split_below() && (!SMVS_WLDS || a.CA + a.CB <= WLDS_MAX_CIN)) {   // coarse plane: latency regime
        dim3 grd(wx, a.Ho, B * ncog), blk(256);
        if (stride == 1) hipLaunchKernelGGL((conv3x3_kernel<1, true>), grd, blk, 0, st, a);
        else             hipLaunchKernelGGL((conv3x3_kernel<2, true>), grd, blk, 0, st, a);
        return;
    }
    dim3 grd(wx, (a.Ho + 3) / 4, B * ncog), blk(256);
    if (stride == 1 && wx * ((a.Ho + 3) / 4) * ncog >= g_rows_from())          // throughput regime (one sample's geometry): 4 output rows per lane
        hipLaunchKernelGGL((conv3x3_rows_kernel<4>), dim3(wx, (a.Ho + 15) / 16, B * ncog), blk, 0, st, a);
    else if (stride == 1) hipLaunchKernelGGL((conv3x3_kernel<1, false>), grd, blk, 0, st, a);
    else             hipLaunchKernelGGL((conv3x3_kernel<2, false>), grd, blk, 0, st, a);
}

// one stride-1 convolution as a job of a level-batched launch; returns its workgroup count
static int conv_job(ConvJob& j, const ConvArgs& a, int B, const float* wm, int blk0, int Bh = 0)
{
    if (Bh <= 0) Bh = B;                      // the batch the kernel VARIANT is chosen for (see launch_conv)
    j.blk0 = blk0;
    if (wm && !a.inA_cs && mfma_conv_ok(a.CA, a.CB, a.Cout) && !g_red_direct_only()) {
        j.kind = 2; j.m = mfma_args(1, a, wm);
        j.gx = ((a.Wo + 31) / 32) * a.Ho * B; j.gy = a.Cout / 32;
        j.m.ntiles = j.gx;
        // Many (tile, cout tile) units -- the levels of the full-resolution stage: a unit per wave (or per two waves) instead of
        // per workgroup, so that a wave's K loop is long enough to stream (mfma_conv.h).  Chosen from the geometry of ONE
        // sample: a plane's bits do not depend on batch, chunking or sharding.
        const int units1 = ((a.Wo + 31) / 32) * a.Ho * j.gy;
        if (SMVS_MFMA_UNIT_WAVES && units1 >= g_mfma_units(0)) { j.kind = 4; j.gx = (j.gx * j.gy + 3) / 4; return j.gx; }
        if (SMVS_MFMA_UNIT_WAVES && units1 >= g_mfma_units(1)) { j.kind = 3; j.gx = (j.gx * j.gy + 1) / 2; return j.gx; }
        return j.gx * j.gy;
    }
    j.a = a;
    const int ncog = (a.Cout + COT - 1) / COT;
    j.gx = (a.Wo + 63) / 64;
    const bool split = j.gx * ((a.Ho + 3) / 4) * Bh * ncog < g_split_below() && (!SMVS_WLDS || a.CA + a.CB <= WLDS_MAX_CIN);
    j.kind = split ? 1 : 0;
    j.gy = split ? a.Ho : (a.Ho + 3) / 4;
    // throughput regime (chosen from ONE sample's geometry): 4 / 2 output rows per lane
    static const int rows2 = tune_int("SMVS_CONV_ROWS2_FROM", 1 << 30);      // (2 rows per lane: no gain measured at cascade stage 2; tuning builds)
    const int wg1 = j.gx * ((a.Ho + 3) / 4) * ncog;
    if (!split && wg1 >= g_rows_from()) { j.kind = 6; j.gy = (a.Ho + 15) / 16; }
    else if (!split && wg1 >= rows2) { j.kind = 5; j.gy = (a.Ho + 7) / 8; }
    return j.gx * j.gy * B * ncog;
}

// Bh: the batch the kernel VARIANT is chosen for (0 = B), see launch_conv
static void launch_convT(const ConvArgs& a, int B, hipStream_t st, int Bh = 0)
{
    if (Bh <= 0) Bh = B;
    const int ncog = (a.Cout + COT - 1) / COT;
    const int wx = (a.Wi + 63) / 64;
    if (wx * ((a.Hi + 3) / 4) * Bh * ncog < g_split_below()) {
        hipLaunchKernelGGL(convT3x3s2_kernel<true>, dim3(wx, a.Hi, B * ncog), dim3(256), 0, st, a);
        return;
    }
    hipLaunchKernelGGL(convT3x3s2_kernel<false>, dim3(wx, (a.Hi + 3) / 4, B * ncog), dim3(256), 0, st, a);
}

// ---- plane pipeline ------------------------------------------------------------------------------------------
// Data flow of a plane:
//   * the front (cost-volume plane, encoder) does not depend on the recurrent state: caller's stream, issued for a
//     whole CHUNK of planes per launch (planes = batch dimension);
//   * the four ConvGRU levels only need their encoder level and their own previous state: four level-batched
//     launches per plane on the RECURRENT stream (this chain is what bounds the loop);
//   * the decoder (transposed convolutions with the skip add fused, output convolution, regression update) reads
//     snapshots of the new states: DECODER stream, running under the next plane's recurrent launches.
// Events: encoder done (per chunk), states done (per plane), plane done (per plane; the front of chunk j waits for
// the last plane of chunk j-2, whose ring entries it reuses).  A single plane runs on the caller's stream alone.
int resolve_heights(const float* depth, int depth_is_4d, const smvs_height_gen* gen, int D, int H, int W,
                    int& mode, HeightGen& hg);                                              // regress.hip

constexpr int RING = 2 * NSL_MAX;
struct RedPipe {
    hipStream_t rec, dec;
    hipEvent_t enc[RING], state[RING], done[RING];
};

static RedPipe* red_pipe_create()
{
    RedPipe* p = new RedPipe();
    // the recurrent chain bounds the loop: its stream gets the highest queue priority, so that its workgroups are dispatched
    // ahead of the front's and the decoder's wherever they compete for CUs
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    const int pr = tune_int("SMVS_RED_PRIO", 1);
    bool ok = (pr ? hipStreamCreateWithPriority(&p->rec, hipStreamNonBlocking, hi) : hipStreamCreateWithFlags(&p->rec, hipStreamNonBlocking)) == hipSuccess
           && (pr == 2 ? hipStreamCreateWithPriority(&p->dec, hipStreamNonBlocking, lo) : hipStreamCreateWithFlags(&p->dec, hipStreamNonBlocking)) == hipSuccess;
    for (int r = 0; r < RING && ok; ++r)
        ok = hipEventCreateWithFlags(&p->enc[r], hipEventDisableTiming) == hipSuccess
          && hipEventCreateWithFlags(&p->state[r], hipEventDisableTiming) == hipSuccess
          && hipEventCreateWithFlags(&p->done[r], hipEventDisableTiming) == hipSuccess;

    if (!ok) { delete p; return nullptr; }        // (the few objects created before the failure are abandoned: the device is unusable anyway)
    return p;
}

// The helper streams / events of the plane pipeline come from a per-device pool guarded by a mutex: a call borrows one
// set for its duration and returns it, so the number of sets ever created is the peak number of CONCURRENT calls on a
// device (one per nn.DataParallel replica thread), not the number of threads that ever called -- DataParallel starts
// fresh threads for every forward, which made a thread_local cache leak its streams and events on every forward.
struct RedPipeLease {
    RedPipe* p = nullptr; int dev = -1;
    static std::mutex& mu() { static std::mutex m; return m; }
    static std::vector<RedPipe*>& pool(int dev) { static std::vector<RedPipe*> v[64]; return v[dev]; }
    RedPipeLease()
    {
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { dev = -1; return; }
        {
            std::lock_guard<std::mutex> g(mu());
            auto& v = pool(dev);
            if (!v.empty()) { p = v.back(); v.pop_back(); }
        }
        if (!p) p = red_pipe_create();
    }
    ~RedPipeLease()
    {
        if (!p) return;
        std::lock_guard<std::mutex> g(mu());
        pool(dev).push_back(p);                     // work enqueued on its streams is ordered behind the caller's stream by the final join
    }
    RedPipeLease(const RedPipeLease&) = delete;
    RedPipeLease& operator=(const RedPipeLease&) = delete;
    static void release_all()                   // smvs_shutdown(): destroy every pooled set (none may be on lease)
    {
        std::lock_guard<std::mutex> g(mu());
        for (int d = 0; d < 64; ++d) {
            for (RedPipe* q : pool(d)) {
                (void)hipStreamDestroy(q->rec); (void)hipStreamDestroy(q->dec);
                for (int r = 0; r < RING; ++r) { (void)hipEventDestroy(q->enc[r]); (void)hipEventDestroy(q->state[r]); (void)hipEventDestroy(q->done[r]); }
                delete q;
            }
            pool(d).clear();
        }
    }
};

static std::atomic<int> g_red_streams{2};     // smvs_red_set_streams(): 0 = caller's stream only (capture-safe)

struct RedRun {
    const float* packed; float* state[4]; float* wsf; int B, C, H, W; hipStream_t main;
    // single step: the caller's variance plane and output plane
    const float* cost; float* reg_out;
    // pred loop (pred = true): cost-volume plane built here, regression accumulators updated here
    bool pred; int geo_kind; const float* ref_fea; const float* const* src_fea; int n_src; const double* geo;
    const float* depth; int depth_is_4d; const smvs_height_gen* gen; double* acc; int D;
    smvs_height_gen gen_own;            // the caller's generator with this pipeline's arithmetic default filled in (see red_planes_entry)
    float* reg_volume;                       // pred with acc == null: regularised planes go to (B,D,H,W) instead
    float* block[2];                         // two (B,C,CH,H,W) chunks of variance planes
};

// Everything one call needs to enqueue planes: issue_front() = caller's stream (cost volume, encoder of a chunk);
// issue_recurrent() = the four level-batched launches of a plane; issue_decoder() = its decoder and regression update.
struct RedIssuer {
    const RedRun& r; const RedPipe& P; bool multi; int d_begin, d_end;
    RedLayout L; RedWorkspace ws;
    hipStream_t sR, sD;
    int hs[4], wd[4];
    int hmode = 0; HeightGen hg{};            // heights of the regression update, resolved once per call

    RedIssuer(const RedRun& run, const RedPipe& pipe, bool multi_, int d0, int d1)
        : r(run), P(pipe), multi(multi_), d_begin(d0), d_end(d1), L(red_layout(run.C)), ws(red_workspace(run.B, run.C, run.H, run.W))
    {
        sR = multi ? P.rec : r.main;
        sD = multi ? P.dec : r.main;
        for (int g = 0; g < 4; ++g) { hs[g] = r.H >> g; wd[g] = r.W >> g; }
    }
    // Fused plane loop (2 dependent launches per plane, the element-wise stages computed by the consuming convolutions into
    // LDS: mfma_conv.h, FuseB).  Built and measured in round 4 -- correct (every RED test passes with it on) and SLOWER:
    // 80 / 152 us per plane at cascade stages 1 / 2 against 62 / 95 for the four-launch chain (profiles/r04_red_fusion.txt).
    // The premise was wrong: the two element-wise launches cost 5-8 us of kernel + ~3 us of gap each, not 13 us; computing
    // their stage per tile (3-12x redundantly: halos, one tile per output-channel group) inside the latency chain of the
    // convolutions costs 10-33 us per launch, and what bounded the plane was the decoder chain on the other stream.  Kept
    // for tuning builds (SMVS_RED_FUSED=1); the shipped library never takes this path.
    bool fused() const { return tune_int("SMVS_RED_FUSED", 0) == 1; }
    float* state_buf(int g, int k) const { return (k & 1) ? r.wsf + ws.halt[g] : r.state[g]; }     // state entering plane k of this call
    float* gates_buf(int g, int k) const { return r.wsf + ((k & 1) ? ws.gates2[g] : ws.gates[g]); }
    int CH() const { return ws.CH; }
    double* stats_of(int k) const { return (double*)(r.wsf + ws.stats[k & 1]); }
    float* e_of(int k, int i) const { return r.wsf + ws.e[i] + (size_t)(k % ws.NSL) * ws.e_stride[i]; }
    float* hsnap_of(int k, int g) const { return r.wsf + ws.hsnap[g] + (size_t)(k % ws.NSL) * ws.hsnap_stride[g]; }
    float* xg_of(int k, int g) const { return r.wsf + ws.xg[g] + (size_t)(k % ws.NSL) * ws.xg_stride[g]; }
    float* xc_of(int k, int g) const { return r.wsf + ws.xc[g] + (size_t)(k % ws.NSL) * ws.xc_stride[g]; }
    // The ConvGRU convolutions act on cat(x, h) / cat(x, r*h): their x half does not depend on the recurrent state.  With the
    // split on, the front computes it (+ bias) for a whole chunk of planes (planes = batch, caller's stream, under the previous
    // chunk's chain) and the chain's two convolutions per plane walk the hidden channels only (8 of 40 input channels at level 1
    // of cascade stage 1, half of them at the coarser levels), starting from those sums.  Built and measured in round 4 (every
    // RED test passes with it on; profiles/r04_red_xsplit.txt): the chain's convolutions get 18.3 -> 13.0 / 15.4 -> 11.8 us
    // shorter at stage 1, but a convolution over half the channels costs well over half the time (tile set-up, epilogue, the extra
    // store + load of the partial sums) and the x halves compete with the chain for the CUs: 57 -> 54 us per plane at stage 1,
    // 98 -> 100 at stage 2, 262 -> 288 at stage 3.  Off; tuning builds: SMVS_RED_XSPLIT=1.
    bool xsplit() const { return !fused() && tune_int("SMVS_RED_XSPLIT", 0) == 1; }
    // Per-level chains (one stream per ConvGRU level, every level walking its own 4 launches per plane, no event between levels):
    // built and measured in round 6, REJECTED -- 175 / 198 / 377 us per plane at cascade stages 1 / 2 / 3 against 55 / 92 / 226 for
    // the level-batched chain, 19.2-20.2 ms per cascade forward against 7.6-7.7 (profiles/r06_red_per_level.txt).  Six concurrent
    // streams of dependent small launches cost the runtime far more per launch than the levels' different lengths cost the batched
    // launch; the code is gone.

    // the variance plane of plane k as a strided view for the convolutions that read it
    void cost_view(int k, ConvArgs& a) const
    {
        if (!r.pred) { a.inA = r.cost; return; }                                  // single step: the caller's dense plane
        const size_t HW = (size_t)r.H * r.W;
        a.inA = r.block[(k / ws.CH) & 1] + (size_t)(k % ws.CH) * HW;
        a.inA_cs = (size_t)ws.CH * HW; a.inA_bs = (size_t)r.C * ws.CH * HW;
    }

    // cost-volume planes + encoder of planes [k0, k0+n) on the caller's stream (n <= CH, k0 a multiple of CH)
    int issue_front(int k0, int n)
    {
        const int d = d_begin + k0;
        const int B = r.B, C = r.C, ch = ws.CH;
        const int enc_in[3] = {C, 16, 32}, enc_out[3] = {16, 32, 64};
        if (r.pred) {
            float* blk = r.block[(k0 / ch) & 1];
            const int rc = r.gen
                ? (r.geo_kind == 0
                   ? smvs_rpc_costvol_fwd_gen(r.ref_fea, r.src_fea, r.n_src, r.geo, r.gen, blk, B, C, r.D, r.H, r.W, d, d + n, ch, 0, r.main)
                   : smvs_homo_costvol_fwd_gen(r.ref_fea, r.src_fea, r.n_src, r.geo, r.gen, blk, B, C, r.D, r.H, r.W, d, d + n, ch, 0, r.main))
                : (r.geo_kind == 0
                   ? smvs_rpc_costvol_fwd(r.ref_fea, r.src_fea, r.n_src, r.geo, r.depth, r.depth_is_4d, blk, B, C, r.D, r.H, r.W, d, d + n, ch, 0, r.main)
                   : smvs_homo_costvol_fwd(r.ref_fea, r.src_fea, r.n_src, r.geo, r.depth, r.depth_is_4d, blk, B, C, r.D, r.H, r.W, d, d + n, ch, 0, r.main));
            if (rc) return rc;
        }
        // encoder: e1 = relu(conv1(-cost)), e2 = relu(conv2(e1)), e3 = relu(conv3(e2)); samples = (plane, batch)
        for (int i = 0; i < 3; ++i) {
            ConvArgs a{};
            if (i == 0) { cost_view(k0, a); if (r.pred) { a.inA_ps = (size_t)r.H * r.W; a.inA_bmod = B; } }
            else a.inA = e_of(k0, i - 1);
            a.CA = enc_in[i]; a.scaleA = i == 0 ? -1.0f : 1.0f;
            a.w = r.packed + L.conv_w[i]; a.out = e_of(k0, i); a.Cout = enc_out[i];
            a.Hi = hs[i]; a.Wi = wd[i]; a.Ho = hs[i + 1]; a.Wo = wd[i + 1]; a.relu = 1;
            launch_conv(2, a, n * B, r.main, r.packed + L.conv_wm[i], r.pred ? ch * B : B);
        }
        if (xsplit()) {
            ConvJobs jg{}, jc{};
            jg.n = jc.n = 4;
            int nbg = 0, nbc = 0;
            const int Bh = r.pred ? ch * B : B;
            for (int q = 0; q < 4; ++q) {
                const int g = 3 - q, hc = HID[g];
                const int cx = g == 0 ? C : enc_out[g - 1];
                for (int pass = 0; pass < 2; ++pass) {
                    ConvArgs a{};
                    if (g == 0) { cost_view(k0, a); if (r.pred) { a.inA_ps = (size_t)r.H * r.W; a.inA_bmod = B; } }
                    else a.inA = e_of(k0, g - 1);
                    a.CA = cx; a.scaleA = g == 0 ? -1.0f : 1.0f; a.CinW = cx + hc;
                    a.w = r.packed + (pass ? L.out_w[g] : L.gate_w[g]); a.bias = r.packed + (pass ? L.out_b[g] : L.gate_b[g]);
                    a.out = pass ? xc_of(k0, g) : xg_of(k0, g);
                    a.Cout = pass ? hc : 2 * hc; a.Hi = a.Ho = hs[g]; a.Wi = a.Wo = wd[g];
                    if (pass) nbc += conv_job(jc.j[q], a, n * B, r.packed + L.out_wm[g], nbc, Bh);
                    else      nbg += conv_job(jg.j[q], a, n * B, r.packed + L.gate_wm[g], nbg, Bh);
                }
            }
            hipLaunchKernelGGL(conv_jobs_kernel, dim3(nbg), dim3(256), 0, r.main, jg);
            hipLaunchKernelGGL(conv_jobs_kernel, dim3(nbc), dim3(256), 0, r.main, jc);
        }
        if (multi) (void)hipEventRecord(P.enc[k0 % RING], r.main);
        return SMVS_OK;
    }

    // gates, gate apply, candidates, combine of all four levels: one launch each
    int issue_recurrent(int k)
    {
        const int d = d_begin + k;
        const int B = r.B, C = r.C;
        float* wsf = r.wsf;
        const float* packed = r.packed;
        const int enc_out[3] = {16, 32, 64};
        double* stats = stats_of(k);
        double* stats_next = d + 1 < d_end ? stats_of(k + 1) : nullptr;
        if (multi && k % ws.CH == 0) (void)hipStreamWaitEvent(sR, P.enc[k % RING], 0);   // later planes of the chunk follow on the same stream
        ConvJobs gate{}, cand{};
        GruJobs gru{};
        gate.n = cand.n = gru.n = 4; gru.B = B;
        int nb_gate = 0, nb_cand = 0, nb_gru = 0;
        for (int q = 0; q < 4; ++q) {
            const int g = 3 - q;                                          // coarse levels first: longest channel loops
            const int hc = HID[g], hw = hs[g] * wd[g];
            const int cx = g == 0 ? C : enc_out[g - 1];
            const float sx = g == 0 ? -1.0f : 1.0f;
            double* sg = stats + (size_t)g * B * 3 * NSLOT * 2;           // [b][reset,update][slot][2], then [b][slot][2] for the output norm
            double* so = sg + (size_t)B * 2 * NSLOT * 2;
            const bool xs = xsplit();
            ConvArgs a{};
            if (xs) { a.inA = r.state[g]; a.CA = hc; a.scaleA = 1.0f; a.wc0 = cx; a.CinW = cx + hc; a.init = xg_of(k, g); }
            else {
                if (g == 0) cost_view(k, a); else a.inA = e_of(k, g - 1);
                a.CA = cx; a.scaleA = sx; a.inB = r.state[g]; a.CB = hc; a.bias = packed + L.gate_b[g];
            }
            a.w = packed + L.gate_w[g]; a.out = wsf + ws.gates[g]; a.stats = sg; a.ngroups = 2;
            a.Cout = 2 * hc; a.Hi = a.Ho = hs[g]; a.Wi = a.Wo = wd[g];
            nb_gate += conv_job(gate.j[q], a, B, packed + L.gate_wm[g], nb_gate);
            ConvArgs o{};
            if (xs) { o.inA = wsf + ws.rh[g]; o.CA = hc; o.scaleA = 1.0f; o.wc0 = cx; o.CinW = cx + hc; o.init = xc_of(k, g); }
            else {
                if (g == 0) cost_view(k, o); else o.inA = e_of(k, g - 1);
                o.CA = cx; o.scaleA = sx; o.inB = wsf + ws.rh[g]; o.CB = hc; o.bias = packed + L.out_b[g];
            }
            o.w = packed + L.out_w[g]; o.out = wsf + ws.cand[g]; o.stats = so; o.ngroups = 1;
            o.Cout = hc; o.Hi = o.Ho = hs[g]; o.Wi = o.Wo = wd[g];
            nb_cand += conv_job(cand.j[q], o, B, packed + L.out_wm[g], nb_cand);
            GruJob& u = gru.j[q];
            u.gates = wsf + ws.gates[g]; u.stats_g = sg; u.stats_o = so;
            u.rn_w = packed + L.rn_w[g]; u.rn_b = packed + L.rn_b[g]; u.un_w = packed + L.un_w[g]; u.un_b = packed + L.un_b[g];
            u.on_w = packed + L.on_w[g]; u.on_b = packed + L.on_b[g];
            u.h = r.state[g]; u.h_out = r.state[g]; u.rh = wsf + ws.rh[g]; u.cand = wsf + ws.cand[g]; u.hsnap = hsnap_of(k, g);
            u.zero_next = stats_next ? stats_next + (size_t)g * B * 3 * NSLOT * 2 : nullptr;
            u.vec = tune_int("SMVS_GRU_VEC", 1) == 1 && hw % 4 == 0 && ((uintptr_t)u.h | (uintptr_t)u.h_out | (uintptr_t)u.hsnap | (uintptr_t)u.gates | (uintptr_t)u.rh | (uintptr_t)u.cand) % 16 == 0;
            u.HC = hc; u.HW = hw; u.gx = (int)(((size_t)hc * hw / (u.vec ? 4 : 1) + 255) / 256); u.blk0 = nb_gru;
            nb_gru += u.gx * B;
        }
        if (tune_int("SMVS_RED_SPLIT_JOBS", 0)) {                       // tuning builds: one launch per level, to time the jobs
            for (int pass = 0; pass < 2; ++pass) {
                const ConvJobs& all = pass ? cand : gate;
                for (int q = 0; q < 4; ++q) {
                    ConvJobs one{};
                    one.n = 1; one.j[0] = all.j[q]; one.j[0].blk0 = 0;
                    const int nb = (q < 3 ? all.j[q + 1].blk0 : (pass ? nb_cand : nb_gate)) - all.j[q].blk0;
                    hipLaunchKernelGGL(conv_jobs_kernel, dim3(nb), dim3(256), 0, sR, one);
                }
                if (!pass) hipLaunchKernelGGL(gru_gate_apply_kernel, dim3(nb_gru), dim3(256), 0, sR, gru);
            }
        } else {
        hipLaunchKernelGGL(conv_jobs_kernel, dim3(nb_gate), dim3(256), 0, sR, gate);
        hipLaunchKernelGGL(gru_gate_apply_kernel, dim3(nb_gru), dim3(256), 0, sR, gru);
        hipLaunchKernelGGL(conv_jobs_kernel, dim3(nb_cand), dim3(256), 0, sR, cand);
        }
        hipLaunchKernelGGL(gru_combine_kernel, dim3(nb_gru), dim3(256), 0, sR, gru);
        // the decoder runs per chunk: only the chunk's last plane signals it (an event per plane costs the chain ~1 us each)
        if (multi && ((k + 1) % ws.CH == 0 || d + 1 == d_end)) (void)hipEventRecord(P.state[k % RING], sR);
        return SMVS_OK;
    }

    // Fused chain of plane k: L_A = [combine of plane k-1 on load] + gate convolutions, L_B = [gate apply on load] + candidate
    // convolutions.  The state entering plane k lives in state_buf(., k) (the caller's tensors for even k, the workspace for odd
    // k): L_A(k) computes it from plane k-1's raw gates / candidates and stores it there (and into the decoder's snapshot of plane
    // k-1); gate buffers alternate likewise (L_A(k) reads plane k-1's update gate while it writes plane k's gates).
    // After the last plane issue_final_combine() produces the final state in the caller's tensors.
    int issue_recurrent_fused(int k)
    {
        const int d = d_begin + k;
        const int B = r.B, C = r.C;
        float* wsf = r.wsf;
        const float* packed = r.packed;
        const int enc_out[3] = {16, 32, 64};
        double* stats = stats_of(k);
        double* stats_prev = stats_of(k + 1);                                 // ring of two: plane k-1's = plane k+1's
        const bool has_next = d + 1 < d_end;
        if (multi && k % ws.CH == 0) (void)hipStreamWaitEvent(sR, P.enc[k % RING], 0);
        ConvJobs gate{}, cand{};
        gate.n = cand.n = 4;
        int nb_gate = 0, nb_cand = 0;
        for (int q = 0; q < 4; ++q) {
            const int g = 3 - q;
            const int hc = HID[g];
            const int cx = g == 0 ? C : enc_out[g - 1];
            const float sx = g == 0 ? -1.0f : 1.0f;
            const size_t lvl = (size_t)g * B * 3 * NSLOT * 2;
            double* sg = stats + lvl;
            double* so = sg + (size_t)B * 2 * NSLOT * 2;
            FuseB fa{};                                                        // gate convolution: B = h(k) = combine of plane k-1
            fa.HC = hc; fa.nslot = NSLOT;
            if (k > 0) {
                fa.mode = FUSE_COMBINE;
                fa.gates = gates_buf(g, k - 1); fa.cand = wsf + ws.cand[g]; fa.h = state_buf(g, k - 1);
                fa.h_out = state_buf(g, k); fa.hsnap = hsnap_of(k - 1, g);
                fa.stats_g = stats_prev + lvl; fa.stats_o = stats_prev + lvl + (size_t)B * 2 * NSLOT * 2;
                fa.gw = packed + L.un_w[g]; fa.gb = packed + L.un_b[g]; fa.ow = packed + L.on_w[g]; fa.ob = packed + L.on_b[g];
            }
            ConvArgs a{};
            if (g == 0) cost_view(k, a); else a.inA = e_of(k, g - 1);
            a.CA = cx; a.scaleA = sx; a.inB = state_buf(g, k); a.CB = hc;
            a.w = packed + L.gate_w[g]; a.bias = packed + L.gate_b[g]; a.out = gates_buf(g, k); a.stats = sg; a.ngroups = 2;
            a.Cout = 2 * hc; a.Hi = a.Ho = hs[g]; a.Wi = a.Wo = wd[g];
            nb_gate += conv_job(gate.j[q], a, B, packed + L.gate_wm[g], nb_gate);
            gate.j[q].fuse = fa; gate.j[q].m.fuse = fa;
            FuseB fb{};                                                        // candidate convolution: B = r * h(k)
            fb.mode = FUSE_APPLY; fb.HC = hc; fb.nslot = NSLOT;
            fb.gates = gates_buf(g, k); fb.h = state_buf(g, k); fb.stats_g = sg;
            fb.gw = packed + L.rn_w[g]; fb.gb = packed + L.rn_b[g];
            // plane k+1's statistics (= plane k-1's buffer: its last reader, L_A(k), is behind us) are cleared by this launch
            if (has_next && q == 0) { fb.zero = stats_prev; fb.zero_n = B * 4 * 3 * NSLOT * 2; }
            ConvArgs o{};
            if (g == 0) cost_view(k, o); else o.inA = e_of(k, g - 1);
            o.CA = cx; o.scaleA = sx; o.inB = state_buf(g, k); o.CB = hc;
            o.w = packed + L.out_w[g]; o.bias = packed + L.out_b[g]; o.out = wsf + ws.cand[g]; o.stats = so; o.ngroups = 1;
            o.Cout = hc; o.Hi = o.Ho = hs[g]; o.Wi = o.Wo = wd[g];
            nb_cand += conv_job(cand.j[q], o, B, packed + L.out_wm[g], nb_cand);
            cand.j[q].fuse = fb; cand.j[q].m.fuse = fb;
        }
        if (k > 0) hipLaunchKernelGGL(conv_jobs_fused_kernel, dim3(nb_gate), dim3(256), 0, sR, gate);
        else       hipLaunchKernelGGL(conv_jobs_kernel, dim3(nb_gate), dim3(256), 0, sR, gate);       // plane 0 of the call: the state is the caller's
        if (multi && k > 0) (void)hipEventRecord(P.state[(k - 1) % RING], sR);                         // plane k-1's snapshots are complete
        hipLaunchKernelGGL(conv_jobs_fused_kernel, dim3(nb_cand), dim3(256), 0, sR, cand);
        return SMVS_OK;
    }

    // the combine stage of the call's last plane: final state into the caller's tensors, snapshots for its decoder
    int issue_final_combine(int k)
    {
        const int B = r.B;
        float* wsf = r.wsf;
        const float* packed = r.packed;
        double* stats = stats_of(k);
        GruJobs gru{};
        gru.n = 4; gru.B = B;
        int nb = 0;
        for (int q = 0; q < 4; ++q) {
            const int g = 3 - q;
            const int hc = HID[g], hw = hs[g] * wd[g];
            GruJob& u = gru.j[q];
            double* sg = stats + (size_t)g * B * 3 * NSLOT * 2;
            u.gates = gates_buf(g, k); u.stats_g = sg; u.stats_o = sg + (size_t)B * 2 * NSLOT * 2;
            u.un_w = packed + L.un_w[g]; u.un_b = packed + L.un_b[g]; u.on_w = packed + L.on_w[g]; u.on_b = packed + L.on_b[g];
            u.h = state_buf(g, k); u.h_out = r.state[g]; u.cand = wsf + ws.cand[g]; u.hsnap = hsnap_of(k, g);
            u.HC = hc; u.HW = hw; u.gx = (int)(((size_t)hc * hw + 255) / 256); u.blk0 = nb;
            nb += u.gx * B;
        }
        hipLaunchKernelGGL(gru_combine_kernel, dim3(nb), dim3(256), 0, sR, gru);
        if (multi) (void)hipEventRecord(P.state[k % RING], sR);
        return SMVS_OK;
    }

    // can every job of the fused launches hold its B tile in the kernel's LDS?  (geometry only)
    bool fused_fits() const
    {
        const int enc_out[3] = {16, 32, 64};
        for (int g = 0; g < 4; ++g) {
            const int hc = HID[g], cx = g == 0 ? r.C : enc_out[g - 1];
            for (int pass = 0; pass < 2; ++pass) {
                ConvArgs a{};
                ConvJob j{};
                a.CA = cx; a.CB = hc; a.Cout = pass ? hc : 2 * hc; a.Hi = a.Ho = hs[g]; a.Wi = a.Wo = wd[g];
                static const float dummy = 0.0f;
                (void)conv_job(j, a, r.B, &dummy, 0);
                if (j.kind >= 4 || fuse_tile_floats(j.kind, hc) > FUSE_TILE_FLOATS) return false;
            }
        }
        return true;
    }

    // Decoder of planes [k0, k0+n) (one chunk; k0 a multiple of CH) from their state snapshots, output convolution, regression
    // update: the planes are the batch dimension of three transposed convolutions and one output-convolution launch -- 4
    // launches per CHUNK.  (Per plane, as until round 3, the decoder was a second chain of 4 dependent launches beside the
    // recurrent one and cost it ~10 us of interference per plane; nothing in it depends on the neighbouring planes.)  The kernel
    // variants are chosen for a full chunk, so a plane's bits do not depend on the range or its chunking.
    int issue_decoder(int k0, int n)
    {
        const int d = d_begin + k0;
        const int B = r.B, ch = ws.CH;
        float* wsf = r.wsf;
        const float* packed = r.packed;
        const size_t npix = (size_t)B * r.H * r.W;
        if (multi) (void)hipStreamWaitEvent(sD, P.state[(k0 + n - 1) % RING], 0);
        for (int g = 3; g >= 1; --g) {
            // sum[g-1] = relu(upconv{g}(state4' or sum[g])) + state{g}'
            ConvArgs u{};
            u.inA = g == 3 ? hsnap_of(k0, 3) : wsf + ws.sum[g]; u.CA = HID[g]; u.scaleA = 1.0f;
            u.w = packed + L.up_w[g - 1]; u.out = wsf + ws.sum[g - 1]; u.skip = hsnap_of(k0, g - 1); u.Cout = HID[g - 1];
            u.Hi = hs[g]; u.Wi = wd[g]; u.Ho = hs[g - 1]; u.Wo = wd[g - 1]; u.relu = 1;
            launch_convT(u, n * B, sD, ch * B);
        }
        // reg = upconv2d(up1 + state1') : ConvTranspose2d stride 1 == correlation with flipped taps; the regression
        // update of the pred loop rides in its epilogue, folded over the chunk's planes in plane order
        OutConvArgs f{};
        f.in = wsf + ws.sum[0]; f.NP = n; f.w = packed + L.up2d_w; f.bias = packed + L.up2d_b;
        f.B = B; f.H = r.H; f.W = r.W; f.d = d;
        if (!r.pred) { f.reg = r.reg_out; f.reg_bstride = (size_t)r.H * r.W; f.reg_pstride = 0; }
        else if (r.reg_volume) { f.reg = r.reg_volume + (size_t)d * r.H * r.W; f.reg_bstride = (size_t)r.D * r.H * r.W; f.reg_pstride = (size_t)r.H * r.W; }
        else {
            f.exp_sum = r.acc; f.depth_img = r.acc + npix; f.max_prob = r.acc + 2 * npix;
            f.depth = r.depth; f.hmode = hmode; f.hg = hg; f.D = r.D;
        }
        hipLaunchKernelGGL(out_conv_regress_kernel, dim3((r.W + 63) / 64, (r.H + 3) / 4, B), dim3(256), 0, sD, f);
        if (multi) (void)hipEventRecord(P.done[(k0 + n - 1) % RING], sD);
        return SMVS_OK;
    }
};

static int red_run_planes(const RedRun& r, int d_begin, int d_end)
{
    RedPipeLease lease;
    RedPipe* pp = lease.p;
    if (!pp) return fail(SMVS_ERR_LAUNCH, "could not create the regulariser's streams/events");
    const int nplanes = d_end - d_begin;
    if (nplanes <= 0) return SMVS_OK;
    // one plane alone: the cross-stream hops cost more than they buy (measured) -> caller's stream only
    const bool multi = nplanes > 1 && tune_int("SMVS_RED_STREAMS", 2) != 0 && g_red_streams.load(std::memory_order_relaxed) != 0;
    RedIssuer is(r, *pp, multi, d_begin, d_end);
    const RedPipe& P = *pp;
    const bool fused = is.fused() && is.fused_fits();
    if (r.pred && !r.reg_volume)
        if (int rc0 = resolve_heights(r.depth, r.depth_is_4d, r.gen, r.D, r.H, r.W, is.hmode, is.hg)) return rc0;
    // the first plane's statistics are cleared here; afterwards each level clears the next plane's buffer itself
    (void)hipMemsetAsync(r.wsf + is.ws.stats[0], 0, (size_t)r.B * 4 * 3 * NSLOT * 2 * sizeof(double), r.main);

    // The host's enqueue rate binds the small stages (tools/host_bound_probe.py); a second host thread did not help
    // (the runtime serialises), fewer calls per plane do: chunked front, level-batched launches.
    int rc = SMVS_OK;
    const int ch = is.CH();
    for (int k0 = 0; k0 < nplanes && !rc; k0 += ch) {
        const int n = nplanes - k0 < ch ? nplanes - k0 : ch;
        // this half of the ring was last used by chunk j-2; its last plane leaving the decoder implies all of it
        if (multi && k0 >= 2 * ch) (void)hipStreamWaitEvent(r.main, P.done[(k0 - ch - 1) % RING], 0);
        rc = is.issue_front(k0, n);
        for (int k = k0; k < k0 + n && !rc; ++k)
            rc = fused ? is.issue_recurrent_fused(k) : is.issue_recurrent(k);
        if (rc) break;
        if (fused) {
            // (tuning builds) plane k's snapshots come out of plane k+1's first launch: the decoder runs one chunk behind
            if (k0 > 0) rc = is.issue_decoder(k0 - ch, ch);
            if (!rc && k0 + n == nplanes) {
                rc = is.issue_final_combine(nplanes - 1);
                if (!rc) rc = is.issue_decoder(k0, n);
            }
        } else {
            rc = is.issue_decoder(k0, n);
        }
    }
    if (rc) {
        // a launch failed mid-loop: drain the helper streams before the lease goes back to the pool and the caller
        // (who will raise) frees or reuses the workspace, states and accumulators earlier planes still touch
        if (multi) { (void)hipStreamSynchronize(P.rec); (void)hipStreamSynchronize(P.dec); }
        return rc;
    }
    // join: the caller's stream continues only after the last plane's decoder (which implies the rest)
    if (multi) (void)hipStreamWaitEvent(r.main, P.done[(nplanes - 1) % RING], 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "red planes launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

}  // namespace smvs

extern "C" {

#ifdef SMVS_TIMING
// profiling builds only: copies (and clears) the MFMA body's phase counters, 8 x 8 unsigned long long
SMVS_EXPORT void smvs_debug_mfma_timing(unsigned long long* out)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(smvs::smvs_mfma_timing), sizeof(unsigned long long) * 64);
    unsigned long long z[64] = {};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(smvs::smvs_mfma_timing), z, sizeof(z));
}
#endif

SMVS_EXPORT int smvs_red_set_streams(int n) { return smvs::g_red_streams.exchange(n == 0 ? 0 : 2); }

SMVS_EXPORT int smvs_shutdown(void) { smvs::RedPipeLease::release_all(); return SMVS_OK; }

SMVS_EXPORT size_t smvs_red_packed_floats(int C) { return C > 0 ? smvs::red_layout(C).total : 0; }

SMVS_EXPORT size_t smvs_red_workspace_bytes(int B, int C, int H, int W)
{
    if (B < 1 || C < 1 || H < 8 || W < 8 || (H % 8) || (W % 8)) return 0;
    if ((long long)(C + 8) * H * W * 4 >= (1ll << 31)) return 0;       // plane beyond 32-bit offsets: unsupported
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
        hipLaunchKernelGGL(mfma_pack_kernel, dim3((n + 255) / 256), dim3(256), 0, st, src, packed + dst, cin, cout, 9, 0);
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

// ---- a single 3x3 / pad 1 layer of the regulariser as a stand-alone call (training path) --------------------------------------
// The TRAINING forward of the regulariser's layers and their input gradients (modules/module.py:_Conv3x3NativeFn) on the kernels of the
// plane loop above -- MIOpen's kernels take ~19 us per call on these shapes (2 600 calls per training step of the 48/32/8 cascade,
// profiles/r04_train_step.txt), the transposed ones go through im2col / GEMM / col2im, and the concatenation of (x, h) is a launch of
// its own.
//   smvs_conv3x3_packed_floats(cin, cout): floats of the packed weights (direct order + the MFMA order where the layer qualifies)
//   smvs_conv3x3_pack(w, packed, cin, cout, layout): the correlation from cin to cout channels whose weights are read from w as
//       layout 0: w[co][ci][ky][kx]         -- an nn.Conv2d weight (stride 1 or 2); or, for the input gradient of an nn.ConvTranspose2d
//                                              with weight (cout, cin, 3, 3): the same array
//       layout 1: w[ci][co][ky][kx], taps kept as SCATTER taps of the stride-2 transposed kernel (kind 2 below) -- an nn.ConvTranspose2d
//                                              weight; or the input gradient of a stride-2 nn.Conv2d with weight (cin, cout, 3, 3)
//       layout 2: w[ci][co][2-ky][2-kx]     -- a stride-1 nn.ConvTranspose2d as a correlation; or the input gradient of a stride-1
//                                              nn.Conv2d with weight (cin, cout, 3, 3)
//   smvs_conv3x3_fwd(kind, ...): out = [relu](layer(cat(xA (B,CA,H,W), xB (B,CB,H,W) or null)) + bias)
//       kind 0: correlation, stride 1, out (B,Cout,H,W);  kind 1: correlation, stride 2 (H, W even), out (B,Cout,H/2,W/2);
//       kind 2: transposed convolution, stride 2, output_padding 1 (weights of layout 1; xB and bias unused), out (B,Cout,2H,2W)
//       init (kinds 0 / 1; same shape as out, or null; may be `out` itself): added to the sums -- an input gradient that continues the
//       gradient contributions already collected for that tensor
SMVS_EXPORT size_t smvs_conv3x3_packed_floats(int cin, int cout)
{
    using namespace smvs;
    if (cin < 1 || cout < 1) return 0;
    return packed_conv_floats(cin, cout) + (mfma_conv_ok(cin, 0, cout) ? mfma_packed_floats(cin, cout, 9) : 0);
}

SMVS_EXPORT int smvs_conv3x3_pack(const float* w, float* packed, int cin, int cout, int layout, void* stream)
{
    using namespace smvs;
    if (!w || !packed) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (cin < 1 || cout < 1 || layout < 0 || layout > 2) return fail(SMVS_ERR_ARG, "bad channel count or layout");
    hipStream_t st = (hipStream_t)stream;
    const int n = (int)packed_conv_floats(cin, cout);
    hipLaunchKernelGGL(pack_conv_kernel, dim3((n + 255) / 256), dim3(256), 0, st, w, packed, cin, cout, layout);
    if (layout != 1 && mfma_conv_ok(cin, 0, cout)) {
        const int nm = (int)mfma_packed_floats(cin, cout, 9);
        hipLaunchKernelGGL(mfma_pack_kernel, dim3((nm + 255) / 256), dim3(256), 0, st, w, packed + n, cin, cout, 9, layout == 2 ? 1 : 0);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "conv3x3_pack launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

SMVS_EXPORT int smvs_conv3x3_fwd(int kind, const float* xA, int CA, const float* xB, int CB, const float* packed, const float* bias, const float* init,
                                 float* out, int B, int Cout, int H, int W, int relu, void* stream)
{
    using namespace smvs;
    if (!xA || !packed || !out || (CB > 0 && !xB)) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (kind < 0 || kind > 2) return fail(SMVS_ERR_ARG, "kind must be 0, 1 or 2");
    if (B < 1 || CA < 1 || CB < 0 || Cout < 1 || H < 1 || W < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if (kind == 1 && ((H | W) & 1)) return fail(SMVS_ERR_ARG, "stride 2: H and W must be even");
    if (kind == 2 && (CB > 0 || bias || init)) return fail(SMVS_ERR_ARG, "transposed layer: one operand, no bias, no initial sums");
    const int Ho = kind == 1 ? H / 2 : kind == 2 ? 2 * H : H, Wo = kind == 1 ? W / 2 : kind == 2 ? 2 * W : W;
    if ((long long)(CA > CB ? CA : CB) * H * W * 4 >= (1ll << 31) || (long long)Cout * Ho * Wo * 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "plane too large");
    if ((long long)B * ((Cout + COT - 1) / COT) > 65535) return fail(SMVS_ERR_ARG, "batch too large");
    const int Cin = CA + CB;
    ConvArgs a{};
    a.inA = xA; a.CA = CA; a.scaleA = 1.0f; a.inB = CB > 0 ? xB : nullptr; a.CB = CB;
    a.w = packed; a.bias = bias; a.init = init; a.out = out; a.Cout = Cout; a.Hi = H; a.Wi = W; a.Ho = Ho; a.Wo = Wo; a.relu = relu ? 1 : 0;
    if (kind == 2) launch_convT(a, B, (hipStream_t)stream);
    else {
        // the MFMA kernel reads its per-channel vectors as aligned float4 and pairs input channels: both operands even, bias 16-byte aligned
        const bool mf = mfma_conv_ok(Cin, 0, Cout) && CA % 2 == 0 && (!bias || ((uintptr_t)bias & 15) == 0);
        launch_conv(kind == 1 ? 2 : 1, a, B, (hipStream_t)stream, mf ? packed + packed_conv_floats(Cin, Cout) : nullptr);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "conv3x3_fwd launch: %s", hipGetErrorString(e));
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
    const RedWorkspace ws = red_workspace(B, C, H, W);
    if (workspace_bytes < ws.total * sizeof(float)) return fail(SMVS_ERR_ARG, "workspace too small: %zu < %zu bytes", workspace_bytes, ws.total * sizeof(float));
    if ((long long)(C + 8) * H * W * 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "plane too large");
    RedRun r{};
    r.packed = packed; r.state[0] = state1; r.state[1] = state2; r.state[2] = state3; r.state[3] = state4;
    r.wsf = (float*)workspace; r.B = B; r.C = C; r.H = H; r.W = W; r.main = (hipStream_t)stream;
    r.cost = cost; r.reg_out = reg_out; r.pred = false;
    return red_run_planes(r, 0, 1);
}

// The whole plane loop of compute_depth_when_pred (networks/casred.py:191-231) for planes [d_begin,d_end)
// in ONE call: per plane the fused warp+variance build of that plane, the RED step and the float64
// streaming-regression update are enqueued back to back from C (8 launches per plane + 4 per chunk of planes, no Python
// or allocator work in between).  acc = (3,B,H,W) float64 [exp_sum, depth_img, max_prob], zeroed by the
// caller before plane 0; states as in smvs_red_step_fwd.  geo_kind 0: rpc (B,V,170); 1: composed
// homographies (B,n_src,4,4).
SMVS_EXPORT size_t smvs_red_pred_workspace_bytes(int B, int C, int H, int W)
{
    const size_t r = smvs_red_workspace_bytes(B, C, H, W);
    if (r == 0) return 0;
    return r + 2 * (size_t)smvs::red_chunk(B, C, H, W) * B * C * H * W * sizeof(float) + 64;   // two chunks of variance planes
}

static int red_planes_entry(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                            const double* geo, const float* depth, int depth_is_4d, const smvs_height_gen* gen, const float* packed,
                            float* state1, float* state2, float* state3, float* state4, double* acc, float* reg_volume,
                            void* workspace, size_t workspace_bytes,
                            int B, int C, int D, int H, int W, int d_begin, int d_end, void* stream)
{
    using namespace smvs;
    if ((!acc && !reg_volume) || !workspace) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (geo_kind != 0 && geo_kind != 1) return fail(SMVS_ERR_ARG, "geo_kind must be 0 (rpc) or 1 (homography)");
    const size_t need = smvs_red_pred_workspace_bytes(B, C, H, W);
    if (need == 0) return fail(SMVS_ERR_ARG, "plane %dx%d must be a positive multiple of 8 in both dimensions", H, W);
    if (workspace_bytes < need) return fail(SMVS_ERR_ARG, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    if (d_begin < 0 || d_end > D || d_begin > d_end) return fail(SMVS_ERR_ARG, "bad plane range [%d,%d) of %d", d_begin, d_end, D);
    if (!packed || !state1 || !state2 || !state3 || !state4 || !ref_fea || !src_fea || !geo || (!depth && !gen))
        return fail(SMVS_ERR_ARG, "null pointer argument");
    const size_t red_bytes = smvs_red_workspace_bytes(B, C, H, W);
    RedRun r{};
    r.packed = packed; r.state[0] = state1; r.state[1] = state2; r.state[2] = state3; r.state[3] = state4;
    r.wsf = (float*)workspace; r.B = B; r.C = C; r.H = H; r.W = W; r.main = (hipStream_t)stream;
    r.pred = true; r.geo_kind = geo_kind; r.ref_fea = ref_fea; r.src_fea = src_fea; r.n_src = n_src; r.geo = geo;
    r.depth = depth; r.depth_is_4d = depth_is_4d; r.gen = gen; r.acc = acc; r.reg_volume = reg_volume; r.D = D;
    // Arithmetic of the variance planes: what the call carries; a call that carries nothing gets the reference's rounding
    // sequence (SMVS_ARITH_EXACT), NOT the process default of the stand-alone builds -- behind a peaky softmax the fused
    // volume's 1e-5 can move a regressed height by more than north_star's 1e-3 m (2.1e-3 m at 3 of 294 912 pixels of the
    // conditioned 768 x 384 cascade), and the build is a few per cent of a pipeline's time (include/satmvs.h, smvs_set_arith).
    if (gen) {
        r.gen_own = *gen;
        if (!(r.gen_own.arith & SMVS_CALL_ARITH_MASK)) r.gen_own.arith |= SMVS_CALL_ARITH_EXACT;
        r.gen = &r.gen_own;
    } else if (!(r.depth_is_4d & SMVS_CALL_ARITH_MASK)) {
        r.depth_is_4d |= SMVS_CALL_ARITH_EXACT;
    }
    const size_t blk = (size_t)red_chunk(B, C, H, W) * B * C * H * W;
    r.block[0] = (float*)((char*)workspace + ((red_bytes + 15) & ~(size_t)15));
    r.block[1] = r.block[0] + blk;
    return red_run_planes(r, d_begin, d_end);
}

SMVS_EXPORT int smvs_red_pred_planes(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                                     const double* geo, const float* depth, int depth_is_4d, const float* packed,
                                     float* state1, float* state2, float* state3, float* state4, double* acc,
                                     void* workspace, size_t workspace_bytes,
                                     int B, int C, int D, int H, int W, int d_begin, int d_end, void* stream)
{
    if (!acc) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    return red_planes_entry(geo_kind, ref_fea, src_fea, n_src, geo, depth, depth_is_4d, nullptr, packed, state1, state2, state3, state4,
                            acc, nullptr, workspace, workspace_bytes, B, C, D, H, W, d_begin, d_end, stream);
}

SMVS_EXPORT int smvs_red_pred_planes_gen(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                                         const double* geo, const smvs_height_gen* gen, const float* packed,
                                         float* state1, float* state2, float* state3, float* state4, double* acc,
                                         void* workspace, size_t workspace_bytes,
                                         int B, int C, int D, int H, int W, int d_begin, int d_end, void* stream)
{
    if (!acc || !gen) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    return red_planes_entry(geo_kind, ref_fea, src_fea, n_src, geo, nullptr, 0, gen, packed, state1, state2, state3, state4,
                            acc, nullptr, workspace, workspace_bytes, B, C, D, H, W, d_begin, d_end, stream);
}

// Same plane pipeline, but the regularised planes are written to reg_volume (B,D,H,W) (planes [d_begin,d_end))
// instead of being folded into the regression accumulators: RED_Regularization.forward of the whole-volume
// network (networks/casred.py:22-62, modules/module.py:625-647) without ever materialising the (B,C,D,H,W)
// variance volume.  The caller applies softmax + regression (smvs_softmax_regress_fwd) to reg_volume.
SMVS_EXPORT int smvs_red_volume_planes(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                                       const double* geo, const float* depth, int depth_is_4d, const float* packed,
                                       float* state1, float* state2, float* state3, float* state4, float* reg_volume,
                                       void* workspace, size_t workspace_bytes,
                                       int B, int C, int D, int H, int W, int d_begin, int d_end, void* stream)
{
    if (!reg_volume) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    return red_planes_entry(geo_kind, ref_fea, src_fea, n_src, geo, depth, depth_is_4d, nullptr, packed, state1, state2, state3, state4,
                            nullptr, reg_volume, workspace, workspace_bytes, B, C, D, H, W, d_begin, d_end, stream);
}

SMVS_EXPORT int smvs_red_volume_planes_gen(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                                           const double* geo, const smvs_height_gen* gen, const float* packed,
                                           float* state1, float* state2, float* state3, float* state4, float* reg_volume,
                                           void* workspace, size_t workspace_bytes,
                                           int B, int C, int D, int H, int W, int d_begin, int d_end, void* stream)
{
    if (!reg_volume || !gen) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    return red_planes_entry(geo_kind, ref_fea, src_fea, n_src, geo, nullptr, 0, gen, packed, state1, state2, state3, state4,
                            nullptr, reg_volume, workspace, workspace_bytes, B, C, D, H, W, d_begin, d_end, stream);
}

}  // extern "C"
