// costreg.hip -- native forward of the 3-D convolutional cost regulariser (eval mode).
//
// Replaces CostRegNet.forward (/root/reference/modules/module.py:546-577) with its Conv3d / Deconv3d
// blocks (:324-410): conv0 (C->8), three [stride-2 conv, conv] pairs to 16/32/64 channels, three stride-2
// transposed convolutions back with additive skips, and the final 3x3x3 `prob` convolution to one channel;
// every block conv + BatchNorm3d + ReLU.  Used by CascadeMVSNet (networks/casmvs.py) and UCSNet
// (networks/ucs.py).  BatchNorm runs in inference form (running statistics folded into a per-channel
// scale/shift in the epilogue); training (batch statistics, autograd) stays on the PyTorch composite.
//
// Kernels, 11 launches per volume:
//   * conv3..conv6 (>= 32 output channels): float32 MFMA implicit GEMM (mfma_conv.h, 27 taps);
//   * stride-1 8/16-channel layers (conv0, conv2, prob): conv3d_s1_kernel -- one aligned row load per input
//     row, x+-1 taps by DPP wave shifts, channel pairs on v_pk_fma_f32;
//   * stride-2 conv1: conv3d_kernel<2>, direct gathers (one lane per output voxel, 8 output channels per lane,
//     wave-uniform weights from a packed buffer [cout/8][cin][27][8], zero padding from the buffer range check);
//   * transposed conv7/conv9/conv11: convT3d_kernel (one lane per input voxel -> 2x2x2 outputs), and on the
//     coarse levels convT3d_split_kernel (linear voxel -> lane mapping, input channels split over 4 waves).
// BN scale/shift + ReLU + skip-add are fused in every epilogue.
#include <stdlib.h>

#include "smvs_device.h"
#include "smvs_host.h"
#include "mfma_conv.h"

namespace smvs {

constexpr int CR_COT = 8;
constexpr int CR_S1_TILE = 62;                // outputs of a wave of the stride-1 row kernel (64 loaded columns minus the two halo lanes)
constexpr int CR_NL = 11;                    // conv0,1,2,3,4,5,6, conv7,9,11 (transposed), prob

struct CrLayer { int cin, cout, stride, transposed, bn, relu; };

static void cr_layers(int C, CrLayer L[CR_NL])
{
    const CrLayer t[CR_NL] = {
        {C, 8, 1, 0, 1, 1}, {8, 16, 2, 0, 1, 1}, {16, 16, 1, 0, 1, 1}, {16, 32, 2, 0, 1, 1}, {32, 32, 1, 0, 1, 1},
        {32, 64, 2, 0, 1, 1}, {64, 64, 1, 0, 1, 1}, {64, 32, 2, 1, 1, 1}, {32, 16, 2, 1, 1, 1}, {16, 8, 2, 1, 1, 1},
        {8, 1, 1, 0, 0, 0}};
    for (int i = 0; i < CR_NL; ++i) L[i] = t[i];
}

static inline size_t cr_packed_conv(int cin, int cout) { return (size_t)((cout + CR_COT - 1) / CR_COT) * cin * 27 * CR_COT; }

struct CrLayout { size_t w[CR_NL], scale[CR_NL], shift[CR_NL], wm[CR_NL], total; };   // wm: MFMA-order weights

static bool cr_use_mfma(const CrLayer& l) { return !l.transposed && mfma_conv_ok(l.cin, 0, l.cout); }

static CrLayout cr_layout(int C)
{
    CrLayer L[CR_NL];
    cr_layers(C, L);
    CrLayout o{};
    size_t p = 0;
    for (int i = 0; i < CR_NL; ++i) {
        o.w[i] = p; p += cr_packed_conv(L[i].cin, L[i].cout);
        const int cp = ((L[i].cout + CR_COT - 1) / CR_COT) * CR_COT;
        o.scale[i] = p; p += cp;
        o.shift[i] = p; p += cp;
        o.wm[i] = p;
        if (cr_use_mfma(L[i])) p += mfma_packed_floats(L[i].cin, L[i].cout, 27);
    }
    o.total = p;
    return o;
}

// transposed: src is (cin, cout, 27) instead of (cout, cin, 27); flip (with transposed): taps mirrored in all three dimensions -- the
// ADJOINT of a stride-1 correlation of weight (cin, cout, 27) as a correlation from cin to cout channels (smvs_conv3d_pack layout 2)
// tail > 0 (single layers, smvs_conv3d_pack): `tail` ones and `tail` zeros follow the weights -- the identity scale / shift of the epilogue
__global__ void cr_pack_conv_kernel(const float* __restrict__ src, float* __restrict__ dst, int cin, int cout, int transposed, int flip, int tail)
{
    const int ncog = (cout + CR_COT - 1) / CR_COT;
    const int n = ncog * cin * 27 * CR_COT;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + 2 * tail; i += gridDim.x * blockDim.x) {
        if (i >= n) { dst[i] = i < n + tail ? 1.0f : 0.0f; continue; }
        const int j = i % CR_COT, k = (i / CR_COT) % 27, ci = (i / (CR_COT * 27)) % cin, cog = i / (CR_COT * 27 * cin);
        const int co = cog * CR_COT + j;
        float v = 0.0f;
        if (co < cout) v = transposed ? src[((size_t)ci * cout + co) * 27 + (flip ? 26 - k : k)] : src[((size_t)co * cin + ci) * 27 + k];
        dst[i] = v;
    }
}

// BatchNorm3d in inference form: y = x*scale + shift, scale = gamma/sqrt(var+eps), shift = beta - mean*scale
__global__ void cr_pack_bn_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                                  const float* __restrict__ var, float* __restrict__ scale, float* __restrict__ shift,
                                  int cout, int cpad, int has_bn)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cpad) return;
    float s = 1.0f, t = 0.0f;
    if (has_bn && c < cout) {
        s = gamma[c] / sqrtf(var[c] + 1e-5f);
        t = beta[c] - mean[c] * s;
    }
    scale[c] = s;
    shift[c] = t;
}

struct Conv3Args {
    const float* in; const float* w; const float* scale; const float* shift;
    const float* skip;                       // added after BN+ReLU (x = skip + block(x)), or null
    float* out;
    int Cin, Cout, Di, Hi, Wi, Do, Ho, Wo, relu;
};

typedef const float __attribute__((address_space(4))) * cw3_t;

// 3x3x3 correlation, pad 1, stride STRIDE; lane = one output voxel, CR_COT channels.
template <int STRIDE>
__global__ __launch_bounds__(256)
void conv3d_kernel(const Conv3Args a)
{
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yz = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int oy = yz % a.Ho, od = yz / a.Ho;
    const int ncog = (a.Cout + CR_COT - 1) / CR_COT;
    const int cog = blockIdx.z % ncog, b = blockIdx.z / ncog;
    const bool active = ox < a.Wo && od < a.Do;
    const int HWi = a.Hi * a.Wi;
    const size_t vol_i = (size_t)a.Di * HWi;

    uint32_t off[27];
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int id = od * STRIDE - 1 + kd, iy = oy * STRIDE - 1 + ky, ix = ox * STRIDE - 1 + kx;
                const bool in = active && id >= 0 && id < a.Di && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
                off[(kd * 3 + ky) * 3 + kx] = in ? (uint32_t)((id * a.Hi + iy) * a.Wi + ix) * 4u : SMVS_OOB;
            }
    const BufRsrc rs = make_rsrc(a.in + (size_t)b * a.Cin * vol_i, (uint32_t)((size_t)a.Cin * vol_i * 4));
    float acc[CR_COT];
#pragma unroll
    for (int j = 0; j < CR_COT; ++j) acc[j] = 0.0f;
    const cw3_t wbase = (cw3_t)(uintptr_t)(a.w + (size_t)cog * a.Cin * 27 * CR_COT);
    for (int ci = 0; ci < a.Cin; ++ci) {
        const int choff = (int)((size_t)ci * vol_i * 4);
        const cw3_t wc = wbase + (size_t)ci * 27 * CR_COT;
#pragma unroll
        for (int kd = 0; kd < 3; ++kd) {
            float v[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) v[k] = llvm_raw_buffer_load_f32(rs.v, (int)off[kd * 9 + k], choff, 0);
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int j = 0; j < CR_COT; ++j) acc[j] = fmaf(v[k], wc[(kd * 9 + k) * CR_COT + j], acc[j]);
        }
    }
    if (!active) return;
    const size_t vol_o = (size_t)a.Do * a.Ho * a.Wo;
    const size_t pos = ((size_t)od * a.Ho + oy) * a.Wo + ox;
    // skip values first, all in flight together (a load -> add -> store chain per channel is a memory round trip each;
    // channels beyond Cout read channel 0 and are not stored)
    float sk[CR_COT];
#pragma unroll
    for (int j = 0; j < CR_COT; ++j) {
        const int co = cog * CR_COT + j;
        sk[j] = a.skip ? a.skip[((size_t)b * a.Cout + (co < a.Cout ? co : 0)) * vol_o + pos] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < CR_COT; ++j) {
        const int co = cog * CR_COT + j;
        if (co < a.Cout) {
            float r = fmaf(acc[j], a.scale[co], a.shift[co]);
            if (a.relu) r = fmaxf(r, 0.0f);
            const size_t o = ((size_t)b * a.Cout + co) * vol_o + pos;
            if (a.skip) r = sk[j] + r;
            a.out[o] = r;
        }
    }
}

// Stride-1 3x3x3 correlation for the full-resolution layers (conv0, prob): the direct kernel above is bound by the
// texture-address path -- 27 gathers per channel, the +-1 x-taps unaligned (8 clocks each on this part).  Here a
// wave loads 64 consecutive x of one (d,y) row with ONE coalesced load per input row and computes the 62 inner ones: the
// x-1 / x+1 taps come from the neighbouring lanes (DPP wave shift), lanes 0 and 63 only carry the halo (round 3; before,
// a wave computed 64 outputs and fetched the halo with a second, two-lane load per row: the kernel was bound by the issue
// of those loads -- 18 instead of 9 per input channel).  Row validity is wave-uniform, so the row offset rides in the scalar offset and
// the 27-entry per-lane offset table disappears.  With the loads cheap the kernel turns VALU-bound, so the 8
// output channels are 4 register pairs on v_pk_fma_f32 (weight pairs straight from SGPRs).
// COT = output channels a lane computes: 8, or 2 for the single-channel `prob` layer (same packed weights: the first pair of each
// group of 8 -- with 8 the prob layer spent 7/8 of its arithmetic on padding channels).
template <int COT>
__global__ __launch_bounds__(256)
void conv3d_s1_kernel(const Conv3Args a)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ox = blockIdx.x * CR_S1_TILE - 1 + lane;      // lanes 1..62 own an output, lanes 0 / 63 are the west / east halo
    const int yz = blockIdx.y * 4 + wave;
    const int oy = yz % a.Ho, od = yz / a.Ho;
    const int ncog = (a.Cout + CR_COT - 1) / CR_COT;
    const int cog = blockIdx.z % ncog, b = blockIdx.z / ncog;
    if (od >= a.Do) return;                                  // wave-uniform
    const bool active = lane >= 1 && lane <= CR_S1_TILE && ox < a.Wo;
    const int HWi = a.Hi * a.Wi;
    const size_t vol_i = (size_t)a.Di * HWi;
    const uint32_t vc = (ox >= 0 && ox < a.Wi) ? (uint32_t)ox * 4u : SMVS_OOB;     // zero padding / beyond the row: the range check returns 0
    int rowoff[9];                                           // (kd,ky) -> byte offset of the input row, wave-uniform
#pragma unroll
    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int id = od - 1 + kd, iy = oy - 1 + ky;
            rowoff[kd * 3 + ky] = (id >= 0 && id < a.Di && iy >= 0 && iy < a.Hi) ? (id * a.Hi + iy) * a.Wi * 4 : -1;
        }
    const BufRsrc rs = make_rsrc(a.in + (size_t)b * a.Cin * vol_i, (uint32_t)((size_t)a.Cin * vol_i * 4));
    typedef const f32x2 __attribute__((address_space(4))) * cw3p_t;
    f32x2 acc2[COT / 2];
#pragma unroll
    for (int j = 0; j < COT / 2; ++j) acc2[j] = (f32x2)(0.0f);
    const cw3_t wbase = (cw3_t)(uintptr_t)(a.w + (size_t)cog * a.Cin * 27 * CR_COT);
    for (int ci = 0; ci < a.Cin; ++ci) {
        const int choff = (int)((size_t)ci * vol_i * 4);
        const cw3p_t wc = (cw3p_t)(wbase + (size_t)ci * 27 * CR_COT);
        float c[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int so = rowoff[r] >= 0 ? choff + rowoff[r] : (int)SMVS_OOB;      // scalar select
            c[r] = llvm_raw_buffer_load_f32(rs.v, (int)vc, so, 0);
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            // lane i <- lane i-1 (wave_shr:1) ; lane i <- lane i+1 (wave_shl:1); lanes 0 / 63 get 0 and produce no output
            const float l = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c[r]), 0x138, 0xf, 0xf, false));
            const float rr = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c[r]), 0x130, 0xf, 0xf, false));
            const f32x2 t0 = {l, c[r]};
            const f32x2 t1 = {rr, rr};
#pragma unroll
            for (int j = 0; j < COT / 2; ++j) {
                acc2[j] = __builtin_elementwise_fma(__builtin_shufflevector(t0, t0, 0, 0), wc[(r * 3 + 0) * (CR_COT / 2) + j], acc2[j]);
                acc2[j] = __builtin_elementwise_fma(__builtin_shufflevector(t0, t0, 1, 1), wc[(r * 3 + 1) * (CR_COT / 2) + j], acc2[j]);
                acc2[j] = __builtin_elementwise_fma(__builtin_shufflevector(t1, t1, 0, 0), wc[(r * 3 + 2) * (CR_COT / 2) + j], acc2[j]);
            }
        }
    }
    if (!active) return;
    const size_t vol_o = (size_t)a.Do * a.Ho * a.Wo;
    const size_t pos = ((size_t)od * a.Ho + oy) * a.Wo + ox;
    float sk[COT];                               // skip values first, all in flight together (see conv3d_kernel)
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = cog * CR_COT + j;
        sk[j] = a.skip ? a.skip[((size_t)b * a.Cout + (co < a.Cout ? co : 0)) * vol_o + pos] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < COT; ++j) {
        const int co = cog * CR_COT + j;
        if (co < a.Cout) {
            const float av = (j & 1) ? acc2[j >> 1].y : acc2[j >> 1].x;
            float r = fmaf(av, a.scale[co], a.shift[co]);
            if (a.relu) r = fmaxf(r, 0.0f);
            const size_t o = ((size_t)b * a.Cout + co) * vol_o + pos;
            if (a.skip) r = sk[j] + r;
            a.out[o] = r;
        }
    }
}

// ConvTranspose3d(k=3, stride=2, pad=1, output_padding=1): lane = one INPUT voxel (d,y,x) -> the 2x2x2 output
// block at (2d,2y,2x).  Per dimension an even output takes tap k=1 from input i; an odd output takes k=2 from
// input i and k=0 from input i+1.
__global__ __launch_bounds__(256)
void convT3d_kernel(const Conv3Args a)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yz = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int y = yz % a.Hi, d = yz / a.Hi;
    const int ncog = (a.Cout + CR_COT - 1) / CR_COT;
    const int cog = blockIdx.z % ncog, b = blockIdx.z / ncog;
    const bool active = x < a.Wi && d < a.Di;
    const int HWi = a.Hi * a.Wi;
    const size_t vol_i = (size_t)a.Di * HWi;
    uint32_t off[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int dd = d + (q >> 2), yy = y + ((q >> 1) & 1), xx = x + (q & 1);
        off[q] = (active && dd < a.Di && yy < a.Hi && xx < a.Wi) ? (uint32_t)((dd * a.Hi + yy) * a.Wi + xx) * 4u : SMVS_OOB;
    }
    const BufRsrc rs = make_rsrc(a.in + (size_t)b * a.Cin * vol_i, (uint32_t)((size_t)a.Cin * vol_i * 4));
    float acc[8][CR_COT];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int j = 0; j < CR_COT; ++j) acc[p][j] = 0.0f;
    const cw3_t wbase = (cw3_t)(uintptr_t)(a.w + (size_t)cog * a.Cin * 27 * CR_COT);
    for (int ci = 0; ci < a.Cin; ++ci) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = llvm_raw_buffer_load_f32(rs.v, (int)off[q], (int)((size_t)ci * vol_i * 4), 0);
        const cw3_t wc = wbase + (size_t)ci * 27 * CR_COT;
        // output parity p = (pd,py,px); per dimension: parity 0 -> (delta 0, k 1); parity 1 -> (delta 0, k 2), (delta 1, k 0)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int pd = p >> 2, py = (p >> 1) & 1, px = p & 1;
#pragma unroll
            for (int td = 0; td <= pd; ++td)
#pragma unroll
                for (int ty = 0; ty <= py; ++ty)
#pragma unroll
                    for (int tx = 0; tx <= px; ++tx) {
                        const int kd = pd ? (td ? 0 : 2) : 1, ky = py ? (ty ? 0 : 2) : 1, kx = px ? (tx ? 0 : 2) : 1;
                        const int q = (td << 2) | (ty << 1) | tx;
                        const int k = (kd * 3 + ky) * 3 + kx;
#pragma unroll
                        for (int j = 0; j < CR_COT; ++j) acc[p][j] = fmaf(v[q], wc[k * CR_COT + j], acc[p][j]);
                    }
        }
    }
    if (!active) return;
    const size_t vol_o = (size_t)a.Do * a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < CR_COT; ++j) {
        const int co = cog * CR_COT + j;
        if (co < a.Cout) {
            const float sc = a.scale[co], sh = a.shift[co];
            // the 2x2x2 block as four float2 rows (x pairs; 8-byte aligned: Wo is even); skip rows loaded before the first use
            size_t o[4];
            float2 sk[4];
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                o[pr] = ((size_t)b * a.Cout + co) * vol_o + ((size_t)(2 * d + (pr >> 1)) * a.Ho + (2 * y + (pr & 1))) * a.Wo + 2 * x;
                sk[pr] = a.skip ? *reinterpret_cast<const float2*>(a.skip + o[pr]) : make_float2(0.0f, 0.0f);
            }
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                float r0 = fmaf(acc[2 * pr][j], sc, sh), r1 = fmaf(acc[2 * pr + 1][j], sc, sh);
                if (a.relu) { r0 = fmaxf(r0, 0.0f); r1 = fmaxf(r1, 0.0f); }
                if (a.skip) { r0 = sk[pr].x + r0; r1 = sk[pr].y + r1; }
                *reinterpret_cast<float2*>(a.out + o[pr]) = make_float2(r0, r1);
            }
        }
    }
}

// Transposed convolution on the COARSE levels (a few thousand input voxels: 27..600 tiles on a 256-CU part).
// The kernel above wastes most lanes there (rows of 12..48 voxels in 64-lane tiles) and its time is one wave's
// serial walk over the input channels.  Here lanes are consecutive LINEAR input voxels (every lane busy), the 4
// waves of a workgroup split the input channels and reduce through LDS, and the 8 corner loads of the next
// channel are in flight while the current one is multiplied.
__global__ __launch_bounds__(256)
void convT3d_split_kernel(const Conv3Args a)
{
    __shared__ float part[3][64][64 + 1];                    // [wave-1][parity*8 + cout][lane]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int HWi = a.Hi * a.Wi;
    const int vol = a.Di * HWi;
    const int v = blockIdx.x * 64 + lane;
    const bool active = v < vol;
    const int d = v / HWi, y = (v % HWi) / a.Wi, x = v % a.Wi;
    const int ncog = (a.Cout + CR_COT - 1) / CR_COT;
    const int cog = blockIdx.z % ncog, b = blockIdx.z / ncog;
    const size_t vol_i = (size_t)vol;
    uint32_t off[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int dd = d + (q >> 2), yy = y + ((q >> 1) & 1), xx = x + (q & 1);
        off[q] = (active && dd < a.Di && yy < a.Hi && xx < a.Wi) ? (uint32_t)((dd * a.Hi + yy) * a.Wi + xx) * 4u : SMVS_OOB;
    }
    const BufRsrc rs = make_rsrc(a.in + (size_t)b * a.Cin * vol_i, (uint32_t)((size_t)a.Cin * vol_i * 4));
    float acc[8][CR_COT];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int j = 0; j < CR_COT; ++j) acc[p][j] = 0.0f;
    const cw3_t wbase = (cw3_t)(uintptr_t)(a.w + (size_t)cog * a.Cin * 27 * CR_COT);
    const int per = (a.Cin + 3) / 4, c0 = wave * per, c1 = min(a.Cin, c0 + per);
#define SMVS_T3_LOAD(V, CC) \
    { _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) V[q_] = llvm_raw_buffer_load_f32(rs.v, (int)off[q_], (int)((size_t)(CC) * vol_i * 4), 0); }
#define SMVS_T3_FMA(V, CC)                                                                             \
    {                                                                                                  \
        const cw3_t wc = wbase + (size_t)(CC) * 27 * CR_COT;                                           \
        _Pragma("unroll") for (int p = 0; p < 8; ++p) {                                                \
            const int pd = p >> 2, py = (p >> 1) & 1, px = p & 1;                                      \
            _Pragma("unroll") for (int td = 0; td <= pd; ++td)                                         \
                _Pragma("unroll") for (int ty = 0; ty <= py; ++ty)                                     \
                    _Pragma("unroll") for (int tx = 0; tx <= px; ++tx) {                               \
                        const int kd = pd ? (td ? 0 : 2) : 1, ky = py ? (ty ? 0 : 2) : 1, kx = px ? (tx ? 0 : 2) : 1; \
                        const int q = (td << 2) | (ty << 1) | tx;                                      \
                        const int k = (kd * 3 + ky) * 3 + kx;                                          \
                        _Pragma("unroll") for (int j = 0; j < CR_COT; ++j) acc[p][j] = fmaf(V[q], wc[k * CR_COT + j], acc[p][j]); \
                    }                                                                                  \
        }                                                                                              \
    }
    // prefetch loads are unconditional (past the end: the last channel again): a load under a branch makes the compiler
    // wait with vmcnt(0) before every multiply block, i.e. also for the loads it has just issued (mfma_conv.h)
    float v0[8], v1[8];
    const int cl = max(c1 - 1, 0);
    SMVS_T3_LOAD(v0, min(c0, cl))
    for (int cc = c0; cc < c1; cc += 2) {
        SMVS_T3_LOAD(v1, min(cc + 1, cl))
        __builtin_amdgcn_sched_barrier(0);
        SMVS_T3_FMA(v0, cc)
        __builtin_amdgcn_sched_barrier(0);
        SMVS_T3_LOAD(v0, min(cc + 2, cl))
        __builtin_amdgcn_sched_barrier(0);
        if (cc + 1 < c1) SMVS_T3_FMA(v1, cc + 1)
        __builtin_amdgcn_sched_barrier(0);
    }
#undef SMVS_T3_LOAD
#undef SMVS_T3_FMA
    if (wave > 0) {
#pragma unroll
        for (int p = 0; p < 8; ++p)
#pragma unroll
            for (int j = 0; j < CR_COT; ++j) part[wave - 1][p * CR_COT + j][lane] = acc[p][j];
    }
    __syncthreads();
    if (wave > 0 || !active) return;
    const size_t vol_o = (size_t)a.Do * a.Ho * a.Wo;
#pragma unroll
    for (int j = 0; j < CR_COT; ++j) {
        const int co = cog * CR_COT + j;
        if (co < a.Cout) {
            const float sc = a.scale[co], sh = a.shift[co];
            size_t o[4];                         // four float2 rows, skip rows loaded before the first use (see convT3d_kernel)
            float2 sk[4];
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                o[pr] = ((size_t)b * a.Cout + co) * vol_o + ((size_t)(2 * d + (pr >> 1)) * a.Ho + (2 * y + (pr & 1))) * a.Wo + 2 * x;
                sk[pr] = a.skip ? *reinterpret_cast<const float2*>(a.skip + o[pr]) : make_float2(0.0f, 0.0f);
            }
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                float r[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int p = 2 * pr + e;
                    const float s = acc[p][j] + part[0][p * CR_COT + j][lane] + part[1][p * CR_COT + j][lane] + part[2][p * CR_COT + j][lane];
                    r[e] = fmaf(s, sc, sh);
                    if (a.relu) r[e] = fmaxf(r[e], 0.0f);
                }
                if (a.skip) { r[0] = sk[pr].x + r[0]; r[1] = sk[pr].y + r[1]; }
                *reinterpret_cast<float2*>(a.out + o[pr]) = make_float2(r[0], r[1]);
            }
        }
    }
}

// One layer on the kernels above: wm = MFMA-order weights (null: direct kernels)
static void cr_launch_layer(const CrLayer& l, const float* in, const float* w, const float* scale, const float* shift, const float* wm,
                            const float* skip, float* out, int B, const int (&di)[3], const int (&dout)[3], bool direct_only, hipStream_t st)
{
    Conv3Args a{};
    a.in = in; a.w = w; a.scale = scale; a.shift = shift;
    a.skip = skip; a.out = out; a.Cin = l.cin; a.Cout = l.cout; a.relu = l.relu;
    a.Di = di[0]; a.Hi = di[1]; a.Wi = di[2];
    a.Do = dout[0]; a.Ho = dout[1]; a.Wo = dout[2];
    const int ncog = (l.cout + CR_COT - 1) / CR_COT;
    if (wm) {
        MfmaConvArgs m{};
        m.inA = in; m.CA = l.cin; m.scaleA = 1.0f; m.w = wm;
        m.scale = a.scale; m.shift = a.shift; m.skip = skip; m.out = out;
        m.Cout = l.cout; m.relu = l.relu; m.stride = l.stride;
        m.Di = a.Di; m.Hi = a.Hi; m.Wi = a.Wi; m.Do = a.Do; m.Ho = a.Ho; m.Wo = a.Wo;
        mfma_conv_launch<27>(m, B, st);
    } else if (l.transposed) {
        dim3 grd((a.Wi + 63) / 64, (a.Hi * a.Di + 3) / 4, B * ncog);
        static const int split_below = tune_int("SMVS_CONV_SPLIT_BELOW", 1024);
        if ((long long)((a.Di * a.Hi * a.Wi + 63) / 64) * B * ncog < split_below / 2 && !direct_only)
            hipLaunchKernelGGL(convT3d_split_kernel, dim3((a.Di * a.Hi * a.Wi + 63) / 64, 1, B * ncog), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL(convT3d_kernel, grd, dim3(256), 0, st, a);
    } else {
        dim3 grd((a.Wo + 63) / 64, (a.Ho * a.Do + 3) / 4, B * ncog);
        const dim3 grd1((a.Wo + CR_S1_TILE - 1) / CR_S1_TILE, (a.Ho * a.Do + 3) / 4, B * ncog);
        static const bool gather = tune_int("SMVS_CONV3D_GATHER", 0) == 1;
        if (l.stride == 1 && !gather && l.cout <= 2) hipLaunchKernelGGL(conv3d_s1_kernel<2>, grd1, dim3(256), 0, st, a);
        else if (l.stride == 1 && !gather) hipLaunchKernelGGL(conv3d_s1_kernel<CR_COT>, grd1, dim3(256), 0, st, a);
        else if (l.stride == 1)       hipLaunchKernelGGL(conv3d_kernel<1>, grd, dim3(256), 0, st, a);
        else               hipLaunchKernelGGL(conv3d_kernel<2>, grd, dim3(256), 0, st, a);
    }
}

struct CrWorkspace { size_t c0, t1, c2, t3, c4, t5, t6, x7, x9, x11, total; };

static CrWorkspace cr_workspace(int B, int D, int H, int W)
{
    CrWorkspace w{};
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 3) & ~(size_t)3; return r; };
    const size_t v1 = (size_t)D * H * W, v2 = v1 / 8, v4 = v1 / 64, v8 = v1 / 512;
    w.c0 = take(B * 8 * v1);  w.t1 = take(B * 16 * v2); w.c2 = take(B * 16 * v2); w.t3 = take(B * 32 * v4);
    w.c4 = take(B * 32 * v4); w.t5 = take(B * 64 * v8); w.t6 = take(B * 64 * v8);
    w.x7 = take(B * 32 * v4); w.x9 = take(B * 16 * v2); w.x11 = take(B * 8 * v1);
    w.total = o;
    return w;
}

}  // namespace smvs

extern "C" {

SMVS_EXPORT size_t smvs_costreg_packed_floats(int C) { return C > 0 ? smvs::cr_layout(C).total : 0; }

SMVS_EXPORT size_t smvs_costreg_workspace_bytes(int B, int C, int D, int H, int W)
{
    if (B < 1 || C < 1 || D < 8 || H < 8 || W < 8 || (D % 8) || (H % 8) || (W % 8)) return 0;
    if ((long long)D * H / 4 + 1 > 65535) return 0;         // beyond one launch grid: unsupported (callers fall back)
    return smvs::cr_workspace(B, D, H, W).total * sizeof(float);
}

// params: HOST array of 51 device pointers: for conv0, conv1, conv2, conv3, conv4, conv5, conv6, conv7, conv9,
// conv11: conv.weight, bn.weight, bn.bias, bn.running_mean, bn.running_var; then prob.weight.
SMVS_EXPORT int smvs_costreg_pack_weights(const float* const* params, int C, float* packed, void* stream)
{
    using namespace smvs;
    if (!params || !packed) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (C < 1) return fail(SMVS_ERR_ARG, "non-positive channel count");
    for (int i = 0; i < 51; ++i)
        if (!params[i]) return fail(SMVS_ERR_ARG, "null parameter pointer %d", i);
    CrLayer L[CR_NL];
    cr_layers(C, L);
    const CrLayout lay = cr_layout(C);
    hipStream_t st = (hipStream_t)stream;
    for (int i = 0; i < CR_NL; ++i) {
        const float* const* q = params + (i < 10 ? i * 5 : 50);
        const int n = (int)cr_packed_conv(L[i].cin, L[i].cout);
        hipLaunchKernelGGL(cr_pack_conv_kernel, dim3((n + 255) / 256), dim3(256), 0, st, q[0], packed + lay.w[i],
                           L[i].cin, L[i].cout, L[i].transposed, 0, 0);
        if (cr_use_mfma(L[i])) {
            const int nm = (int)mfma_packed_floats(L[i].cin, L[i].cout, 27);
            hipLaunchKernelGGL(mfma_pack_kernel, dim3((nm + 255) / 256), dim3(256), 0, st, q[0], packed + lay.wm[i],
                               L[i].cin, L[i].cout, 27, 0);
        }
        const int cp = ((L[i].cout + CR_COT - 1) / CR_COT) * CR_COT;
        hipLaunchKernelGGL(cr_pack_bn_kernel, dim3(1), dim3(64), 0, st, i < 10 ? q[1] : q[0], i < 10 ? q[2] : q[0],
                           i < 10 ? q[3] : q[0], i < 10 ? q[4] : q[0], packed + lay.scale[i], packed + lay.shift[i],
                           L[i].cout, cp, L[i].bn);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "costreg_pack_weights launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

// vol (B,C,D,H,W) variance volume -> out (B,1,D,H,W) regularised cost.  D, H, W multiples of 8.
SMVS_EXPORT int smvs_costreg_fwd(const float* packed, const float* vol, float* out, void* workspace,
                                 size_t workspace_bytes, int B, int C, int D, int H, int W, void* stream)
{
    using namespace smvs;
    if (!packed || !vol || !out || !workspace) return fail(SMVS_ERR_ARG, "null pointer argument");
    const size_t need = smvs_costreg_workspace_bytes(B, C, D, H, W);
    if (need == 0) return fail(SMVS_ERR_ARG, "volume %dx%dx%d must be a positive multiple of 8 in every dimension", D, H, W);
    if (workspace_bytes < need) return fail(SMVS_ERR_ARG, "workspace too small: %zu < %zu bytes", workspace_bytes, need);
    if ((long long)D * H / 4 + 1 > 65535) return fail(SMVS_ERR_ARG, "volume too large for one launch grid");
    CrLayer L[CR_NL];
    cr_layers(C, L);
    const CrLayout lay = cr_layout(C);
    const CrWorkspace ws = cr_workspace(B, D, H, W);
    const bool direct_only = tune_int("SMVS_CONV_DIRECT", 0) == 1;      // A/B switch (tuning builds): direct kernels only
    float* f = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int dims[4][3] = {{D, H, W}, {D / 2, H / 2, W / 2}, {D / 4, H / 4, W / 4}, {D / 8, H / 8, W / 8}};
    struct Step { int layer; const float* in; float* out; const float* skip; int lin, lout; };
    const Step steps[CR_NL] = {
        {0, vol, f + ws.c0, nullptr, 0, 0},        {1, f + ws.c0, f + ws.t1, nullptr, 0, 1},
        {2, f + ws.t1, f + ws.c2, nullptr, 1, 1},  {3, f + ws.c2, f + ws.t3, nullptr, 1, 2},
        {4, f + ws.t3, f + ws.c4, nullptr, 2, 2},  {5, f + ws.c4, f + ws.t5, nullptr, 2, 3},
        {6, f + ws.t5, f + ws.t6, nullptr, 3, 3},  {7, f + ws.t6, f + ws.x7, f + ws.c4, 3, 2},
        {8, f + ws.x7, f + ws.x9, f + ws.c2, 2, 1}, {9, f + ws.x9, f + ws.x11, f + ws.c0, 1, 0},
        {10, f + ws.x11, out, nullptr, 0, 0}};
    for (int i = 0; i < CR_NL; ++i) {
        const Step& s = steps[i];
        const CrLayer& l = L[s.layer];
        if ((long long)l.cin * dims[s.lin][0] * dims[s.lin][1] * dims[s.lin][2] * 4 >= (1ll << 32))
            return fail(SMVS_ERR_ARG, "layer %d input larger than 4 GiB per batch item", i);
        cr_launch_layer(l, s.in, packed + lay.w[s.layer], packed + lay.scale[s.layer], packed + lay.shift[s.layer],
                        cr_use_mfma(l) && !direct_only ? packed + lay.wm[s.layer] : nullptr, s.skip, s.out, B, dims[s.lin], dims[s.lout],
                        direct_only, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "costreg_fwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

// ---- single layers: the TRAINING forward of CostRegNet's convolutions and their input gradients (include/satmvs.h) ----------------------
// packed buffer of one layer: [direct-order weights][ones (cout padded to 8)][zeros][MFMA-order weights where that kernel takes the layer]
SMVS_EXPORT size_t smvs_conv3d_packed_floats(int cin, int cout)
{
    if (cin < 1 || cout < 1) return 0;
    const size_t cp = (size_t)((cout + smvs::CR_COT - 1) / smvs::CR_COT) * smvs::CR_COT;
    return smvs::cr_packed_conv(cin, cout) + 2 * cp + (smvs::mfma_conv_ok(cin, 0, cout) ? smvs::mfma_packed_floats(cin, cout, 27) : 0);
}

SMVS_EXPORT int smvs_conv3d_pack(const float* w, float* packed, int cin, int cout, int layout, void* stream)
{
    using namespace smvs;
    if (!w || !packed) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (cin < 1 || cout < 1) return fail(SMVS_ERR_ARG, "non-positive channel count");
    if (layout < 0 || layout > 2) return fail(SMVS_ERR_ARG, "layout must be 0, 1 or 2");
    hipStream_t st = (hipStream_t)stream;
    const int n = (int)cr_packed_conv(cin, cout);
    const int cp = ((cout + CR_COT - 1) / CR_COT) * CR_COT;
    hipLaunchKernelGGL(cr_pack_conv_kernel, dim3((n + 2 * cp + 255) / 256), dim3(256), 0, st, w, packed, cin, cout, layout != 0, layout == 2, cp);
    if (mfma_conv_ok(cin, 0, cout) && layout != 1) {
        const int nm = (int)mfma_packed_floats(cin, cout, 27);
        hipLaunchKernelGGL(mfma_pack_kernel, dim3((nm + 255) / 256), dim3(256), 0, st, w, packed + n + 2 * cp, cin, cout, 27, layout == 2);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "conv3d_pack launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

SMVS_EXPORT int smvs_conv3d_fwd(int kind, const float* in, const float* packed, const float* skip, float* out, int B, int Cin, int Cout,
                                int Di, int Hi, int Wi, int relu, void* stream)
{
    using namespace smvs;
    if (!in || !packed || !out) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (kind < 0 || kind > 2) return fail(SMVS_ERR_ARG, "kind must be 0 (stride 1), 1 (stride 2) or 2 (transposed, stride 2)");
    if (B < 1 || Cin < 1 || Cout < 1 || Di < 1 || Hi < 1 || Wi < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if (kind == 1 && ((Di | Hi | Wi) & 1)) return fail(SMVS_ERR_ARG, "stride-2 layer needs even D, H, W (got %dx%dx%d)", Di, Hi, Wi);
    const int di[3] = {Di, Hi, Wi};
    const int dout[3] = {kind == 1 ? Di / 2 : kind == 2 ? Di * 2 : Di, kind == 1 ? Hi / 2 : kind == 2 ? Hi * 2 : Hi,
                         kind == 1 ? Wi / 2 : kind == 2 ? Wi * 2 : Wi};
    const long long vi = (long long)Di * Hi * Wi, vo = (long long)dout[0] * dout[1] * dout[2];
    if (Cin * vi * 4 >= (1ll << 32) || Cout * vo * 4 >= (1ll << 32)) return fail(SMVS_ERR_ARG, "layer input or output larger than 4 GiB per batch item");
    const int ncog = (Cout + CR_COT - 1) / CR_COT;
    const long long rows = kind == 2 ? (long long)Di * Hi : (long long)dout[0] * dout[1];
    if ((rows + 3) / 4 > 65535 || (long long)B * ncog > 65535 || vo / 32 * B >= (1ll << 31)) return fail(SMVS_ERR_ARG, "volume too large for one launch grid");
    const CrLayer l{Cin, Cout, kind == 0 ? 1 : 2, kind == 2, 0, relu ? 1 : 0};
    const size_t n = cr_packed_conv(Cin, Cout);
    const int cp = ncog * CR_COT;
    const bool mf = kind != 2 && mfma_conv_ok(Cin, 0, Cout);
    cr_launch_layer(l, in, packed, packed + n, packed + n + cp, mf ? packed + n + 2 * cp : nullptr, skip, out, B, di, dout, false, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "conv3d_fwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

}  // extern "C"
