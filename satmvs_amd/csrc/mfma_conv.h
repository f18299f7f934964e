// mfma_conv.h -- float32 MFMA implicit-GEMM convolution shared by the regularisers (red.hip 2-D, costreg.hip 3-D).
//
// For the levels whose channel counts reach the matrix-core tile (Cout >= 32) the convolution is the GEMM
//   D[cout][pos] = sum_kk W[cout][kk] * X[kk][pos],   kk = (input channel, tap),
// on v_mfma_f32_32x32x2_f32 (exact float32: the result is a k-ordered fmaf chain, so numerics stay in the
// same class as the direct kernels).  Rows (M) are 32 output channels, columns (N) 32 consecutive output
// positions along x -- so every accumulator register of a lane belongs to ONE output position and the store
// of a register across lanes is a coalesced 128-B row -- and K advances two (channel, tap) pairs per MFMA.
//   * 27 taps (3-D): X operand: one buffer load per lane and MFMA (lane -> position l&31, k-slot l>>5); the per-lane
//     tap offsets (zero padding = out-of-range offset) are fixed per tile, the channel pair rides in the scalar
//     offset: no vector ALU work in the K loop.  W operand: one coalesced load per lane and MFMA from a buffer
//     pre-packed in exactly the order the lanes consume it ([channel pair][step][cout tile][k-slot][32]).
//   * 9 taps (2-D): a CU's texture path, not the matrix pipe, bounded that scheme on the small grids (18 dword loads
//     per 9 MFMAs and wave: measured ~1700 clocks per batch against 576 of MFMA issue).  So k-slot = CHANNEL of the
//     pair and step = tap: a lane's nine X values are three rows of three consecutive floats = 3 dwordx3 loads (edge
//     columns fixed up with selects), its nine weights are contiguous = 3 dwordx4 loads
//     ([channel pair][cout tile][k-slot][32][12]): 6 loads per batch instead of 18.
//   * The coarse levels are tiny (a few thousand positions), so a workgroup's 4 waves split K (input
//     channels) four ways and reduce through LDS: 4x the waves in flight for the same tile.
// Epilogue (wave 0): bias / BatchNorm scale-shift, ReLU, skip add, GroupNorm statistics.
#pragma once
#include <type_traits>

#include "smvs_device.h"

namespace smvs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float mf32x3 __attribute__((ext_vector_type(3)));
typedef float mf32x4 __attribute__((ext_vector_type(4)));
__device__ mf32x3 llvm_raw_buffer_load_v3f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v3f32");
__device__ mf32x4 llvm_raw_buffer_load_v4f32(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
constexpr int MFMA9_WSLOT = 12;              // 9-tap weights: a lane's 9 taps padded to three float4
#ifndef SMVS_MFMA_PREFETCH
#define SMVS_MFMA_PREFETCH 2                 // 9-tap kernels, one cout tile per workgroup: channel pairs in flight per wave
#endif

// ---- element-wise ConvGRU stage computed by the CONSUMING convolution (RED plane loop, red.hip) ------------------------------
// The plane loop of the RED regulariser is a chain of dependent launches whose cost is their number, not their work
// (DESIGN.md section 6): gate convolution -> gate apply -> candidate convolution -> combine.  GroupNorm(1, C) needs the statistics
// of the WHOLE convolution output before an element can be normalised, so the element-wise stages cannot move into their
// producer's epilogue; they can move into their CONSUMER, which starts after the producer's launch boundary anyway: the
// consuming convolution computes the stage once per input tile (+ the 3x3 halo) into LDS and reads its hidden-state operand
// (the B half of the concatenated input) from there.  2 dependent launches per plane instead of 4:
//   mode FUSE_APPLY    candidate convolution:  B = sigmoid(GN(reset gate)) * h                            (module.py:34-44)
//   mode FUSE_COMBINE  NEXT plane's gate convolution:  B = h' = u h + (1 - u) tanh(GN(candidate)), u = sigmoid(GN(update gate));
//                      the workgroups of output-channel group 0 also store h' (new state + the decoder's snapshot)   (module.py:45-57)
// Same float32 operations as gru_gate_apply_kernel / gru_combine_kernel (red.hip), so the bits do not change.
enum { FUSE_NONE = 0, FUSE_APPLY = 1, FUSE_COMBINE = 2 };
struct FuseB {
    int mode, HC, nslot;
    const float* gates;                      // raw gate convolution (B, 2 HC, h, w) of the plane the stage belongs to
    const float* cand;                       // raw candidate convolution (B, HC, h, w)                      (combine)
    const float* h;                          // hidden state the stage reads (B, HC, h, w)
    float* h_out; float* hsnap;              // where the new state goes (another buffer than h), and its copy for the decoder   (combine)
    const double *stats_g, *stats_o;         // [b][reset, update][nslot][2] and [b][nslot][2] partial sums
    const float *gw, *gb, *ow, *ob;          // affine of the gate norm used (reset: apply, update: combine) and of the output norm
    double* zero; int zero_n;                // statistics buffer of a later plane, cleared by workgroup 0 of the job (or null)
};

// sigmoid / tanh of the ConvGRU gates on the hardware's exp2 and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each): 6-7 instructions
// instead of libm's ~40 -- the fused launches evaluate them 3-12 times per element (tile halos, one tile per output-channel
// group), and the stand-alone element-wise kernels use the same two functions so that a plane's bits do not depend on the path.
// |error| <= 2e-7 absolute (tolerance of the regulariser against the reference: 2e-5).
__device__ __forceinline__ float fuse_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float fuse_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// mean / rstd of one norm group from its 64 partial (sum, sumsq) slots, by ONE wave (lane = slot); same arithmetic as
// gn_coeffs (red.hip)
__device__ __forceinline__ void fuse_gn_coeffs(const double* st, double n, float& mean, float& rstd)
{
    const int l = threadIdx.x & 63;
    double a = st[l * 2], q = st[l * 2 + 1];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { a += __shfl_xor(a, m, 64); q += __shfl_xor(q, m, 64); }
    const double mu = a / n;
    double var = q / n - mu * mu;
    var = var < 0.0 ? 0.0 : var;
    mean = (float)mu;
    rstd = (float)(1.0 / sqrt(var + (double)1e-5f));
}

// Fill the LDS tile [HC][TH][TW] (rows y0 .. y0+TH-1, columns x0 .. x0+TW-1 of sample b; zeros outside the image) with the
// stage's output.  Called by the NWF waves that share the tile (wf = this wave's index among them); the caller synchronises.
// writer: this workgroup stores the new state for the tile's interior (rows y0+1 .. y0+TH-2, columns x0+1 .. x0+TW-2).
// A thread takes E elements per round and issues ALL their loads before the first use: on these small grids a wave sees
// the full memory latency of every dependent access, and an element at a time cost one round trip each (measured: the
// fused launches took 38 us instead of 13).
// TH, TW are compile-time: the index arithmetic is four integer divisions per element, ~35 instructions each by a run-time divisor.
template <int NWF, int E, int TH, int TW>
__device__ __forceinline__ void fuse_fill_tile(const FuseB& f, int b, int Hh, int Ww, int y0, int x0, float* tile, int wf, bool writer)
{
#ifndef SMVS_FUSE_ABLATE
#define SMVS_FUSE_ABLATE 0                    // profiling builds only (wrong results): 1 no fill at all, 2 no operand loads, 4 no gate arithmetic, 8 no statistics
#endif
    if (SMVS_FUSE_ABLATE & 1) return;
    const int HC = f.HC, HW = Hh * Ww;
    constexpr int per = TH * TW;
    const int tot = HC * per;
    const int lane = threadIdx.x & 63;
    const bool comb = f.mode == FUSE_COMBINE;
    const float* gsrc = f.gates + ((size_t)b * 2 * HC + (comb ? HC : 0)) * HW;     // update half (combine) / reset half (apply)
    const float* csrc = comb ? f.cand + (size_t)b * HC * HW : f.h;
    const float* hsrc = f.h + (size_t)b * HC * HW;
    float mg = 0.0f, sg = 0.0f, mo = 0.0f, so = 0.0f;
    bool have = false;
    const int rounds = (tot + NWF * 64 * E - 1) / (NWF * 64 * E);           // wave-uniform trip count (the statistics use wave shuffles)
    for (int it = 0; it < rounds; ++it) {
        const int base = wf * 64 + lane + it * NWF * 64 * E;
        int c[E], p[E];
        bool in[E];
        float vg[E], vh[E], vc[E];
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const int i = base + u * NWF * 64;
            const int ii = i < tot ? i : 0;
            const int cc = ii / per, r = ii - cc * per, ty = r / TW, tx = r - ty * TW;
            const int y = y0 + ty, x = x0 + tx;
            in[u] = i < tot && y >= 0 && y < Hh && x >= 0 && x < Ww;
            c[u] = cc;
            p[u] = in[u] ? y * Ww + x : 0;
            const size_t e = (size_t)cc * HW + p[u];
            if (SMVS_FUSE_ABLATE & 2) { vg[u] = 0.1f * (float)u; vh[u] = 0.5f; vc[u] = 0.25f; }
            else { vg[u] = gsrc[e]; vh[u] = hsrc[e]; vc[u] = comb ? csrc[e] : 0.0f; }
        }
        if (!have && !(SMVS_FUSE_ABLATE & 8)) {                // statistics: loaded behind the first round's operands
            const double n = (double)HC * HW;
            fuse_gn_coeffs(f.stats_g + ((size_t)b * 2 + (comb ? 1 : 0)) * f.nslot * 2, n, mg, sg);
            if (comb) fuse_gn_coeffs(f.stats_o + (size_t)b * f.nslot * 2, n, mo, so);
            have = true;
        }
#pragma unroll
        for (int u = 0; u < E; ++u) {
            const int i = base + u * NWF * 64;
            const int cc = c[u];
            float v;
            const float g = (SMVS_FUSE_ABLATE & 4) ? vg[u] : fuse_sigmoid(fmaf((vg[u] - mg) * sg, f.gw[cc], f.gb[cc]));
            if (!comb) v = g * vh[u];
            else {
                const float yv = (SMVS_FUSE_ABLATE & 4) ? vc[u] : fuse_tanh(fmaf((vc[u] - mo) * so, f.ow[cc], f.ob[cc]));
                v = g * vh[u] + (1.0f - g) * yv;
                if (writer && in[u]) {
                    const int r = i - cc * per, ty = r / TW, tx = r - ty * TW;
                    if (ty >= 1 && ty < TH - 1 && tx >= 1 && tx < TW - 1) {
                        const size_t e = ((size_t)b * HC + cc) * HW + p[u];
                        f.h_out[e] = v; f.hsnap[e] = v;
                    }
                }
            }
            if (i < tot) tile[i] = in[u] ? v : 0.0f;
        }
    }
}

struct MfmaConvArgs {
    const float* inA; int CA;                // first CA input channels (must be even)
    const float* inB; int CB;                // next CB channels (concat input), or null/0
    const float* w;                          // packed [ (CA+CB)/2 ][ TAPS ][ NT ][ 2 ][ 32 ]
    const float* bias;                       // (Cout) or null
    const float* scale; const float* shift;  // (Cout) BatchNorm inference affine, or null
    const float* skip;                       // same shape as out, added after ReLU, or null
    float* out;
    double* stats; int ngroups; int nslot;   // GroupNorm partial sums (see red.hip), or null
    int Cout, relu, stride;
    int Di, Hi, Wi, Do, Ho, Wo;              // 2-D: Di = Do = 1
    float scaleA;                            // multiplies the A-tensor inputs (-1 feeds -cost)
    int ntiles;                              // tiles of 32 positions in the launch (KS < NW variants: decode / bound of the flat unit index)
    FuseB fuse;                              // 9-tap kernels of the RED plane loop: the B operand computed into LDS (mode != FUSE_NONE)
    // Partial convolutions over a SLICE of a layer's input channels (RED plane loop: the state-independent half of the ConvGRU
    // convolutions runs ahead of the recurrent chain, red.hip): the layer's packed weights hold wcipN channel pairs (0 = (CA+CB)/2),
    // this launch uses pairs wcip0 ..; `init` (same shape as out, or null) is added to the sums before statistics / activation.
    const float* init; int wcip0, wcipN;
};
constexpr int MFMA_FUSE_TW = 34, MFMA_FUSE_TH = 3;           // LDS tile of one unit's B operand: 3 rows x (32 + 2) columns per channel

// W pack kernel: src (Cout, Cin, TAPS) [conv] -> dst [cip][p][nt][h][32]
// step p of channel pair cip covers kk = 2p + h in the 2*TAPS-long (ci0 taps..., ci1 taps...) list.
// adjoint = 1: src is an nn.Conv2d / nn.Conv3d weight (cin, cout, 3, 3[, 3]) -- of the layer whose ADJOINT (gradient with respect to its input:
// the correlation of the output gradient with the transposed, tap-flipped weights) is packed: W'[co][ci][t] = src[ci][co][8 - t].
static __global__ void mfma_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int cin, int cout, int taps, int adjoint)
{
    const int nt = (cout + 31) / 32;
    if (taps == 9) {                             // [cip][nt][h][32][12]: step p = tap p, k-slot h = channel 2*cip + h
        const int n9 = (cin / 2) * nt * 64 * MFMA9_WSLOT;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n9; i += gridDim.x * blockDim.x) {
            const int t = i % MFMA9_WSLOT, j = (i / MFMA9_WSLOT) & 31, h = (i / (MFMA9_WSLOT * 32)) & 1;
            const int tl = (i / (MFMA9_WSLOT * 64)) % nt, cip = i / (MFMA9_WSLOT * 64 * nt);
            const int co = tl * 32 + j, ci = 2 * cip + h;
            dst[i] = (t < 9 && co < cout) ? (adjoint ? src[((size_t)ci * cout + co) * 9 + (8 - t)] : src[((size_t)co * cin + ci) * 9 + t]) : 0.0f;
        }
        return;
    }
    const int n = (cin / 2) * taps * nt * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = i & 31, h = (i >> 5) & 1, t = (i >> 6) % nt, p = (i / (64 * nt)) % taps, cip = i / (64 * nt * taps);
        const int kk = 2 * p + h;
        const int ci = 2 * cip + (kk >= taps ? 1 : 0), tap = kk >= taps ? kk - taps : kk;
        const int co = t * 32 + j;
        dst[i] = co < cout ? (adjoint ? src[((size_t)ci * cout + co) * taps + (taps - 1 - tap)] : src[((size_t)co * cin + ci) * taps + tap]) : 0.0f;
    }
}

#ifdef SMVS_TIMING
// profiling builds only: per-wave phase stamps (shader clocks) of the 9-tap MFMA body, keyed by Cout/32
// [k][0] waves, [1] entry -> first operands landed, [2] K loop, [3] LDS reduce (wave 0), [4] epilogue (wave 0), [5] whole wave 0
__device__ unsigned long long smvs_mfma_timing[8][8];
__device__ __forceinline__ unsigned long long mfma_now() { unsigned long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return t; }
#define SMVS_MT(...) __VA_ARGS__
#else
#define SMVS_MT(...)
#endif

// NT = cout tiles per workgroup (blockIdx.y walks the rest), NW = waves per workgroup, KS = waves that split the K range of
// one (tile, cout tile) unit and reduce through LDS.
// KS = NW (default; latency regime: a few hundred tiles on 256 CUs): one unit per workgroup, (bx, by) = (tile, cout tile block).
// KS < NW (throughput regime, round 3): NW/KS units per workgroup, unit = bx * NW/KS + wave / KS of a flat
//   (cout tile block, tile) index.  With K split four ways a 32-channel layer of the full-resolution stage is 36 MFMAs per
//   wave -- all prologue -- and three of four waves idle through the epilogue (MFMA pipe 20 % busy,
//   profiles/r03_mfma_utilisation.txt); KS = 1 / 2 gives 144-288 MFMAs per wave at 4-5 resident waves per SIMD.
// smem: (NW - NW/KS)*NT*16*64 floats.
// FUSE (9 taps only): the B operand is a.fuse's element-wise stage, computed into `tile` (LDS, NW/KS tiles of
// CB * MFMA_FUSE_TH * MFMA_FUSE_TW floats) before the K loop.
template <int TAPS, int NT, int NW, int KS = NW, bool FUSE = false>
__device__ __forceinline__ void mfma_conv_body(const MfmaConvArgs& a, int bx, int by, float* smem, float* tile = nullptr)
{
    static_assert(!FUSE || TAPS == 9, "fused element-wise stages exist for the 2-D kernels only");
    static_assert(NW % KS == 0, "waves per unit");
    constexpr int TW = NW / KS;                            // units per workgroup
    constexpr int KD = TAPS == 27 ? 3 : 1;
    SMVS_MT(const unsigned long long mt0 = mfma_now(); unsigned long long mt1 = mt0;)
    float (*red)[NT * 16][64] = (float (*)[NT * 16][64])smem;   // partial accumulators of waves 1..NW-1
    const int nt_all = (a.Cout + 31) / 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kw = KS == NW ? wave : wave % KS, tw = KS == NW ? 0 : wave / KS;   // K share, unit inside the workgroup
    const int unit = bx * TW + tw;
    const bool unit_ok = KS == NW || unit < a.ntiles * (nt_all / NT);          // the last workgroup may hold fewer units
    const int nt0 = (KS == NW ? by : unit / a.ntiles) * NT;
    const int j = lane & 31, h = lane >> 5;
    // tile -> (b, od, oy, x0)
    const int xt = (a.Wo + 31) / 32;
    int t = KS == NW ? bx : unit % a.ntiles;
    const int x0 = (t % xt) * 32; t /= xt;
    const int oy = t % a.Ho; t /= a.Ho;
    const int od = t % a.Do;
    const int b = t / a.Do;
    const int ox = x0 + j;
    const bool pos_ok = unit_ok && ox < a.Wo;
    const int HWi = a.Hi * a.Wi;
    const size_t vol_i = (size_t)a.Di * HWi;
    const int Cin = a.CA + a.CB;

    // per-lane offsets in consumption order: step p -> kk = 2p + h -> (channel of the pair, tap)
    uint32_t off[TAPS];
#pragma unroll
    for (int p = 0; p < TAPS; ++p) {
        const int kkA = 2 * p, kkB = 2 * p + 1;            // compile-time for each half
        const int tapA = kkA >= TAPS ? kkA - TAPS : kkA, chA = kkA >= TAPS ? 1 : 0;
        const int tapB = kkB >= TAPS ? kkB - TAPS : kkB, chB = kkB >= TAPS ? 1 : 0;
        const int tap = h ? tapB : tapA, ch = h ? chB : chA;
        const int kd = tap / 9 % 3, ky = (TAPS == 27 ? tap % 9 : tap) / 3, kx = tap % 3;
        const int id = od * a.stride - (KD == 3 ? 1 : 0) + (KD == 3 ? kd : 0);
        const int iy = oy * a.stride - 1 + ky, ix = ox * a.stride - 1 + kx;
        const bool in = pos_ok && id >= 0 && id < a.Di && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
        off[p] = in ? (uint32_t)(((size_t)ch * vol_i + (size_t)(id * a.Hi + iy) * a.Wi + ix) * 4) : SMVS_OOB;
    }
    const BufRsrc rA = make_rsrc(a.inA + (size_t)b * a.CA * vol_i, (uint32_t)((size_t)a.CA * vol_i * 4));
    const BufRsrc rB = make_rsrc(a.CB ? a.inB + (size_t)b * a.CB * vol_i : a.inA, (uint32_t)((size_t)a.CB * vol_i * 4));
    const BufRsrc rW = make_rsrc(a.w, (uint32_t)((size_t)(a.wcipN ? a.wcipN : Cin / 2) * (TAPS == 9 ? MFMA9_WSLOT : TAPS) * nt_all * 64 * 4));

    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;

    // K loop, split over the 4 waves by input-channel pairs.  A wave walks its channel pairs in batches
    // of BS steps (BS X loads + NT*BS W loads), double-buffered: the loads of batch q+1 are in flight
    // while the BS*NT MFMAs of batch q issue.  These levels have only a few hundred tiles, so there is
    // no occupancy to hide latency with -- the prefetch is what keeps the matrix pipe fed.
    constexpr int BS = (TAPS == 27) ? 9 : 9;               // steps per batch (TAPS is a multiple of 9)
    constexpr int NB = TAPS / BS;                          // batches per channel pair
    const int ncip = Cin / 2, per = ncip / KS;
    const int q_end = per * NB;
    struct Batch { float x[BS]; float w[NT][BS]; float sx; };
    // explicit variants so that every off[] / register index is a compile-time constant
#define SMVS_LOAD_BATCH(G, BT, Q)                                                                        \
    {                                                                                                    \
        const int cip_ = kw * per + (Q) / NB;                                                            \
        const bool fromA_ = 2 * cip_ < a.CA;                      /* wave-uniform: scalar selects */    \
        const int choff_ = (int)((size_t)(fromA_ ? 2 * cip_ : 2 * cip_ - a.CA) * vol_i * 4);            \
        i32x4 rx_;                                                                                       \
        rx_.x = fromA_ ? rA.v.x : rB.v.x; rx_.y = fromA_ ? rA.v.y : rB.v.y;                              \
        rx_.z = fromA_ ? rA.v.z : rB.v.z; rx_.w = rA.v.w;                                                \
        const float sx_ = fromA_ ? a.scaleA : 1.0f;                                                      \
        _Pragma("unroll") for (int s_ = 0; s_ < BS; ++s_) {                                              \
            BT.x[s_] = llvm_raw_buffer_load_f32(rx_, (int)off[(G) * BS + s_], choff_, 0);                \
            _Pragma("unroll") for (int n_ = 0; n_ < NT; ++n_)                                            \
                BT.w[n_][s_] = llvm_raw_buffer_load_f32(rW.v, lane * 4, (((cip_ + a.wcip0) * TAPS + (G) * BS + s_) * nt_all + nt0 + n_) * 256, 0); \
        }                                                                                                \
        BT.sx = sx_;                                              /* applied at MMA time: no wait on the loads here */ \
    }
#define SMVS_MMA_BATCH(BT)                                                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < BS; ++s_)                                                    \
        _Pragma("unroll") for (int n_ = 0; n_ < NT; ++n_)                                                \
            acc[n_] = __builtin_amdgcn_mfma_f32_32x32x2f32(BT.w[n_][s_], BT.x[s_] * BT.sx, acc[n_], 0, 0, 0);
    static_assert(NB == 1 || NB == 3, "batching assumes 9 or 27 taps");
    if (NB == 3) {
        // per channel pair: batches g = 0,1,2, two register sets used alternately; the loop body covers TWO pairs so that the
        // sets are back in their roles at the back edge (no register rotation: copying a set waits for its loads).  All
        // loads are unconditional (see the 9-tap loop); past the end they fetch the last pair again and the MFMAs are skipped.
        Batch b0, b1;
        const int qe = q_end - 3;                                  // first batch of the last pair
        SMVS_LOAD_BATCH(0, b0, 0)
        for (int q = 0; q < q_end; q += 6) {
            SMVS_LOAD_BATCH(1, b1, q + 1)
            __builtin_amdgcn_sched_barrier(0);
            SMVS_MMA_BATCH(b0)
            __builtin_amdgcn_sched_barrier(0);
            SMVS_LOAD_BATCH(2, b0, q + 2)
            __builtin_amdgcn_sched_barrier(0);
            SMVS_MMA_BATCH(b1)
            __builtin_amdgcn_sched_barrier(0);
            SMVS_LOAD_BATCH(0, b1, min(q + 3, qe))
            __builtin_amdgcn_sched_barrier(0);
            SMVS_MMA_BATCH(b0)
            __builtin_amdgcn_sched_barrier(0);
            const bool second = q + 3 < q_end;                      // wave-uniform
            SMVS_LOAD_BATCH(1, b0, min(q + 4, qe + 1))
            __builtin_amdgcn_sched_barrier(0);
            if (second) SMVS_MMA_BATCH(b1)
            __builtin_amdgcn_sched_barrier(0);
            SMVS_LOAD_BATCH(2, b1, min(q + 5, qe + 2))
            __builtin_amdgcn_sched_barrier(0);
            if (second) SMVS_MMA_BATCH(b0)
            __builtin_amdgcn_sched_barrier(0);
            SMVS_LOAD_BATCH(0, b0, min(q + 6, qe))
            __builtin_amdgcn_sched_barrier(0);
            if (second) SMVS_MMA_BATCH(b1)
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // 9 taps: one batch per channel pair = 3 row loads (X) + 3*NT weight loads, NPF-1 pairs in flight while one is
        // multiplied.  Lane (j, h): position x0 + j, channel 2*cip + h of the pair.
        constexpr int NPF = NT == 1 ? SMVS_MFMA_PREFETCH : 2;
        const int ix0 = ox * a.stride - 1;                       // column of the west tap
        const bool padL = ix0 < 0, padR = ix0 + 2 >= a.Wi;       // west / east tap is zero padding
        uint32_t roff[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * a.stride - 1 + ky;
            const bool in = pos_ok && iy >= 0 && iy < a.Hi;
            roff[ky] = in ? (uint32_t)(((size_t)h * vol_i + (size_t)iy * a.Wi + (padL ? 0 : ix0)) * 4) : SMVS_OOB;
        }
        struct Batch9 { mf32x3 x[3]; mf32x4 w[NT][2]; float w8[NT]; float sx; };   // no dead load components: the compiler would reuse them
                                                                                    // as scratch and then has to wait for the load in flight
        Batch9 bb[NPF];
        const float* tl = nullptr;                                 // this unit's LDS tile of the B operand (FUSE)
        if constexpr (FUSE) {
            tl = tile + (size_t)tw * a.CB * (MFMA_FUSE_TH * MFMA_FUSE_TW);
            if (unit_ok)
                fuse_fill_tile<KS, 16, MFMA_FUSE_TH, MFMA_FUSE_TW>(a.fuse, b, a.Hi, a.Wi, oy - 1, x0 - 1, const_cast<float*>(tl), kw, nt0 == 0);
            if (a.fuse.zero && bx == 0 && by == 0)
                for (int i = threadIdx.x; i < a.fuse.zero_n; i += NW * 64) a.fuse.zero[i] = 0.0;
            __syncthreads();
        }
        // X operand of channel pair cip: from memory (A half; B half when not fused) or from the LDS tile (fused B half)
        auto load9 = [&](Batch9& B, int cip, auto lds_tag) {
            constexpr bool LDS = decltype(lds_tag)::value;
            const bool fromA = 2 * cip < a.CA;                      // wave-uniform: scalar selects
            if constexpr (LDS) {
                const float* tc = tl + (size_t)(2 * cip - a.CA + h) * (MFMA_FUSE_TH * MFMA_FUSE_TW) + j;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) { B.x[ky].x = tc[ky * MFMA_FUSE_TW]; B.x[ky].y = tc[ky * MFMA_FUSE_TW + 1]; B.x[ky].z = tc[ky * MFMA_FUSE_TW + 2]; }
            } else {
                const int choff = (int)((size_t)(fromA ? 2 * cip : 2 * cip - a.CA) * vol_i * 4);
                i32x4 rx;
                rx.x = fromA ? rA.v.x : rB.v.x; rx.y = fromA ? rA.v.y : rB.v.y;
                rx.z = fromA ? rA.v.z : rB.v.z; rx.w = rA.v.w;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) B.x[ky] = llvm_raw_buffer_load_v3f32(rx, (int)roff[ky], choff, 0);
            }
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                const int wo = ((cip + a.wcip0) * nt_all + nt0 + n) * (64 * MFMA9_WSLOT * 4);
                B.w[n][0] = llvm_raw_buffer_load_v4f32(rW.v, lane * (MFMA9_WSLOT * 4), wo, 0);
                B.w[n][1] = llvm_raw_buffer_load_v4f32(rW.v, lane * (MFMA9_WSLOT * 4) + 16, wo, 0);
                B.w8[n] = llvm_raw_buffer_load_f32(rW.v, lane * (MFMA9_WSLOT * 4) + 32, wo, 0);
            }
            B.sx = fromA ? a.scaleA : 1.0f;                         // applied at MMA time: no wait on the loads here
        };
        auto mma9 = [&](const Batch9& B, auto lds_tag) {
            constexpr bool LDS = decltype(lds_tag)::value;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                // loaded columns start at max(ix0, 0): shift right by one where the west tap is padding, drop the east one
                // (the LDS tile carries its zero padding itself)
                const float l0 = B.x[ky].x, l1 = B.x[ky].y, l2 = B.x[ky].z;
                float xv[3];
                xv[0] = LDS ? l0 : (padL ? 0.0f : l0);
                xv[1] = LDS ? l1 : (padL ? l0 : l1);
                xv[2] = LDS ? l2 : (padR ? 0.0f : (padL ? l1 : l2));
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int p = ky * 3 + kx;
#pragma unroll
                    for (int n = 0; n < NT; ++n)
                        acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(p < 8 ? B.w[n][(p >> 2) & 1][p & 3] : B.w8[n], xv[kx] * B.sx, acc[n], 0, 0, 0);
                }
            }
        };
        // Every load is UNCONDITIONAL (past the end: the last pair again, unused): a load under a branch makes the number
        // of outstanding operations unknown at the join, and the compiler then waits with vmcnt(0) before every batch of
        // MFMAs -- i.e. also for the batch it has just issued, and the prefetch hides nothing (that was the case until
        // round 2: ~1900 clocks per batch against 576 of MFMA issue).
        // Channel pairs [c_lo, c_hi) of this wave, in order (the accumulation order is the unfused kernel's)
        auto run = [&](int c_lo, int c_hi, auto lds_tag) {
#pragma unroll
            for (int i = 0; i < NPF - 1; ++i) load9(bb[i], min(c_lo + i, c_hi - 1), lds_tag);
            for (int q = c_lo; q < c_hi; q += NPF) {
#pragma unroll
                for (int i = 0; i < NPF; ++i) {
                    load9(bb[(i + NPF - 1) % NPF], min(q + i + NPF - 1, c_hi - 1), lds_tag);
                    __builtin_amdgcn_sched_barrier(0);
                    if (q + i < c_hi) mma9(bb[i], lds_tag);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        const int c_lo = kw * per, c_hi = c_lo + q_end;
        if constexpr (FUSE) {
            const int cA = a.CA / 2;
            if (c_lo < min(c_hi, cA)) run(c_lo, min(c_hi, cA), std::false_type());
            if (max(c_lo, cA) < c_hi) run(max(c_lo, cA), c_hi, std::true_type());
        } else {
            SMVS_MT(asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); mt1 = mfma_now();)
            run(c_lo, c_hi, std::false_type());
        }
    }
#undef SMVS_LOAD_BATCH
#undef SMVS_MMA_BATCH

    // ---- split-K reduction through LDS: waves 1..3 publish, wave 0 sums and finishes -----------------
    SMVS_MT(asm volatile("s_nop 7\n s_nop 7" ::: "memory"); const unsigned long long mt2 = mfma_now();)
    if (FUSE) __syncthreads();                // the B tile shares its LDS with the partial accumulators: every wave is out of its K loop
    if (KS > 1) {
        if (kw > 0) {
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[tw * (KS - 1) + kw - 1][n * 16 + r][lane] = acc[n][r];
        }
        __syncthreads();
        if (kw > 0) return;
#pragma unroll 1
        for (int k = 0; k < KS - 1; ++k)     // one partial at a time: 16*NT loads in flight, not 16*NT*(KS-1)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] += red[tw * (KS - 1) + k][n * 16 + r][lane];
    }
    if (!unit_ok) return;                    // (after the barrier) nothing to finish: the per-channel vectors below are indexed by the unit

    SMVS_MT(const unsigned long long mt3 = mfma_now();)
    // D layout of 32x32 MFMA: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31
    const size_t vol_o = (size_t)a.Do * a.Ho * a.Wo;
    const size_t pos = ((size_t)od * a.Ho + oy) * a.Wo + ox;
    float s1 = 0.0f, s2 = 0.0f, t1 = 0.0f, t2 = 0.0f;     // stats of norm group 0 / 1
    // Every load of the epilogue is issued before the first use: on these small grids a wave sees the full memory
    // latency of each dependent access, and a load -> add -> store chain per register cost 16 round trips (measured:
    // the epilogue took as long as the K loop).  Cout is a multiple of 32 (mfma_conv_ok), so every row of the tile is
    // a real channel and the per-channel vectors are read as aligned float4 (rows r..r+3 are consecutive channels).
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int cb = (nt0 + n) * 32 + 4 * h;                      // channel of register 0; register r: cb + (r&3) + 8*(r>>2)
        float bi[16], sc[16], sh[16], sk[16], ini[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 zero = make_float4(0.0f, 0.0f, 0.0f, 0.0f), one = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
            const float4 tb = a.bias ? *reinterpret_cast<const float4*>(a.bias + cb + 8 * g) : zero;
            const float4 ts = a.scale ? *reinterpret_cast<const float4*>(a.scale + cb + 8 * g) : one;
            const float4 th = a.scale ? *reinterpret_cast<const float4*>(a.shift + cb + 8 * g) : zero;
            bi[4 * g] = tb.x; bi[4 * g + 1] = tb.y; bi[4 * g + 2] = tb.z; bi[4 * g + 3] = tb.w;
            sc[4 * g] = ts.x; sc[4 * g + 1] = ts.y; sc[4 * g + 2] = ts.z; sc[4 * g + 3] = ts.w;
            sh[4 * g] = th.x; sh[4 * g + 1] = th.y; sh[4 * g + 2] = th.z; sh[4 * g + 3] = th.w;
        }
        const size_t o0 = ((size_t)b * a.Cout + cb) * vol_o + pos;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            sk[r] = (a.skip && pos_ok) ? a.skip[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * vol_o] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            ini[r] = (a.init && pos_ok) ? a.init[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * vol_o] : 0.0f;
        if (pos_ok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb + (r & 3) + 8 * (r >> 2);
                float v = acc[n][r];
                if (a.init) v += ini[r];
                if (a.bias) v += bi[r];
                if (a.scale) v = fmaf(v, sc[r], sh[r]);
                if (a.stats) {
                    if (a.ngroups == 2 && co >= a.Cout / 2) { t1 += v; t2 = fmaf(v, v, t2); }
                    else { s1 += v; s2 = fmaf(v, v, s2); }
                }
                if (a.relu) v = fmaxf(v, 0.0f);
                if (a.skip) v = sk[r] + v;
                a.out[o0 + (size_t)((r & 3) + 8 * (r >> 2)) * vol_o] = v;
            }
        }
    }
    if (a.stats) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64);
            t1 += __shfl_xor(t1, m, 64); t2 += __shfl_xor(t2, m, 64);
        }
        if (lane == 0) {
            const int slot = (KS == NW ? bx : unit) % a.nslot;
            double* st = a.stats + (((size_t)b * a.ngroups + 0) * a.nslot + slot) * 2;
            atomicAdd(st, (double)s1);
            atomicAdd(st + 1, (double)s2);
            if (a.ngroups == 2) {
                double* su = a.stats + (((size_t)b * a.ngroups + 1) * a.nslot + slot) * 2;
                atomicAdd(su, (double)t1);
                atomicAdd(su + 1, (double)t2);
            }
        }
    }
    SMVS_MT(if (TAPS == 9 && lane == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long mt4 = mfma_now();
        unsigned long long* T = smvs_mfma_timing[(a.Cout / 32) & 7];
        atomicAdd(&T[0], 1ull); atomicAdd(&T[1], mt1 - mt0); atomicAdd(&T[2], mt2 - mt1); atomicAdd(&T[3], mt3 - mt2);
        atomicAdd(&T[4], mt4 - mt3); atomicAdd(&T[5], mt4 - mt0);
    })
}

template <int TAPS, int NT, int NW, int KS = NW>
__global__ __launch_bounds__(NW * 64)
void mfma_conv_kernel(const MfmaConvArgs a)
{
    __shared__ float smem[KS > 1 ? (NW - NW / KS) * NT * 16 * 64 : 1];
    mfma_conv_body<TAPS, NT, NW, KS>(a, blockIdx.x, blockIdx.y, smem);
}

// true if the MFMA kernel serves this layer
inline bool mfma_conv_ok(int CA, int CB, int Cout)
{
    const int Cin = CA + CB;
    return (Cout == 32 || Cout == 64 || Cout == 128) && Cin % 8 == 0 && CA % 2 == 0 && Cin >= 8;
}

inline size_t mfma_packed_floats(int cin, int cout, int taps) { return (size_t)(cin / 2) * (taps == 9 ? MFMA9_WSLOT : taps) * ((cout + 31) / 32) * 64; }

template <int TAPS>
inline void mfma_conv_launch(const MfmaConvArgs& a, int B, hipStream_t st, int Bh = 0)   // Bh: batch the variant is chosen for (0 = B)
{
    const int tiles = ((a.Wo + 31) / 32) * a.Ho * a.Do * B;
    const int tiles_h = ((a.Wo + 31) / 32) * a.Ho * a.Do * (Bh > 0 ? Bh : B);
    const int nt = a.Cout / 32, ncip = (a.CA + a.CB) / 2;
    // Few tiles (the coarse levels): latency, not throughput, sets the time -- one cout tile per workgroup
    // and as many K-splitting waves as the channel count divides into.  Many tiles: one workgroup carries
    // every cout tile so the X operand is loaded once.
    static const int small = tune_int("SMVS_MFMA_SMALL", 1024);
    static const int maxw = tune_int("SMVS_MFMA_WAVES", 8);
    if (tiles_h < small) {
        if (ncip % 16 == 0 && maxw >= 16)     hipLaunchKernelGGL((mfma_conv_kernel<TAPS, 1, 16>), dim3(tiles, nt), dim3(1024), 0, st, a);
        else if (ncip % 8 == 0 && maxw >= 8) hipLaunchKernelGGL((mfma_conv_kernel<TAPS, 1, 8>), dim3(tiles, nt), dim3(512), 0, st, a);
        else                    hipLaunchKernelGGL((mfma_conv_kernel<TAPS, 1, 4>), dim3(tiles, nt), dim3(256), 0, st, a);
    } else if (nt == 1) hipLaunchKernelGGL((mfma_conv_kernel<TAPS, 1, 4>), dim3(tiles), dim3(256), 0, st, a);
    else if (nt == 2)   hipLaunchKernelGGL((mfma_conv_kernel<TAPS, 2, 4>), dim3(tiles), dim3(256), 0, st, a);
    else                hipLaunchKernelGGL((mfma_conv_kernel<TAPS, 4, 4>), dim3(tiles), dim3(256), 0, st, a);
}

}  // namespace smvs
