// mfma_conv.h -- float32 MFMA implicit-GEMM convolution shared by the regularisers (red.hip 2-D, costreg.hip 3-D).
//
// For the levels whose channel counts reach the matrix-core tile (Cout >= 32) the convolution is the GEMM
//   D[cout][pos] = sum_kk W[cout][kk] * X[kk][pos],   kk = (input channel, tap),
// on v_mfma_f32_32x32x2_f32 (exact float32: the result is a k-ordered fmaf chain, so numerics stay in the
// same class as the direct kernels).  Rows (M) are 32 output channels, columns (N) 32 consecutive output
// positions along x -- so every accumulator register of a lane belongs to ONE output position and the store
// of a register across lanes is a coalesced 128-B row -- and K advances two (channel, tap) pairs per MFMA.
//   * X operand: one buffer load per lane and MFMA (lane -> position l&31, k-slot l>>5); the per-lane tap
//     offsets (zero padding = out-of-range offset) are fixed per tile, the channel pair rides in the scalar
//     offset: no vector ALU work in the K loop.
//   * W operand: one coalesced load per lane and MFMA from a buffer pre-packed in exactly the order the
//     lanes consume it ([channel pair][step][cout tile][k-slot][32]).
//   * The coarse levels are tiny (a few thousand positions), so a workgroup's 4 waves split K (input
//     channels) four ways and reduce through LDS: 4x the waves in flight for the same tile.
// Epilogue (wave 0): bias / BatchNorm scale-shift, ReLU, skip add, GroupNorm statistics.
#pragma once
#include "smvs_device.h"

namespace smvs {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef SMVS_MFMA_PREFETCH
#define SMVS_MFMA_PREFETCH 2                 // 9-tap kernels, one cout tile per workgroup: channel pairs in flight per wave
#endif

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
};

// W pack kernel: src (Cout, Cin, TAPS) [conv] -> dst [cip][p][nt][h][32]
// step p of channel pair cip covers kk = 2p + h in the 2*TAPS-long (ci0 taps..., ci1 taps...) list.
static __global__ void mfma_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int cin, int cout, int taps)
{
    const int nt = (cout + 31) / 32;
    const int n = (cin / 2) * taps * nt * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int j = i & 31, h = (i >> 5) & 1, t = (i >> 6) % nt, p = (i / (64 * nt)) % taps, cip = i / (64 * nt * taps);
        const int kk = 2 * p + h;
        const int ci = 2 * cip + (kk >= taps ? 1 : 0), tap = kk >= taps ? kk - taps : kk;
        const int co = t * 32 + j;
        dst[i] = co < cout ? src[((size_t)co * cin + ci) * taps + tap] : 0.0f;
    }
}

// NT = cout tiles per workgroup (blockIdx.y walks the rest), NW = waves splitting K.
// (bx, by) = the workgroup's grid coordinates; smem: (NW-1)*NT*16*64 floats.
template <int TAPS, int NT, int NW>
__device__ __forceinline__ void mfma_conv_body(const MfmaConvArgs& a, int bx, int by, float* smem)
{
    constexpr int KD = TAPS == 27 ? 3 : 1;
    float (*red)[NT * 16][64] = (float (*)[NT * 16][64])smem;   // partial accumulators of waves 1..NW-1
    const int nt_all = (a.Cout + 31) / 32, nt0 = by * NT;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    // tile -> (b, od, oy, x0)
    const int xt = (a.Wo + 31) / 32;
    int t = bx;
    const int x0 = (t % xt) * 32; t /= xt;
    const int oy = t % a.Ho; t /= a.Ho;
    const int od = t % a.Do;
    const int b = t / a.Do;
    const int ox = x0 + j;
    const bool pos_ok = ox < a.Wo;
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
    const BufRsrc rW = make_rsrc(a.w, (uint32_t)((size_t)(Cin / 2) * TAPS * nt_all * 64 * 4));

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
    const int ncip = Cin / 2, per = ncip / NW;
    const int q_end = per * NB;
    struct Batch { float x[BS]; float w[NT][BS]; float sx; };
    // explicit variants so that every off[] / register index is a compile-time constant
#define SMVS_LOAD_BATCH(G, BT, Q)                                                                        \
    {                                                                                                    \
        const int cip_ = wave * per + (Q) / NB;                                                          \
        const bool fromA_ = 2 * cip_ < a.CA;                      /* wave-uniform: scalar selects */    \
        const int choff_ = (int)((size_t)(fromA_ ? 2 * cip_ : 2 * cip_ - a.CA) * vol_i * 4);            \
        i32x4 rx_;                                                                                       \
        rx_.x = fromA_ ? rA.v.x : rB.v.x; rx_.y = fromA_ ? rA.v.y : rB.v.y;                              \
        rx_.z = fromA_ ? rA.v.z : rB.v.z; rx_.w = rA.v.w;                                                \
        const float sx_ = fromA_ ? a.scaleA : 1.0f;                                                      \
        _Pragma("unroll") for (int s_ = 0; s_ < BS; ++s_) {                                              \
            BT.x[s_] = llvm_raw_buffer_load_f32(rx_, (int)off[(G) * BS + s_], choff_, 0);                \
            _Pragma("unroll") for (int n_ = 0; n_ < NT; ++n_)                                            \
                BT.w[n_][s_] = llvm_raw_buffer_load_f32(rW.v, lane * 4, ((cip_ * TAPS + (G) * BS + s_) * nt_all + nt0 + n_) * 256, 0); \
        }                                                                                                \
        BT.sx = sx_;                                              /* applied at MMA time: no wait on the loads here */ \
    }
#define SMVS_MMA_BATCH(BT)                                                                               \
    _Pragma("unroll") for (int s_ = 0; s_ < BS; ++s_)                                                    \
        _Pragma("unroll") for (int n_ = 0; n_ < NT; ++n_)                                                \
            acc[n_] = __builtin_amdgcn_mfma_f32_32x32x2f32(BT.w[n_][s_], BT.x[s_] * BT.sx, acc[n_], 0, 0, 0);
    static_assert(NB == 1 || NB == 3, "batching assumes 9 or 27 taps");
    if (NB == 3) {
        Batch b0, b1;
        // per channel pair: batches g = 0,1,2; pipeline: [L0] (M0|L1) (M1|L2) (M2|L0') ...
        SMVS_LOAD_BATCH(0, b0, 0)
        for (int q = 0; q < q_end; q += 3) {
            SMVS_LOAD_BATCH(1, b1, q + 1)
            __builtin_amdgcn_sched_barrier(0);
            SMVS_MMA_BATCH(b0)
            __builtin_amdgcn_sched_barrier(0);
            SMVS_LOAD_BATCH(2, b0, q + 2)
            __builtin_amdgcn_sched_barrier(0);
            SMVS_MMA_BATCH(b1)
            __builtin_amdgcn_sched_barrier(0);
            if (q + 3 < q_end) SMVS_LOAD_BATCH(0, b1, q + 3)
            __builtin_amdgcn_sched_barrier(0);
            SMVS_MMA_BATCH(b0)
            __builtin_amdgcn_sched_barrier(0);
            if (q + 3 < q_end) {
                // rotate: the prefetched batch sits in b1, the loop expects it in b0
                b0.sx = b1.sx;
#pragma unroll
                for (int s_ = 0; s_ < BS; ++s_) {
                    b0.x[s_] = b1.x[s_];
#pragma unroll
                    for (int n_ = 0; n_ < NT; ++n_) b0.w[n_][s_] = b1.w[n_][s_];
                }
            }
        }
    } else {
        // 9 taps: one batch per channel pair, NPF-1 pairs in flight while one is multiplied (deeper than 2 measured
        // no faster on the coarse levels -- the MFMA issue time of the wave's K range is the chain -- and slower on the large ones)
        constexpr int NPF = NT == 1 ? SMVS_MFMA_PREFETCH : 2;
        Batch bb[NPF];
#pragma unroll
        for (int i = 0; i < NPF - 1; ++i)
            if (i < q_end) SMVS_LOAD_BATCH(0, bb[i], i)
        for (int q = 0; q < q_end; q += NPF) {
#pragma unroll
            for (int i = 0; i < NPF; ++i) {
                if (q + i + NPF - 1 < q_end) SMVS_LOAD_BATCH(0, bb[(i + NPF - 1) % NPF], q + i + NPF - 1)
                __builtin_amdgcn_sched_barrier(0);
                if (q + i < q_end) SMVS_MMA_BATCH(bb[i])
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#undef SMVS_LOAD_BATCH
#undef SMVS_MMA_BATCH

    // ---- split-K reduction through LDS: waves 1..3 publish, wave 0 sums and finishes -----------------
    if (wave > 0) {
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][n * 16 + r][lane] = acc[n][r];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll 1
    for (int k = 0; k < NW - 1; ++k)         // one partial at a time: 16*NT loads in flight, not 16*NT*(NW-1)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][r] += red[k][n * 16 + r][lane];

    // D layout of 32x32 MFMA: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31
    const size_t vol_o = (size_t)a.Do * a.Ho * a.Wo;
    const size_t pos = ((size_t)od * a.Ho + oy) * a.Wo + ox;
    float s1 = 0.0f, s2 = 0.0f, t1 = 0.0f, t2 = 0.0f;     // stats of norm group 0 / 1
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (nt0 + n) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (co < a.Cout && pos_ok) {
                float v = acc[n][r];
                if (a.bias) v += a.bias[co];
                if (a.scale) v = fmaf(v, a.scale[co], a.shift[co]);
                if (a.stats) {
                    if (a.ngroups == 2 && co >= a.Cout / 2) { t1 += v; t2 = fmaf(v, v, t2); }
                    else { s1 += v; s2 = fmaf(v, v, s2); }
                }
                if (a.relu) v = fmaxf(v, 0.0f);
                const size_t o = ((size_t)b * a.Cout + co) * vol_o + pos;
                if (a.skip) v = a.skip[o] + v;
                a.out[o] = v;
            }
        }
    if (a.stats) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64);
            t1 += __shfl_xor(t1, m, 64); t2 += __shfl_xor(t2, m, 64);
        }
        if (lane == 0) {
            const int slot = bx % a.nslot;
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
}

template <int TAPS, int NT, int NW>
__global__ __launch_bounds__(NW * 64)
void mfma_conv_kernel(const MfmaConvArgs a)
{
    __shared__ float smem[(NW - 1) * NT * 16 * 64];
    mfma_conv_body<TAPS, NT, NW>(a, blockIdx.x, blockIdx.y, smem);
}

// true if the MFMA kernel serves this layer
inline bool mfma_conv_ok(int CA, int CB, int Cout)
{
    const int Cin = CA + CB;
    return (Cout == 32 || Cout == 64 || Cout == 128) && Cin % 8 == 0 && CA % 2 == 0 && Cin >= 8;
}

inline size_t mfma_packed_floats(int cin, int cout, int taps) { return (size_t)(cin / 2) * taps * ((cout + 31) / 32) * 64; }

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
