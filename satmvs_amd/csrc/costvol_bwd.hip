// costvol_bwd.hip -- backward of the fused variance volume (costvol.hip) for training:
// loss.backward() through /root/reference/networks/casred.py:22-53 (train.py:284), i.e. through
//   var = sq/V - (sum/V)^2,  sum = ref + sum_s warped_s,  sq = ref^2 + sum_s warped_s^2
// and through grid_sample w.r.t. the source features (the grid itself is built under no_grad,
// /root/reference/modules/warping.py:322).
//   d var / d warped_s = (2/V) (warped_s - sum/V)      d var / d ref = (2/V) (ref - sum/V)
// Taps are recomputed (float64 RPC chain) instead of saving 2.4 GB of warped volumes; gradients
// are scattered with hardware float32 atomics, like torch's grid_sampler backward.
//
// The scatter is bound by the L2's float atomics (~1 lane-atomic per clock per channel: 270 G/s); round 2 issued 9
// per voxel-channel (1 for the reference gradient + 4 per source tap) = 20 ms at the 3-view 768x384x64 shape.
// Reductions BEFORE the L2 atomics, none of which changes what is summed (only the order, like any atomic scatter):
//   * a lane owns its pixel for a chunk of DCH planes: the reference gradient is summed over the chunk in a
//     register -- one atomic per DCH planes;
//   * BOXED waves.  The taps of a wave's 32 x 2 pixels x DCH planes fall into a small box of each source view (the
//     observation the forward's staging rests on).  When every view's box fits 64 x 8 cells the wave
//       - stages the box of the NEXT channel of every view into LDS with b128 LDS-DMA while it works on this one
//         (two buffers; no register, no wait on the way), and reads its taps from there (2 ds_read2_b32 per tap)
//         instead of 4 gathers from memory;
//       - keeps a PRIVATE gradient box per view in LDS, adds its 4 contributions per tap there and flushes the box
//         once per channel: one L2 atomic per TOUCHED cell (~35 x 5) instead of 4 per tap (512 taps).  The flush
//         reads and clears a cell in one ds_wrxchg; cells that stayed 0 send nothing; it is issued at the top of the
//         NEXT channel so that the atomics drain under that channel's arithmetic.
//     The gradient box is float64 because of the hardware, not the arithmetic: on gfx950 ds_add_f32 retires one
//     lane every 3 clocks (192 clocks per wave instruction) while ds_add_f64 and the integer atomics take 6-8 clocks
//     per instruction (tools/ubench_ldsatomic.hip, profiles/r03_ubench_ldsatomic.txt).  The box therefore sums in
//     double -- more exact than any float32 order -- and is rounded to float32 once, when it is flushed;
//   * waves where some view's box does not fit (large parallax per plane, strong rotation) keep the register scheme
//     of the first half of round 3: where the lane to the east has its north-west cell ON this lane's north-east
//     cell this lane hands its two east contributions over the DPP network (wave_shr:1), and consecutive planes of
//     a lane whose taps fall into the SAME cell are summed in registers and flushed when the cell changes.
// Taps that touch the image border (some corner outside) keep the plain per-corner path; bits of the loss are
// unaffected, gradients differ from round 2 by summation order only (tests: reference-captured gradients,
// tests/golden/grad.npz and train_step.npz, and autograd of the torch composite).
#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

constexpr int MAX_SRC = 7;
constexpr int PX = 32, PY = 2;                     // patch of a wave: 32 x 2 pixels
constexpr int BWD_WAVES = 2;                       // waves of a workgroup, side by side: 64 x 2 pixels (no workgroup barrier anywhere)
constexpr int TILE_X = PX * BWD_WAVES, TILE_Y = PY;

struct CostVolBwdParams {
    const float* grad_var;          // (B,C,D,H,W)
    const float* ref;               // (B,C,H,W)
    const float* src[MAX_SRC];
    float* grad_ref;                // (B,C,H,W), accumulated
    float* grad_src[MAX_SRC];       // (B,C,H,W), accumulated
    const double* geo;
    const float* depth;
    int B, V, C, D, H, W, depth_is_4d;
    int xt, yt, dct, dch;
};

#ifndef SMVS_BWD_OCC
#define SMVS_BWD_OCC 2                 // waves per SIMD the kernel is compiled for
#endif
#ifndef SMVS_BWD_ABLATE
#define SMVS_BWD_ABLATE 0              // timing experiments only (wrong results): 1 no flush atomics, 2 no box adds, 4 no reference atomic, 16 geometry only, 32 box adds as ds_add_u32
#endif
#ifndef SMVS_BWD_DCH8_SRC
#define SMVS_BWD_DCH8_SRC 4           // up to this many source views a lane keeps 8 planes of taps (2 beyond: the register scheme only); measured 3 / 4 views: 4.01 -> 3.66, 6.78 -> 5.88 ms against 4-plane chunks
#endif
#ifndef SMVS_BWD_BOX_AHEAD
#define SMVS_BWD_BOX_AHEAD 0           // 1: feature boxes two channels ahead on three LDS buffers (1-2 source views).  Measured in round 5: 3.91-3.96 against 3.89-3.92 ms (profiles/r05_bwd_prefetch2.txt) -- the boxes' latency is not what the waves wait for; off
#endif
#ifndef SMVS_BWD_FLUSH_TOGETHER
#define SMVS_BWD_FLUSH_TOGETHER 0      // 1: (1-2 source views) the flush exchanges of every view in flight together, one wait.  Measured in round 5: 3.92 against 3.93 ms -- the two LDS round trips per channel are not what the waves wait for; off
#endif
#ifndef SMVS_BWD_KEEP_WEIGHTS
#define SMVS_BWD_KEEP_WEIGHTS 1        // boxed path, 1-2 source views: tap weights kept in registers over the channel loop (A/B switch)
#endif
#ifndef SMVS_BWD_MEAN_MUL
#define SMVS_BWD_MEAN_MUL 1            // interior waves: mean over the views as sum * RN(1/V) instead of the exact division (A/B switch)
#endif
#ifndef SMVS_BWD_LDS
#define SMVS_BWD_LDS 1                 // 0: never take the boxed path (A/B)
#endif
#ifdef SMVS_BWD_TIMING
// profiling builds only (tools/ab_build.sh x -DSMVS_BWD_TIMING, AB_SRC=costvol_bwd.hip): per-wave phase stamps of the boxed path in shader
// clocks, read back through smvs_debug_timing_bwd().  [0] geometry + set-up, [1] top-of-channel wait for the loads, [2] flush of the
// previous channel's boxes, [3] issue of the next loads / boxes, [4] plane loop, [7] waves
__device__ unsigned long long smvs_bwd_timing[8];
__device__ __forceinline__ unsigned long long bnow() { unsigned long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return t; }
#define SMVS_BT(...) __VA_ARGS__
#else
#define SMVS_BT(...)
#endif
// one word per tap: 00 | y0 << 15 | x0 while the geometry runs, then the byte offset the tap's path wants
constexpr uint32_t TAP_DROPPED = 0x80000000u;      // = SMVS_OOB: a load through it returns 0
constexpr uint32_t TAP_PARTIAL = 0xC0000000u;      // | (y0+1) << 15 | (x0+1): some corner lies outside the image

// boxes of one source view, per wave: BOX_H rows of 64 cells -- features float32 (two buffers), gradient float64
constexpr int BOX_W = 64, BOX_H = 8, BOX_CELLS = BOX_H * BOX_W;
constexpr int FBOX_BYTES = BOX_CELLS * 4, GBOX_BYTES = BOX_CELLS * 8;
constexpr int BOX_MAX_SRC = 4;                     // per wave 8 KB per view: 32 KB per workgroup at 2 source views
static_assert(BOX_W == 64 && BOX_H == 8, "DMA slot map and flush are written for 64 x 8");

__device__ __forceinline__ float dpp_from_prev_lane(float v)
{
    // wave_shr:1 -- lane i receives lane i-1's value, lane 0 keeps `old` (0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t dpp_from_prev_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)TAP_DROPPED, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t dpp_from_next_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }   // wave_shl:1

// north pair (nw, ne) and south pair (sw, se) of a tap out of a staged feature box (row pitch 64 dwords)
__device__ __forceinline__ void fbox_read(uint32_t addr, f32x2& north, f32x2& south)
{
    asm volatile("ds_read2_b32 %0, %2 offset1:1\n\t"
                 "ds_read2_b32 %1, %2 offset0:64 offset1:65"
                 : "=&v"(north), "=&v"(south) : "v"(addr) : "memory");
}
__device__ __forceinline__ void fbox_wait(f32x2& north, f32x2& south)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(north), "+v"(south) :: "memory");
}
// counted form: LDS operations of a wave complete in order, so "at most N still in flight" releases everything issued before
// the last N (the straight-line loop body below issues no scalar memory load in between -- those share the counter)
template <int N>
__device__ __forceinline__ void fbox_wait_n(f32x2& north, f32x2& south)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(north), "+v"(south) : "n"(N < 15 ? N : 15) : "memory");
}
// the four corners of a full tap into the gradient box (LDS byte address of the north-west cell); no return value, in order per wave
__device__ __forceinline__ void gbox_add4(uint32_t addr, float c0, float c1, float c2, float c3)
{
#if SMVS_BWD_ABLATE & 32            // timing experiment (wrong results): 32-bit integer adds into the same cells
    const int i0 = (int)c0, i1 = (int)c1, i2 = (int)c2, i3 = (int)c3;
    asm volatile("ds_add_u32 %0, %1\n\t"
                 "ds_add_u32 %0, %2 offset:8\n\t"
                 "ds_add_u32 %0, %3 offset:%5\n\t"
                 "ds_add_u32 %0, %4 offset:%6"
                 :: "v"(addr), "v"(i0), "v"(i1), "v"(i2), "v"(i3), "n"(BOX_W * 8), "n"(BOX_W * 8 + 8) : "memory");
    return;
#endif
    const double d0 = (double)c0, d1 = (double)c1, d2 = (double)c2, d3 = (double)c3;
    asm volatile("ds_add_f64 %0, %1\n\t"
                 "ds_add_f64 %0, %2 offset:8\n\t"
                 "ds_add_f64 %0, %3 offset:%5\n\t"
                 "ds_add_f64 %0, %4 offset:%6"
                 :: "v"(addr), "v"(d0), "v"(d1), "v"(d2), "v"(d3), "n"(BOX_W * 8), "n"(BOX_W * 8 + 8) : "memory");
}
// the same exchange without the wait and without the rounding: the caller issues every view's eight first and waits once (gbox_wait_all)
__device__ __forceinline__ void gbox_take8_nowait(uint32_t addr, double (&d)[BOX_H])
{
    const double zero = 0.0;
    asm volatile("ds_wrxchg_rtn_b64 %0, %8, %9\n\t"
                 "ds_wrxchg_rtn_b64 %1, %8, %9 offset:%10\n\t"
                 "ds_wrxchg_rtn_b64 %2, %8, %9 offset:%11\n\t"
                 "ds_wrxchg_rtn_b64 %3, %8, %9 offset:%12\n\t"
                 "ds_wrxchg_rtn_b64 %4, %8, %9 offset:%13\n\t"
                 "ds_wrxchg_rtn_b64 %5, %8, %9 offset:%14\n\t"
                 "ds_wrxchg_rtn_b64 %6, %8, %9 offset:%15\n\t"
                 "ds_wrxchg_rtn_b64 %7, %8, %9 offset:%16"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
                 : "v"(addr), "v"(zero), "n"(BOX_W * 8), "n"(BOX_W * 16), "n"(BOX_W * 24), "n"(BOX_W * 32),
                   "n"(BOX_W * 40), "n"(BOX_W * 48), "n"(BOX_W * 56)
                 : "memory");
}
template <int NV>
__device__ __forceinline__ void gbox_wait_all(double (&d)[NV][BOX_H])
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < NV; ++s)
#pragma unroll
        for (int i = 0; i < BOX_H; ++i) asm volatile("" : "+v"(d[s][i]));
}
// read and clear the lane's cell of box row I (no wait)
template <int I>
__device__ __forceinline__ void gbox_take_row(uint32_t addr, double& d)
{
    const double zero = 0.0;
    asm volatile("ds_wrxchg_rtn_b64 %0, %1, %2 offset:%3" : "=&v"(d) : "v"(addr), "v"(zero), "n"(I * BOX_W * 8) : "memory");
}
// rows 0 .. NR-1, one wait
template <int NR>
__device__ __forceinline__ void gbox_take_rows(uint32_t addr, double (&d)[BOX_H])
{
    if constexpr (NR > 0) gbox_take_row<0>(addr, d[0]);
    if constexpr (NR > 1) gbox_take_row<1>(addr, d[1]);
    if constexpr (NR > 2) gbox_take_row<2>(addr, d[2]);
    if constexpr (NR > 3) gbox_take_row<3>(addr, d[3]);
    if constexpr (NR > 4) gbox_take_row<4>(addr, d[4]);
    if constexpr (NR > 5) gbox_take_row<5>(addr, d[5]);
    if constexpr (NR > 6) gbox_take_row<6>(addr, d[6]);
    if constexpr (NR > 7) gbox_take_row<7>(addr, d[7]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(d[i]));
}
// read and clear the lane's cell of each of the 8 box rows
__device__ __forceinline__ void gbox_take8(uint32_t addr, float (&v)[BOX_H])
{
    const double zero = 0.0;
    double d[BOX_H];
    asm volatile("ds_wrxchg_rtn_b64 %0, %8, %9\n\t"
                 "ds_wrxchg_rtn_b64 %1, %8, %9 offset:%10\n\t"
                 "ds_wrxchg_rtn_b64 %2, %8, %9 offset:%11\n\t"
                 "ds_wrxchg_rtn_b64 %3, %8, %9 offset:%12\n\t"
                 "ds_wrxchg_rtn_b64 %4, %8, %9 offset:%13\n\t"
                 "ds_wrxchg_rtn_b64 %5, %8, %9 offset:%14\n\t"
                 "ds_wrxchg_rtn_b64 %6, %8, %9 offset:%15\n\t"
                 "ds_wrxchg_rtn_b64 %7, %8, %9 offset:%16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
                 : "v"(addr), "v"(zero), "n"(BOX_W * 8), "n"(BOX_W * 16), "n"(BOX_W * 24), "n"(BOX_W * 32),
                   "n"(BOX_W * 40), "n"(BOX_W * 48), "n"(BOX_W * 56)
                 : "memory");
#pragma unroll
    for (int i = 0; i < BOX_H; ++i) v[i] = (float)d[i];
}

template <int GEO, int NSRC, int DCH>
__global__ __launch_bounds__(64 * BWD_WAVES, SMVS_BWD_OCC)
void costvol_bwd_kernel(const CostVolBwdParams p)
{
    static_assert(DCH * NSRC <= 32, "tap flag masks");
    constexpr bool BOX = SMVS_BWD_LDS && NSRC <= BOX_MAX_SRC;
    // feature boxes: two buffers (the next channel's boxes land while this one's are read) -- three with 1-2 source views, where the
    // boxes are requested TWO channels ahead like the gradient planes (round 5: the first of the eight plane chunks of a tile that asks
    // for a channel's rows takes an HBM miss, ~2-4 us, more than a channel of arithmetic)
    constexpr int NFB = (SMVS_BWD_BOX_AHEAD && NSRC <= 2) ? 3 : 2;
    constexpr int WAVE_LDS = NSRC * (NFB * FBOX_BYTES + GBOX_BYTES);        // [NFB buffers][view] feature boxes, then [view] gradient boxes
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[BOX ? BWD_WAVES * WAVE_LDS : 16];
    uint32_t L = xcd_remap(blockIdx.x, gridDim.x);
    const int xtile = L % p.xt; L /= p.xt;
    const int dchunk = L % p.dct; L /= p.dct;
    const int ytile = L % p.yt;
    const int b = L / p.yt;
    SMVS_BT(const unsigned long long bt_start = bnow(); unsigned long long bt_wait = 0, bt_flush = 0, bt_issue = 0, bt_planes = 0;)
    const int lane = threadIdx.x;                           // blockDim = (64, BWD_WAVES): threadIdx.y = wave
    const int wv_ = threadIdx.y;
    // The wave's pixels: a 32 x 2 patch (the two waves side by side) where the boxed path exists; kernels without it (more
    // than 4 source views) take a 64 x 1 patch (the two waves one above the other): longer runs of east hand-overs and
    // whole-row gathers (measured at the 3-view 768x384x64 shape with every wave on the register scheme: 9.3 ms against
    // 11.9 ms on 32 x 2 -- which is what a wave pays whose boxes do not fit)
    constexpr int pw = BOX ? PX : 64;
    const int lx = lane % pw;
    const int x = xtile * TILE_X + (BOX ? wv_ * PX : 0) + lx;
    const int y = ytile * TILE_Y + (BOX ? lane / PX : wv_);
    const int H = p.H, W = p.W, C = p.C, D = p.D;
    const int HW = H * W;
    const bool active = x < W && y < H;
    const int pix = min(y, H - 1) * W + min(x, W - 1);
    const int d0 = dchunk * DCH, d1 = min(d0 + DCH, D);
    const float half_wm1 = (float)((W - 1) * 0.5), half_hm1 = (float)((H - 1) * 0.5);
    const float fV = (float)p.V, rV = __fdiv_rn(1.0f, fV), two_over_v = 2.0f / fV;

    BufRsrc rs[NSRC];
#pragma unroll
    for (int s = 0; s < NSRC; ++s) rs[s] = make_rsrc(p.src[s] + (size_t)b * C * HW, (uint32_t)C * (uint32_t)HW * 4u);
    // (the boxed path's per-channel traffic -- flush atomics, reference value, gradient planes -- goes through descriptors as well:
    // scalar channel offsets instead of 64-bit address arithmetic per lane, and lanes that have nothing to add carry an out-of-range
    // offset instead of sitting under a branch: round 6, profiles/r06_bwd_phases.txt)

    // the wave's gradient boxes start out clear; every flush leaves them clear again
    const uint32_t wave_lds = BOX ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(lds_addr(lds_all) + (uint32_t)(wv_ * WAVE_LDS))) : 0u;
    const uint32_t gbox_lds = wave_lds + (uint32_t)(NSRC * NFB * FBOX_BYTES);
    if (BOX) {
        double* mine = (double*)(lds_all + wv_ * WAVE_LDS + NSRC * NFB * FBOX_BYTES);
        for (int i = lane; i < NSRC * BOX_CELLS; i += 64) mine[i] = 0.0;
    }

    // ---- A: taps of the chunk's planes (float64 chain), once for all channels ------------------------------------
    uint32_t tt[DCH][NSRC];                                 // see TAP_*: full taps end up as a byte offset (boxed: inside the box; else: inside the plane)
    float tf[DCH][NSRC][2];                                 // x and y fraction of the tap: the four weights are rebuilt per channel (2 registers instead of 4)
    uint32_t take = 0, give = 0;                            // bit d*NSRC+s: fold the west lane's east pair in / hand mine to the east lane
    uint32_t any_partial = 0, any_take = 0, any_hole = 0;   // wave-uniform: some lane of the wave has such a tap (hole: not a full tap)
    uint32_t all_take = 0;                                  // wave-uniform: a 32-lane row of the patch is one run of cells, west to east, in both rows
    bool boxed = BOX;                                       // wave-uniform: every view's box fits
    int box_g0[NSRC];                                       // wave-uniform: element offset of the box's first cell inside an H x W plane
    int box_rows[NSRC];                                     // wave-uniform: rows of the box any full tap of the wave touches (the flush visits no other)
    {
        const cgeo_t geo_b = as_cgeo((GEO == 0) ? p.geo + (size_t)b * p.V * RPC_LEN : p.geo + (size_t)b * (p.V - 1) * 16);
        RpcInv ref_n, src_n[NSRC];
        if (GEO == 0) {
            ref_n = rpc_inv_image(geo_b);       // the forward's reciprocals, bit for bit (recip_scale)
#pragma unroll
            for (int s = 0; s < NSRC; ++s) src_n[s] = rpc_inv_ground(geo_b + (size_t)(s + 1) * RPC_LEN);
        }
        int lo_x[NSRC], hi_x[NSRC], lo_y[NSRC], hi_y[NSRC];  // extent of the lane's full taps, per view
#pragma unroll
        for (int s = 0; s < NSRC; ++s) { lo_x[s] = lo_y[s] = 0x7fffffff; hi_x[s] = hi_y[s] = -0x7fffffff; }
        const double fx = (double)min(x, W - 1), fy = (double)min(y, H - 1);
        // the forward's chain (costvol.hip): image -> ground with the plane-invariant part of the four cubics formed once per
        // pixel, ground -> image for PQ planes per pass so that a view's coefficients are fetched once for all of them
        constexpr int PQ = DCH < 4 ? DCH : 4;
        static_assert(DCH % PQ == 0, "planes per pass");
        float hf[DCH];
        double lat[DCH], lon[DCH];
#pragma unroll
        for (int k = 0; k < DCH; ++k) {
            const int d = min(d0 + k, D - 1);
            hf[k] = p.depth_is_4d ? p.depth[((size_t)b * D + d) * HW + pix] : p.depth[(size_t)b * D + d];
        }
        if (GEO == 0) {
            P2OPix px;
            p2o_pixel(geo_b, ref_n, fx, fy, px);
            p2o_planes<DCH>(launder(geo_b), ref_n, px, hf, lat, lon);
        }
#pragma unroll
        for (int pq = 0; pq < DCH; pq += PQ) {
            const cgeo_t geo_d = launder(geo_b);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                Tap tq[PQ];
                if (GEO == 0) {
                    double samp[PQ], line[PQ], hh[PQ];
#pragma unroll
                    for (int u = 0; u < PQ; ++u) hh[u] = (double)hf[pq + u];
                    o2p_xn<PQ>(geo_d + (size_t)(s + 1) * RPC_LEN, src_n[s], lat + pq, lon + pq, hh, samp, line);
#pragma unroll
                    for (int u = 0; u < PQ; ++u) tq[u] = tap_from_pixel((float)samp[u], (float)line[u], H, W, half_wm1, half_hm1);
                } else {
                    const cgeo_t P = geo_d + s * 16;
#pragma unroll
                    for (int u = 0; u < PQ; ++u) {
                        const double h = (double)hf[pq + u];
                        const double rx = fma(P[1], fy, P[0] * fx) + P[2];
                        const double ry = fma(P[5], fy, P[4] * fx) + P[6];
                        const double rz = fma(P[9], fy, P[8] * fx) + P[10];
                        const double X = fma(rx, h, P[3]), Y = fma(ry, h, P[7]), Z = fma(rz, h, P[11]);
                        tq[u] = tap_from_grid((float)((X / Z) / ((W - 1) * 0.5) - 1.0), (float)((Y / Z) / ((H - 1) * 0.5) - 1.0), H, W);
                    }
                }
#pragma unroll
                for (int u = 0; u < PQ; ++u) {
                    const int k = pq + u;
                    const Tap& t = tq[u];
                    tf[k][s][0] = t.fw; tf[k][s][1] = t.fn;
                    const bool in = active && d0 + k < d1;
                    const bool v0 = t.o_nw != SMVS_OOB, v1 = t.o_ne != SMVS_OOB, v2 = t.o_sw != SMVS_OOB, v3 = t.o_se != SMVS_OOB;
                    uint32_t e = TAP_DROPPED;
                    if (in && v0 && v1 && v2 && v3) e = ((uint32_t)t.y0 << 15) | (uint32_t)t.x0;            // 0 <= y0 <= H-2, 0 <= x0 <= W-2
                    else if (in && (v0 || v1 || v2 || v3)) {
                        e = TAP_PARTIAL | ((uint32_t)(t.y0 + 1) << 15) | (uint32_t)(t.x0 + 1);   // -1 <= y0 <= H-1, -1 <= x0 <= W-1
                    }
                    tt[k][s] = e;
                    const uint32_t bit = 1u << (k * NSRC + s);
                    const bool full = (e & TAP_DROPPED) == 0;
                    if (full) {
                        lo_x[s] = min(lo_x[s], t.x0); hi_x[s] = max(hi_x[s], t.x0);
                        lo_y[s] = min(lo_y[s], t.y0); hi_y[s] = max(hi_y[s], t.y0);
                    }
                    const uint32_t pe = dpp_from_prev_lane(e);
                    const bool tk = full && lx > 0 && (pe & TAP_DROPPED) == 0 && pe + 1u == e;     // same row, one cell to the west
                    const uint32_t nt = dpp_from_next_lane(tk ? 1u : 0u);
                    const bool gv = lx < pw - 1 && nt;
                    if (tk) take |= bit;
                    if (gv) give |= bit;
                    if (__builtin_amdgcn_ballot_w64((e & TAP_PARTIAL) == TAP_PARTIAL) != 0) any_partial |= bit;
                    if (__builtin_amdgcn_ballot_w64(!full) != 0) any_hole |= bit;
                    if (__builtin_amdgcn_ballot_w64(tk) != 0) any_take |= bit;
                    if (BOX && __builtin_amdgcn_ballot_w64(tk) == 0xfffffffefffffffeull) all_take |= bit;      // every lane but the two row starts
                }
                __builtin_amdgcn_sched_barrier(0);          // one view at a time: its 80 coefficients leave the SGPRs before the next view's arrive
            }
        }
        // does every view's box fit?  (extent of the north-west cells + 1 in both directions; a view without a full tap needs none)
        int org_x[NSRC], org_y[NSRC];
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            int ax = lo_x[s], bx = hi_x[s], ay = lo_y[s], by = hi_y[s];
            if (BOX) wave_minmax4(ax, bx, ay, by);
            const bool none = bx < ax;
            if (none) { ax = 0; ay = 0; }
            else if (bx - ax + 2 > BOX_W || by - ay + 2 > BOX_H) boxed = false;
            org_x[s] = ax; org_y[s] = ay;
            box_rows[s] = __builtin_amdgcn_readfirstlane(none ? 0 : min(by - ay + 2, BOX_H));
        }
        // full taps: packed cell -> byte offset inside the box (boxed) or inside the H x W plane
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            box_g0[s] = boxed ? org_y[s] * W + org_x[s] : 0;
            const int pitch = boxed ? BOX_W : W;
            const int org = boxed ? org_y[s] * BOX_W + org_x[s] : 0;
#pragma unroll
            for (int k = 0; k < DCH; ++k) {
                const uint32_t e = tt[k][s];
                if ((e & TAP_DROPPED) == 0) tt[k][s] = (uint32_t)(((int)(e >> 15) * pitch + (int)(e & 0x7fffu) - org) * 4);
            }
        }
    }
    any_partial = __builtin_amdgcn_readfirstlane(any_partial);
    any_take = __builtin_amdgcn_readfirstlane(any_take);
    any_hole = __builtin_amdgcn_readfirstlane(any_hole);
    all_take = __builtin_amdgcn_readfirstlane(all_take);

    // ---- B: channels ---------------------------------------------------------------------------------------------
    const float* refp = p.ref + (size_t)b * C * HW + pix;
    float* grefp = p.grad_ref + (size_t)b * C * HW + pix;
    const float* gp = p.grad_var + (((size_t)b * C) * D + d0) * HW + pix;
    const int W4 = W * 4;

    // per-corner byte offsets of a tap that is not full (border of the image): decoded on the spot, rare
    auto corner_off = [&](uint32_t e, int k) -> uint32_t {
        if ((e & TAP_PARTIAL) != TAP_PARTIAL) return (e & TAP_DROPPED) ? SMVS_OOB : e + (uint32_t)((k & 1) * 4 + (k >> 1) * W4);
        const int y0 = (int)((e >> 15) & 0x7fffu) - 1 + (k >> 1), x0 = (int)(e & 0x7fffu) - 1 + (k & 1);
        return ((uint32_t)y0 < (uint32_t)H && (uint32_t)x0 < (uint32_t)W) ? (uint32_t)(y0 * W + x0) * 4u : SMVS_OOB;
    };
    // the four weights, as tap_from_grid forms them
    auto weights = [&](int k, int s, float (&w4)[4]) {
        asm volatile("" : "+v"(tf[k][s][0]), "+v"(tf[k][s][1]));      // opaque per use: else all 4 * DCH * NSRC products are hoisted out of the channel loop
        const float w = tf[k][s][0], n = tf[k][s][1], ee = 1.0f - w, ss = 1.0f - n;
        w4[0] = ss * ee; w4[1] = ss * w; w4[2] = n * ee; w4[3] = n * w;
    };
    // a border tap's contributions: per corner, straight to memory
    auto scatter_partial = [&](uint32_t e, int s, int c, float c0, float c1, float c2, float c3) {
        float* plane = p.grad_src[s] + ((size_t)b * C + c) * HW;
        const float cc[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t o = corner_off(e, q);
            if (o != SMVS_OOB) unsafeAtomicAdd(plane + (o >> 2), cc[q]);
        }
    };

    // the next channel's reference value and gradient planes are requested a whole iteration ahead in both loops: they are
    // bound by memory latency (SQ_WAIT_ANY 71 % of the wave cycles without it), not by any unit
    float r_next = refp[0], g_next[DCH];
#pragma unroll
    for (int k = 0; k < DCH; ++k) g_next[k] = gp[(size_t)min(k, d1 - d0 - 1) * HW];

    if (SMVS_BWD_ABLATE & 16) { if (tt[0][0] == 0x12345u) p.grad_ref[lane] = tf[0][0][0] + (float)take + (float)give + (float)box_g0[0]; return; }
    if (BOX && boxed) {
        // ================================ boxed waves ==========================================================
        // DMA slot map of a feature box: lane l of instruction j lays down cells (row 4j + l/16, columns 4(l%16) .. +3); cells
        // beyond the image or the tensor hold other rows' data or zeros -- no full tap reads them
        // 1-2 source views: the four weights of every tap stay in registers over the channel loop ({nw, ne}, {sw, se}: 4 per tap instead of
        // the 2 fractions) -- rebuilding them per channel was 10 of a plane's 44 VALU instructions, and the plane loop is bound by VALU
        // issue + the LDS pipe at 2 waves per SIMD (round 6, profiles/r06_bwd_phases.txt); with more views the registers are not there.
        constexpr bool KEEPW = SMVS_BWD_KEEP_WEIGHTS && NSRC <= 2;
        f32x2 kwn[KEEPW ? DCH : 1][NSRC], kws[KEEPW ? DCH : 1][NSRC];
        if constexpr (KEEPW) {
#pragma unroll
            for (int k = 0; k < DCH; ++k)
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
                    const float w = tf[k][s][0], n = tf[k][s][1];
                    const f32x2 ew = {1.0f - w, w};
                    kwn[k][s] = ew * (1.0f - n);
                    kws[k][s] = ew * n;
                }
        }
        // ... and so does the tap's address in the gradient box (interior waves; one v_lshl_add per tap, plane and channel otherwise)
        uint32_t tga[KEEPW ? DCH : 1][NSRC];
        if constexpr (KEEPW) {
#pragma unroll
            for (int k = 0; k < DCH; ++k)
#pragma unroll
                for (int s = 0; s < NSRC; ++s) tga[k][s] = gbox_lds + (uint32_t)(s * GBOX_BYTES) + 2u * tt[k][s];
        }
        auto tap_weights = [&](int k, int s, f32x2& wn, f32x2& ws) __attribute__((always_inline)) {
            if constexpr (KEEPW) { wn = kwn[k][s]; ws = kws[k][s]; }
            else {
                asm volatile("" : "+v"(tf[k][s][0]), "+v"(tf[k][s][1]));      // opaque per use: else every weight is hoisted out of the channel loop
                const float w = tf[k][s][0], n = tf[k][s][1];
                const f32x2 ew = {1.0f - w, w};
                wn = ew * (1.0f - n);
                ws = ew * n;
            }
        };
        BufRsrc rgs[NSRC];
#pragma unroll
        for (int s = 0; s < NSRC; ++s) rgs[s] = make_rsrc(p.grad_src[s] + (size_t)b * C * HW, (uint32_t)C * (uint32_t)HW * 4u);
        const BufRsrc rref = make_rsrc(p.ref + (size_t)b * C * HW, (uint32_t)C * (uint32_t)HW * 4u);
        const BufRsrc rgref = make_rsrc(p.grad_ref + (size_t)b * C * HW, (uint32_t)C * (uint32_t)HW * 4u);
        const uint32_t pix4 = (uint32_t)pix * 4u;
        // the gradient planes of channel cc for this chunk: one descriptor per channel (64-bit scalar arithmetic once), plane k in the scalar offset
        auto load_planes = [&](int cc, float (&dst)[DCH]) {
            const BufRsrc rv = make_rsrc(p.grad_var + (((size_t)b * C + cc) * D + d0) * HW, (uint32_t)(d1 - d0) * (uint32_t)HW * 4u);
#pragma unroll
            for (int k = 0; k < DCH; ++k) dst[k] = llvm_raw_buffer_load_f32(rv.v, (int)pix4, min(k, d1 - d0 - 1) * HW * 4, 0);
        };
        uint32_t dvo[NSRC], gvo[NSRC];                      // (gvo: the lane's cell of box row 0 inside one H x W gradient plane, bytes)
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            dvo[s] = (uint32_t)((box_g0[s] + (lane >> 4) * W + (lane & 15) * 4) * 4);
            gvo[s] = (uint32_t)((box_g0[s] + lane) * 4);
        }
        auto stage = [&](int cn, int par) {
            const int so = __builtin_amdgcn_readfirstlane(cn * HW * 4);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const uint32_t fb = wave_lds + (uint32_t)((par * NSRC + s) * FBOX_BYTES);
                dma_x4_to_lds_at<0>(rs[s], fb, dvo[s], so);
                dma_x4_to_lds_at<FBOX_BYTES / 2>(rs[s], fb, dvo[s], so + 4 * W4);
            }
        };
        auto flush_boxes = [&](int c) {                     // one atomic per touched cell: row i of the box, column = lane
            if constexpr (SMVS_BWD_FLUSH_TOGETHER && NSRC <= 2) {
                // every view's exchanges first, ONE wait (round 5: per view the wave sat through a full LDS round trip, twice per channel)
                double d[NSRC][BOX_H];
#pragma unroll
                for (int s = 0; s < NSRC; ++s) gbox_take8_nowait(gbox_lds + (uint32_t)(s * GBOX_BYTES) + (uint32_t)lane * 8u, d[s]);
                gbox_wait_all<NSRC>(d);
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
                    float* q = p.grad_src[s] + ((size_t)b * C + c) * HW + box_g0[s] + lane;
#pragma unroll
                    for (int i = 0; i < BOX_H; ++i) {
                        const float v = (float)d[s][i];
                        if (v != 0.0f && !((SMVS_BWD_ABLATE & 1) && v != 1234.5f)) unsafeAtomicAdd(q + i * W, v);
                    }
                }
                return;
            }
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                // only the rows some tap of the wave reaches (wave-uniform, known since the geometry: typically 3-4 of the 8) are
                // read, cleared and sent: the flush is VALU + LDS issue like the plane loop (round 6, profiles/r06_bwd_phases.txt)
                // (one branch per view on the row count, straight-line code behind it: a branch per row costs more than the rows save)
                const uint32_t ga = gbox_lds + (uint32_t)(s * GBOX_BYTES) + (uint32_t)lane * 8u;
                const BufRsrc& rg = rgs[s];
                const int so = c * HW * 4;
                auto rows = [&](auto nrc) __attribute__((always_inline)) {
                    constexpr int NR = decltype(nrc)::value;
                    double d[BOX_H];
                    gbox_take_rows<NR>(ga, d);
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        // untouched cells (still 0) send nothing: their lanes carry an out-of-range offset, which the range check drops
                        const float v = (float)d[i];
                        const bool send = v != 0.0f && !((SMVS_BWD_ABLATE & 1) && v != 1234.5f);
                        (void)llvm_raw_buffer_atomic_fadd_f32(v, rg.v, (int)(send ? gvo[s] : SMVS_OOB), so + i * W4, 0);
                    }
                };
                const int nr = box_rows[s];
                if (nr == 3) rows(std::integral_constant<int, 3>{});
                else if (nr == 4) rows(std::integral_constant<int, 4>{});
                else if (nr == 2) rows(std::integral_constant<int, 2>{});
                else if (nr == 5) rows(std::integral_constant<int, 5>{});
                else if (nr != 0) rows(std::integral_constant<int, BOX_H>{});
            }
        };
        // Round 5: the gradient planes run TWO channels ahead.  They are the kernel's only stream that always misses the caches (2.4 GB
        // read once), and with one channel of arithmetic (~3 us) between request and use the waves still waited for them (SQ_WAIT_ANY
        // 37 % of the wave cycles).  Two register sets take turns (the channel loop is unrolled by two so that no set is copied while
        // its loads are in flight); every iteration issues, in this order, the atomics of the previous channel, the boxes and the
        // reference value of the next one and LAST the gradient planes of the one after it -- so that the counted wait at the top of
        // an iteration (at most DCH operations in flight = those newest loads) covers everything this channel needs.
        // (1-2 source views; with more the second register set costs the kernel its occupancy: one channel ahead as before)
        constexpr int AHEAD = NSRC <= 2 ? 2 : 1;
        float g_far[AHEAD == 2 ? DCH : 1];
        if constexpr (AHEAD == 2) {
#pragma unroll
            for (int k = 0; k < DCH; ++k) g_far[k] = gp[((size_t)min(1, C - 1) * D + min(k, d1 - d0 - 1)) * HW];
        }
        float gref_prev = 0.0f;
        SMVS_BT(const unsigned long long bt_geo = bnow();)
        stage(0, 0);
        if constexpr (NFB == 3) stage(min(1, C - 1), 1);
        auto channel = [&](const int c, float (&gcur)[DCH]) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < DCH; ++k)
#pragma unroll
                for (int s = 0; s < NSRC; ++s) asm volatile("" : "+v"(tt[k][s]));       // see the other loop
            // this channel's boxes, r (requested one iteration ago) and g (two iterations ago) have arrived once only the newest DCH
            // operations -- the gradient planes of channel c + 1 -- are still in flight
            // (three box buffers: the boxes of channel c + 1 -- 2 NSRC instructions, issued in front of those planes -- may be in flight too)
            SMVS_BT(const unsigned long long bt0 = bnow();)
            if constexpr (NFB == 3) {
                if (c == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NSRC) : "memory");     // (the prologue issued the boxes of channel 1 last)
                else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NSRC + DCH) : "memory");
            } else if (c == 0 || AHEAD == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the prologue issued the boxes last)
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(DCH) : "memory");
            SMVS_BT(const unsigned long long bt1 = bnow(); bt_wait += bt1 - bt0;)
            const float r = r_next;
            float gq[DCH];
#pragma unroll
            for (int k = 0; k < DCH; ++k) gq[k] = gcur[k];
            if (c > 0) {                                    // the previous channel's sums leave while this one is worked on
                flush_boxes(c - 1);
                if (!((SMVS_BWD_ABLATE & 4) && gref_prev != 1234.5f))
                    (void)llvm_raw_buffer_atomic_fadd_f32(gref_prev, rgref.v, (int)(active ? pix4 : SMVS_OOB), (c - 1) * HW * 4, 0);
            }
            SMVS_BT(const unsigned long long bt2 = bnow(); bt_flush += bt2 - bt1;)
            {
                const int cn = min(c + 1, C - 1), cf = min(c + AHEAD, C - 1);
                r_next = llvm_raw_buffer_load_f32(rref.v, (int)pix4, cn * HW * 4, 0);     // (ahead of the boxes: it must have arrived by the next channel)
                if constexpr (NFB == 3) stage(cf, (c + 2) % 3);
                else stage(cn, (c + 1) & 1);
                load_planes(cf, gcur);
            }
            SMVS_BT(const unsigned long long bt3 = bnow(); bt_issue += bt3 - bt2;)
            const int choff = c * HW * 4;
            const uint32_t fpar = wave_lds + (uint32_t)((NFB == 3 ? c % 3 : (c & 1)) * NSRC * FBOX_BYTES);
            float gref = 0.0f;
            if (any_hole == 0) {
                // Interior of the image (every tap of the chunk is a full tap, every lane and plane is live): straight-line code,
                // the next plane's taps are requested before this plane's are waited for.  (Round 6: a TWO-pass form -- sample all
                // planes, then 8 independent planes of mean / contributions / adds with no wait in between -- measured SLOWER, 178 k
                // against 171 k clocks per wave in this loop: its time is not dependent-issue latency, profiles/r06_bwd_phases.txt.)
                f32x2 nq[2][NSRC], sq[2][NSRC];
#pragma unroll
                for (int s = 0; s < NSRC; ++s) fbox_read(fpar + (uint32_t)(s * FBOX_BYTES) + tt[0][s], nq[0][s], sq[0][s]);
#pragma unroll
                for (int k = 0; k < DCH; ++k) {
                    if (k + 1 < DCH) {
#pragma unroll
                        for (int s = 0; s < NSRC; ++s) fbox_read(fpar + (uint32_t)(s * FBOX_BYTES) + tt[k + 1][s], nq[(k + 1) & 1][s], sq[(k + 1) & 1][s]);
                    }
                    const float g = gq[k] * two_over_v;
                    float wv[NSRC];
                    f32x2 wn[NSRC], ws[NSRC];
                    float sum = r;
#pragma unroll
                    for (int s = 0; s < NSRC; ++s) {
                        // in flight behind this plane's reads: the previous plane's 4 adds per view, the next plane's 2 reads per view
                        if (k == 0) fbox_wait_n<2 * NSRC>(nq[0][s], sq[0][s]);
                        else if (k + 1 < DCH) fbox_wait_n<6 * NSRC>(nq[k & 1][s], sq[k & 1][s]);
                        else fbox_wait_n<4 * NSRC>(nq[k & 1][s], sq[k & 1][s]);
                        tap_weights(k, s, wn[s], ws[s]);
                        const f32x2 acc = __builtin_elementwise_fma(sq[k & 1][s], ws[s], nq[k & 1][s] * wn[s]);
                        float t = acc.x + acc.y;
                        asm volatile("" : "+v"(t));         // (one v_add_f32: left to itself the compiler pairs the views' halves up for a v_pk_add_f32 behind three register moves)
                        wv[s] = t;
                        sum = sum + t;
                    }
                    const float m = SMVS_BWD_MEAN_MUL ? sum * rV : div_by_views(sum, fV, rV);      // (the gradient's tolerance, not the forward's bits: one multiply)
                    gref = fmaf(g, r - m, gref);
#pragma unroll
                    for (int s = 0; s < NSRC; ++s) {
                        const float gw = g * (wv[s] - m);
                        const f32x2 cn = wn[s] * gw, cs = ws[s] * gw;
                        if (!(SMVS_BWD_ABLATE & 2)) gbox_add4(KEEPW ? tga[k][s] : gbox_lds + (uint32_t)(s * GBOX_BYTES) + 2u * tt[k][s], cn.x, cn.y, cs.x, cs.y);
                    }
                }
            } else
#pragma unroll
            for (int k = 0; k < DCH; ++k) {
                const float g = (active && d0 + k < d1) ? gq[k] * two_over_v : 0.0f;
                float wv[NSRC];
                f32x2 wn[NSRC], ws[NSRC];                   // (nw, ne), (sw, se)
                f32x2 north[NSRC], south[NSRC];
                float sum = r;
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
                    asm volatile("" : "+v"(tt[k][s]));      // opaque per plane: addresses and masks derived from it are formed here, not a channel ahead
                    const uint32_t e = tt[k][s];
                    // holes (dropped / border taps) read cell 0 of the box and are overridden below
                    fbox_read(fpar + (uint32_t)(s * FBOX_BYTES) + ((e & TAP_DROPPED) ? 0u : e), north[s], south[s]);
                }
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
                    const uint32_t e = tt[k][s], bit = 1u << (k * NSRC + s);
                    fbox_wait(north[s], south[s]);
                    float a0 = north[s].x, a1 = north[s].y, a2 = south[s].x, a3 = south[s].y;
                    if (any_hole & bit) {                   // wave-uniform: some lane's tap is dropped or touches the border
                        if (e & TAP_DROPPED) { a0 = 0.0f; a1 = 0.0f; a2 = 0.0f; a3 = 0.0f; }
                        if (any_partial & bit) {
                            if ((e & TAP_PARTIAL) == TAP_PARTIAL) {
                                a0 = llvm_raw_buffer_load_f32(rs[s].v, (int)corner_off(e, 0), choff, 0);
                                a1 = llvm_raw_buffer_load_f32(rs[s].v, (int)corner_off(e, 1), choff, 0);
                                a2 = llvm_raw_buffer_load_f32(rs[s].v, (int)corner_off(e, 2), choff, 0);
                                a3 = llvm_raw_buffer_load_f32(rs[s].v, (int)corner_off(e, 3), choff, 0);
                            }
                        }
                    }
                    // the same four products tap_from_grid forms, two per v_pk_mul_f32; the warped value sums north and south
                    // in one v_pk_fma_f32 and the two halves last (another order than the forward's: within its rounding)
                    tap_weights(k, s, wn[s], ws[s]);
                    const f32x2 an = {a0, a1}, as = {a2, a3};
                    const f32x2 acc = __builtin_elementwise_fma(as, ws[s], an * wn[s]);
                    const float t = acc.x + acc.y;
                    wv[s] = t;
                    sum = sum + t;
                }
                const float m = div_by_views(sum, fV, rV);
                gref = fmaf(g, r - m, gref);
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
                    const uint32_t e = tt[k][s], bit = 1u << (k * NSRC + s);
                    const float gw = g * (wv[s] - m);
                    const f32x2 cn = wn[s] * gw, cs = ws[s] * gw;
                    const float c0 = cn.x, c1 = cn.y, c2 = cs.x, c3 = cs.y;
                    const uint32_t ga = gbox_lds + (uint32_t)(s * GBOX_BYTES) + 2u * e;
                    if (any_hole & bit) {
                        if ((e & TAP_DROPPED) == 0) { if (!(SMVS_BWD_ABLATE & 2)) gbox_add4(ga, c0, c1, c2, c3); }
                        else if ((e & TAP_PARTIAL) == TAP_PARTIAL) scatter_partial(e, s, c, c0, c1, c2, c3);
                    } else if (!(SMVS_BWD_ABLATE & 2)) {
                        gbox_add4(ga, c0, c1, c2, c3);
                    }
                }
            }
            gref_prev = gref;
            SMVS_BT(bt_planes += bnow() - bt3;)
        };
        if constexpr (AHEAD == 2) {
            for (int c = 0; c < C; c += 2) {
                channel(c, g_next);
                if (c + 1 < C) channel(c + 1, g_far);
            }
        } else {
            for (int c = 0; c < C; ++c) channel(c, g_next);
        }
        flush_boxes(C - 1);
        if (!((SMVS_BWD_ABLATE & 4) && gref_prev != 1234.5f))
            (void)llvm_raw_buffer_atomic_fadd_f32(gref_prev, rgref.v, (int)(active ? pix4 : SMVS_OOB), (C - 1) * HW * 4, 0);
#ifdef SMVS_BWD_TIMING
        if (lane == 0) {
            atomicAdd(&smvs_bwd_timing[0], bt_geo - bt_start); atomicAdd(&smvs_bwd_timing[1], bt_wait); atomicAdd(&smvs_bwd_timing[2], bt_flush);
            atomicAdd(&smvs_bwd_timing[3], bt_issue); atomicAdd(&smvs_bwd_timing[4], bt_planes); atomicAdd(&smvs_bwd_timing[5], bnow() - bt_start);
            atomicAdd(&smvs_bwd_timing[7], 1ull);
            if (any_hole == 0) {
                // [6]: low 32 bits = plane-view pairs whose rows are single runs; bits 32.. = waves where ALL are; (reported by tools/wave_timing_bwd.py)
                const bool every = all_take == (DCH * NSRC >= 32 ? 0xffffffffu : (1u << (DCH * NSRC)) - 1u);
                atomicAdd(&smvs_bwd_timing[6], (unsigned long long)__builtin_popcount(all_take) + (every ? (1ull << 32) : 0ull));
            }
        }
#endif
        return;
    }

    // ================================ register scheme ==========================================================
    for (int c = 0; c < C; ++c) {
        // Re-materialise the per-tap words every channel: otherwise every lane mask derived from them (full / partial /
        // take / give, ~100 of them) is hoisted out of the loop as an SGPR pair and spilled (measured: 640 SGPR spills,
        // 256 VGPRs, one wave per SIMD).
#pragma unroll
        for (int k = 0; k < DCH; ++k)
#pragma unroll
            for (int s = 0; s < NSRC; ++s) asm volatile("" : "+v"(tt[k][s]));
        asm volatile("" : "+v"(take), "+v"(give));
        const float r = r_next;
        float gq[DCH];
#pragma unroll
        for (int k = 0; k < DCH; ++k) gq[k] = g_next[k];
        {
            const int cn = min(c + 1, C - 1);
            r_next = refp[(size_t)cn * HW];
#pragma unroll
            for (int k = 0; k < DCH; ++k) g_next[k] = gp[((size_t)cn * D + min(k, d1 - d0 - 1)) * HW];
        }
        const int choff = c * HW * 4;
        float gref = 0.0f;
        uint32_t rkey[NSRC];                                // run of planes whose tap sits in the same cell: key + 4 sums
        float racc[NSRC][4];
        bool live1[NSRC];                                   // the run still owns its east pair on some plane
#pragma unroll
        for (int s = 0; s < NSRC; ++s) { rkey[s] = TAP_DROPPED; racc[s][0] = racc[s][1] = racc[s][2] = racc[s][3] = 0.0f; live1[s] = false; }
        auto flush = [&](int s) {
            if ((rkey[s] & TAP_DROPPED) == 0) {
                float* q = p.grad_src[s] + ((size_t)b * C + c) * HW + (rkey[s] >> 2);
                unsafeAtomicAdd(q, racc[s][0]);
                unsafeAtomicAdd(q + W, racc[s][2]);
                if (live1[s]) { unsafeAtomicAdd(q + 1, racc[s][1]); unsafeAtomicAdd(q + W + 1, racc[s][3]); }
            }
        };
#pragma unroll
        for (int k = 0; k < DCH; ++k) {
            const float g = (active && d0 + k < d1) ? gq[k] * two_over_v : 0.0f;
            float wv[NSRC], tw[NSRC][4];
            float sum = r;
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const uint32_t e = tt[k][s], bit = 1u << (k * NSRC + s);
                float a0, a1, a2, a3;
                if (any_partial & bit) {                    // wave-uniform: this tap touches the border somewhere in the wave
                    a0 = llvm_raw_buffer_load_f32(rs[s].v, (int)corner_off(e, 0), choff, 0);
                    a1 = llvm_raw_buffer_load_f32(rs[s].v, (int)corner_off(e, 1), choff, 0);
                    a2 = llvm_raw_buffer_load_f32(rs[s].v, (int)corner_off(e, 2), choff, 0);
                    a3 = llvm_raw_buffer_load_f32(rs[s].v, (int)corner_off(e, 3), choff, 0);
                } else {                                    // full or dropped: one base offset, the rest in scalar / immediate offsets
                    a0 = llvm_raw_buffer_load_f32(rs[s].v, (int)e, choff, 0);
                    a1 = llvm_raw_buffer_load_f32(rs[s].v, (int)e + 4, choff, 0);
                    a2 = llvm_raw_buffer_load_f32(rs[s].v, (int)e, choff + W4, 0);
                    a3 = llvm_raw_buffer_load_f32(rs[s].v, (int)e + 4, choff + W4, 0);
                }
                weights(k, s, tw[s]);
                float t = a0 * tw[s][0];
                t = fmaf(a1, tw[s][1], t);
                t = fmaf(a2, tw[s][2], t);
                t = fmaf(a3, tw[s][3], t);
                wv[s] = t;
                sum = sum + t;
            }
            const float m = div_by_views(sum, fV, rV);
            gref = fmaf(g, r - m, gref);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const uint32_t e = tt[k][s], bit = 1u << (k * NSRC + s);
                const float gw = g * (wv[s] - m);
                float c0 = gw * tw[s][0], c1 = gw * tw[s][1], c2 = gw * tw[s][2], c3 = gw * tw[s][3];
                // contributions funnel EAST (cells: east lane's NW/SW = my NE/SE)
                const bool ge = (give & bit) != 0;
                if (any_take & bit) {                       // wave-uniform: somewhere in the wave an east pair moves one lane on
                    const float p1 = dpp_from_prev_lane(c1), p3 = dpp_from_prev_lane(c3);
                    if (take & bit) { c0 += p1; c2 += p3; }
                }
                if (ge) { c1 = 0.0f; c3 = 0.0f; }
                if ((e & TAP_DROPPED) == 0) {               // full tap: extend the run or start a new one
                    if (e != rkey[s]) {
                        flush(s);
                        rkey[s] = e; racc[s][0] = racc[s][1] = racc[s][2] = racc[s][3] = 0.0f; live1[s] = false;
                    }
                    racc[s][0] += c0; racc[s][1] += c1; racc[s][2] += c2; racc[s][3] += c3;
                    live1[s] = live1[s] || !ge;
                } else if ((e & TAP_PARTIAL) == TAP_PARTIAL) {
                    scatter_partial(e, s, c, c0, c1, c2, c3);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NSRC; ++s) flush(s);
        if (active && !((SMVS_BWD_ABLATE & 4) && gref != 1234.5f)) unsafeAtomicAdd(grefp + (size_t)c * HW, gref);
    }
}

template <int GEO, int NSRC>
static hipError_t launch_bwd_n(CostVolBwdParams p, hipStream_t st)
{
    // planes per lane: the taps of a chunk live in registers (5 per tap)
    constexpr int DCH = NSRC <= SMVS_BWD_DCH8_SRC ? 8 : NSRC <= 4 ? 4 : 2;
    p.dch = DCH < p.D ? DCH : p.D;
    p.dct = (p.D + DCH - 1) / DCH;
    const long long nb = (long long)p.xt * p.yt * p.dct * p.B;
    if (nb >= (1ll << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((costvol_bwd_kernel<GEO, NSRC, DCH>), dim3((unsigned)nb), dim3(64, BWD_WAVES), 0, st, p);
    return hipGetLastError();
}

template <int GEO>
static hipError_t launch_bwd(const CostVolBwdParams& p, hipStream_t st)
{
    switch (p.V - 1) {
    case 1: return launch_bwd_n<GEO, 1>(p, st);
    case 2: return launch_bwd_n<GEO, 2>(p, st);
    case 3: return launch_bwd_n<GEO, 3>(p, st);
    case 4: return launch_bwd_n<GEO, 4>(p, st);
    case 5: return launch_bwd_n<GEO, 5>(p, st);
    case 6: return launch_bwd_n<GEO, 6>(p, st);
    case 7: return launch_bwd_n<GEO, 7>(p, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace smvs

#ifdef SMVS_BWD_TIMING
extern "C" SMVS_EXPORT int smvs_debug_timing_bwd(unsigned long long* out8, int reset)
{
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(smvs::smvs_bwd_timing), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(smvs::smvs_bwd_timing), z, 64) != hipSuccess) return 1; }
    return 0;
}
#endif

extern "C" SMVS_EXPORT int smvs_costvol_bwd(int geo_kind, const float* grad_var, const float* ref_fea,
                                            const float* const* src_fea, int n_src, const double* geo,
                                            const float* depth, int depth_is_4d, float* grad_ref,
                                            float* const* grad_src, int B, int C, int D, int H, int W, void* stream)
{
    using namespace smvs;
    if (!grad_var || !ref_fea || !src_fea || !geo || !depth || !grad_ref || !grad_src) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (geo_kind != 0 && geo_kind != 1) return fail(SMVS_ERR_ARG, "geo_kind must be 0 (rpc) or 1 (homography)");
    if (n_src < 1 || n_src > MAX_SRC) return fail(SMVS_ERR_ARG, "n_src must be in [1,7], got %d", n_src);
    if (B < 1 || C < 1 || D < 1 || H < 1 || W < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if ((long long)C * H * W * 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "feature map larger than 2 GiB per batch item");
    if (H > 32766 || W > 32766) return fail(SMVS_ERR_ARG, "plane larger than 32766 in one dimension (border-tap encoding)");
    CostVolBwdParams p{};
    p.grad_var = grad_var; p.ref = ref_fea; p.grad_ref = grad_ref; p.geo = geo; p.depth = depth;
    for (int s = 0; s < n_src; ++s) {
        if (!src_fea[s] || !grad_src[s]) return fail(SMVS_ERR_ARG, "null source pointer %d", s);
        p.src[s] = src_fea[s]; p.grad_src[s] = grad_src[s];
    }
    p.B = B; p.V = n_src + 1; p.C = C; p.D = D; p.H = H; p.W = W; p.depth_is_4d = (depth_is_4d & ~SMVS_CALL_ARITH_MASK) != 0;
    p.xt = (W + TILE_X - 1) / TILE_X; p.yt = (H + TILE_Y - 1) / TILE_Y;
    hipError_t e = geo_kind == 0 ? launch_bwd<0>(p, (hipStream_t)stream) : launch_bwd<1>(p, (hipStream_t)stream);
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "costvol_bwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}
