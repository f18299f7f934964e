// costvol_kernels.h -- kernels and launchers of the fused plane-sweep warp + variance build (forward).
// Included by costvol.hip (reference-rounding instance, AR = 0) and costvol_fused.hip (contract-tolerance instance, AR = 1):
// two translation units so that the two families of instances compile in parallel.
#pragma once
#include <limits.h>
#include <stdlib.h>

#include <type_traits>

#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

constexpr int MAX_SRC = 7;          // V <= 8 views
constexpr int TILE_X = 64, TILE_Y = 4;

struct CostVolParams {
    const float* ref;               // (B,C,H,W)
    const float* src[MAX_SRC];      // (B,C,H,W) each
    const double* geo;              // rpc: (B,V,170); homography: (B,V-1,16) composed src @ inv(ref)
    const float* depth;             // (B,D,H,W) or (B,D)
    float* out;                     // (B,C,D_out,H,W)
    int B, V, C, D, H, W;
    int d_begin, d_end;             // planes built by this launch
    int D_out, d_out_off;           // plane d lands at index d - d_begin + d_out_off of `out`
    int depth_is_4d;                // HEIGHT_PLANES | HEIGHT_TENSOR | HEIGHT_GENERATED
    HeightGen hg;                   // HEIGHT_GENERATED: hypotheses computed per pixel from the previous stage's map
    int xt, yt, dct, dch;           // tiles in x, y; plane chunks; planes per chunk
    int chunk_major;                // staged kernel: order of the workgroups inside a band of rows, see launch_order()
    int group_rows;                 // ... and tile rows per group of the chunk-major order (<= 1: row by row)
    float rV, r_half_wm1, r_half_hm1;   // RN(1/V), RN(1/((W-1)/2)), RN(1/((H-1)/2)) in float32, divided once on the host
    float kw;                       // fused arithmetic (AR = 1): factor on the tap weights and the ref feature, see smvs_device.h
    const double* pc;               // rpc only, may be null: per-plane folded source cubics (smvs_rpc_plane_coef; layout: smvs_device.h,
                                    // "plane-constant heights")
};

// AR, the arithmetic of the variance: 0 = the reference's float32 rounding sequence, 1 = fused (contract tolerance)
enum { AR_EXACT = 0, AR_FUSED = 1 };

template <int GEO, int NSRC, int CT, int AR>
__global__ __launch_bounds__(TILE_X * TILE_Y)
void costvol_fwd_kernel(const CostVolParams p)
{
    // ---- which tile am I (XCD-aware order: x tile fastest, then plane chunk, then row band) ------
    uint32_t L = xcd_remap(blockIdx.x, gridDim.x);
    const int xtile = L % p.xt; L /= p.xt;
    const int dchunk = L % p.dct; L /= p.dct;
    const int ytile = L % p.yt;
    const int b = L / p.yt;

    const int x = xtile * TILE_X + threadIdx.x;
    const int y = ytile * TILE_Y + threadIdx.y;
    if (x >= p.W || y >= p.H) return;

    const int H = p.H, W = p.W, C = (CT > 0) ? CT : p.C;
    const int HW = H * W;
    const int d0 = p.d_begin + dchunk * p.dch;
    const int d1 = min(d0 + p.dch, p.d_end);
    const int pix = y * W + x;

    BufRsrc rs[NSRC];
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
        rs[s] = make_rsrc(p.src[s] + (size_t)b * C * HW, (uint32_t)C * (uint32_t)HW * 4u);

    // ref feature of this pixel: plane-invariant, kept in registers when C is a compile-time size
    const float* refp = p.ref + (size_t)b * C * HW + pix;
    float refv[CT > 0 ? CT : 1];
    if (CT > 0) {
#pragma unroll
        for (int c = 0; c < CT; ++c) refv[c] = refp[(size_t)c * HW];
    }

    const float fV = (float)p.V;
    const float rV = __fdiv_rn(1.0f, fV);
    const float half_wm1 = (float)((W - 1) * 0.5);
    const float half_hm1 = (float)((H - 1) * 0.5);

    // geometry constants
    const cgeo_t geo_b = as_cgeo((GEO == 0) ? p.geo + (size_t)b * p.V * RPC_LEN
                                            : p.geo + (size_t)b * (p.V - 1) * 16);
    RpcInv ref_n;
    RpcInv src_n[NSRC];
    if (GEO == 0) {
        ref_n = rpc_inv_image(geo_b);
#pragma unroll
        for (int s = 0; s < NSRC; ++s) src_n[s] = rpc_inv_ground(geo_b + (size_t)(s + 1) * RPC_LEN);
    }
    const double fx = (double)x, fy = (double)y;

    float* outp = p.out + (size_t)b * C * p.D_out * HW + pix;

    HeightPix hpx;
    if (p.depth_is_4d == HEIGHT_GENERATED) hg_prepare(p.hg, b, y, x, hpx);
    for (int d = d0; d < d1; ++d) {
        const float hf = p.depth_is_4d == HEIGHT_GENERATED ? hg_height(p.hg, hpx, d)
                         : p.depth_is_4d ? p.depth[((size_t)b * p.D + d) * HW + pix] : p.depth[(size_t)b * p.D + d];
        const double h = (double)hf;

        // Launder the (wave-uniform) coefficient pointer once per plane: without this the compiler
        // hoists all 80*V loop-invariant scalar loads out of the plane loop and spills ~570 SGPRs
        // into VGPR lanes.  Re-issuing the s_loads per plane (scalar-cache hits) keeps the
        // coefficients in SGPRs only while they are used.
        const cgeo_t geo_d = launder(geo_b);

        Tap tap[NSRC];
        if (GEO == 0) {
            double lat, lon;
            rpc_photo2obj(geo_d, ref_n, fx, fy, h, lat, lon);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                double samp, line;
                rpc_obj2photo(geo_d + (size_t)(s + 1) * RPC_LEN, src_n[s], lat, lon, h, samp, line);
                tap[s] = tap_from_pixel((float)samp, (float)line, H, W, half_wm1, half_hm1);
            }
        } else {
            // homo_warping, warping.py:28-38: rot.(x,y,1)*depth + trans, divide, normalise in float64
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const cgeo_t P = geo_d + s * 16;
                const double rx = fma(P[1], fy, P[0] * fx) + P[2];
                const double ry = fma(P[5], fy, P[4] * fx) + P[6];
                const double rz = fma(P[9], fy, P[8] * fx) + P[10];
                const double X = fma(rx, h, P[3]);
                const double Y = fma(ry, h, P[7]);
                const double Z = fma(rz, h, P[11]);
                const float gx = (float)((X / Z) / ((W - 1) * 0.5) - 1.0);
                const float gy = (float)((Y / Z) / ((H - 1) * 0.5) - 1.0);
                tap[s] = tap_from_grid(gx, gy, H, W);
            }
        }

        float* od = outp + (size_t)(d - p.d_begin + p.d_out_off) * HW;
        const size_t ostride = (size_t)p.D_out * HW;
        if constexpr (AR == AR_FUSED) {
#pragma unroll 2
            for (int c = 0; c < C; ++c) {
                const float rk = refp[(size_t)c * HW] * p.kw;
                float df[NSRC];
#pragma unroll
                for (int s = 0; s < NSRC; ++s) df[s] = fused_tap_diff(rs[s], tap[s], c * HW * 4, p.kw, rk);
                od[(size_t)c * ostride] = fused_variance<NSRC>(df, rV);
            }
        } else if (CT > 0) {
            // two channels per step on float2 (packed f32 math); same rounding per element
#pragma unroll 2
            for (int c = 0; c < CT; c += 2) {
                const f32x2 r = {refv[c], refv[c + 1]};
                f32x2 sum = r;
                f32x2 sq = r * r;
                const int choff = c * HW * 4;
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
                    const f32x2 wv = tap_fetch2(rs[s], tap[s], choff, HW * 4);
                    sum = sum + wv;
                    sq = sq + wv * wv;
                }
                const f32x2 m = div_by_views2(sum, fV, rV);
                const f32x2 q = div_by_views2(sq, fV, rV);
                const f32x2 var = q - m * m;
                __builtin_nontemporal_store(var.x, od + (size_t)c * ostride);
                __builtin_nontemporal_store(var.y, od + (size_t)(c + 1) * ostride);
            }
        } else {
#pragma unroll 2
            for (int c = 0; c < C; ++c) {
                const float r = refp[(size_t)c * HW];
                float sum = r;
                float sq = r * r;
                const int choff = c * HW * 4;
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
                    const float wv = tap_fetch(rs[s], tap[s], choff);
                    sum = sum + wv;
                    sq = sq + wv * wv;
                }
                const float m = div_by_views(sum, fV, rV);
                const float q = div_by_views(sq, fV, rV);
                od[(size_t)c * ostride] = q - m * m;
            }
        }
    }
}

// ---- geometry shared by the staged (LDS) kernel ----------------------------------------------------
constexpr int WV_TX = 32, WV_TY = 2;          // ref pixels per wave
#ifndef SMVS_WG_WAVES
#define SMVS_WG_WAVES 2
#endif
constexpr int WV_WAVES = SMVS_WG_WAVES;       // waves per workgroup, stacked in y: 32 x 8 pixels

__device__ __forceinline__ int wave_min(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m, 64));
    return v;
}

__device__ __forceinline__ int wave_max(int v)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m, 64));
    return v;
}

// =====================================================================================================
// Staged kernel (default): source tiles in wave-private LDS, filled by LDS-DMA.
//
// The direct kernel above is bound by the texture-address path: 4 unaligned gathers per channel and
// source = 261 vector-memory instructions per 64 voxels, TA 75 % busy at 8 cycles each
// (profiles/r01_v1_direct_gather_summary.txt).  Here every WAVE owns a 32 x 2 patch of ref pixels and
// works alone -- no workgroup barrier anywhere, waves slide past each other:
//   A. taps of DM_DP consecutive planes for its 64 pixels (float64 chain): LDS address + 4 weights;
//   B. bounding box of those taps in each source image (wave min/max);
//   C. per channel PAIR ("step"): the box -- DM_R rows x DM_BW columns x 2 channels, PLANAR: [row][channel][column] --
//      goes into LDS by 16-byte LDS-DMA chunks (buffer_load_dwordx4 ... lds: lane -> 4 columns of one channel row;
//      4 instructions per step for two sources instead of ~10 of the round-2 dword map; chunks outside the image
//      deposit zeros = zero padding);
//   D. every tap corner of the pair is one ds_read2_b32 (the corner's dword of both channels), feeding packed f32
//      math in a hand-fixed, hazard-free order (smvs_device.h: pk_bilinear2 / pk_finish2); the variance pair leaves
//      through two non-temporal buffer stores.
// The kernel runs AT THE BOARD POWER LIMIT (1400 W, shader clock throttled to ~1.8 GHz: profiles/r03_power.txt), so
// its time follows the ENERGY of a launch, not the overlap of its phases: fewer cycles at the same work only lower
// the clock.  What counts is instructions and bytes moved per voxel.
// Global loads per voxel-channel-source drop from 4 gathers to ~0.6 coalesced DMA lanes; rows shared by
// the patch's two image rows and by the DM_DP planes are fetched once.  Latency is taken off the
// wave's critical path three ways:
//   * two LDS buffers: the DMA for step st+1 is issued before step st is computed and only waited for
//     (counted s_waitcnt vmcnt(N), N = the operations issued after it) when step st+1 begins, so
//     neither load latency nor store acknowledgements stall the arithmetic;
//   * the LDS reads of plane pl+1 are in flight while plane pl is computed (counted lgkmcnt);
//   * ref features run two steps ahead in registers; results leave through buffer stores whose
//     channel base lives in the descriptor (no 64-bit address arithmetic per store).
// A box that does not fit (exotic geometry) makes that wave take the direct gathers for the plane
// group, so results never depend on which path ran.  Bits are identical to the direct kernel and to
// the oracle.
// =====================================================================================================
#ifndef SMVS_BOX_W
#define SMVS_BOX_W 44                 // staged box width: 11 chunks of 4 columns (a 40-column box from any 4-aligned origin)
#endif
#ifndef SMVS_ONE_BASE
#define SMVS_ONE_BASE (-1)            // A/B switch of profiling builds
#endif
#ifndef SMVS_BOX_W8
#define SMVS_BOX_W8 52                // ... of a box shared by 8 planes (3+ sources)
#endif
#ifndef SMVS_WAVES_PER_SIMD
#define SMVS_WAVES_PER_SIMD 3
#endif
constexpr int DM_BW = SMVS_BOX_W;      // staged box width (columns)
#ifndef SMVS_BOX_R
#define SMVS_BOX_R 5
#endif
constexpr int DM_R = SMVS_BOX_R;        // staged box rows: 2 pixel rows + south tap + parallax/rotation slack
constexpr int DM_NBUF = 2;
#ifndef SMVS_ABLATE
#define SMVS_ABLATE 0                 // profiling builds only (tools/ab_build.sh x -DSMVS_ABLATE=n): 1 stores dropped, 2 no staging DMA, 4 no float64 chain, 8 no LDS tap reads, 32 no packed arithmetic, 64 staging DMA issued with every lane out of range (no memory traffic), 128 staging DMA from the first 64 KB of a channel (cache hits), 256 a step does not wait for its DMA to land, 512 no workgroup barrier per step (shared boxes) -- results are WRONG
#endif
#ifndef SMVS_O2P_PLANES
#define SMVS_O2P_PLANES 4          // planes per evaluation pass of a source view's cubics
#endif
#ifndef SMVS_PC_PLANES
#define SMVS_PC_PLANES 4           // ... and of the bivariate (plane-constant) chain: 6 folded coefficients per plane and cubic in SGPRs
#endif
#ifndef SMVS_WPS_DP8
#define SMVS_WPS_DP8 2                // waves per SIMD the 8-plane instance is compiled for
#endif
#ifndef SMVS_WPS_DP8_FUSED
#define SMVS_WPS_DP8_FUSED 2          // the same for the fused-arithmetic instance
#endif
#ifndef SMVS_WPS_NSRC34_DP2
#define SMVS_WPS_NSRC34_DP2 2         // 3-4 sources, 1-2 planes per wave (166 VGPRs at 4 sources): waves per SIMD compiled for
#endif
#ifndef SMVS_WPS_NSRC34_DP4
#define SMVS_WPS_NSRC34_DP4 2         // 3-4 sources, 4 planes per wave: waves per SIMD compiled for
#endif
#ifndef SMVS_NSRC34_DP
#define SMVS_NSRC34_DP 4              // planes per wave of a 3-4 source sweep (A/B switch of profiling builds)
#endif
#ifndef SMVS_DP8_MINC
#define SMVS_DP8_MINC 32              // fewest channels for which a sweep that divides into eights takes 8 planes per wave
#endif
#ifndef SMVS_DP8_HOMO
#define SMVS_DP8_HOMO 1               // 8 planes per wave for the homography variant too (round 3: 0.557 vs 0.583 ms at 768x384x64, C=32)
#endif
#ifndef SMVS_DP8
#define SMVS_DP8 1                    // 8 planes per wave for rpc C=32 sweeps (A/B switch of profiling builds)
#endif
#ifndef SMVS_STORE_AUX
#define SMVS_STORE_AUX 2              // nt: the variance volume streams out once, keep it from evicting feature rows in L2
#endif
constexpr int STORE_AUX = SMVS_STORE_AUX;

#ifdef SMVS_TIMING
// profiling builds only (tools/ab_build.sh x -DSMVS_TIMING): per-wave phase stamps in shader clocks, read back through
// smvs_debug_timing().  [0] geometry phase, [1] box + setup, [2] channel-pair loop, [3] of which spent in the vmcnt waits,
// [4] of which in the lgkmcnt waits of the last plane, [5] DMA issue
__device__ unsigned long long smvs_timing[12];      // [8] heights + plane check (prefetch), [9] reciprocal scales + ref view, [10] source views + taps
__device__ __forceinline__ unsigned long long now() { unsigned long long t = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); return t; }
#define SMVS_T(...) __VA_ARGS__
#else
#define SMVS_T(...)
#endif

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// wait until at most n (0..8, wave-uniform) vector-memory operations are outstanding
__device__ __forceinline__ void wait_vmcnt_upto8(int n)
{
    switch (n) {
    case 0: wait_vmcnt<0>(); break;  case 1: wait_vmcnt<1>(); break;  case 2: wait_vmcnt<2>(); break;
    case 3: wait_vmcnt<3>(); break;  case 4: wait_vmcnt<4>(); break;  case 5: wait_vmcnt<5>(); break;
    case 6: wait_vmcnt<6>(); break;  case 7: wait_vmcnt<7>(); break;  default: wait_vmcnt<8>(); break;
    }
}

// base: LDS byte address of the north-west corner in staging buffer 0.  The weights are kept as two
// register pairs so that v_pk_* can broadcast either half through op_sel (no v_mov to build {w,w}).
struct TapD { uint32_t base[DM_NBUF]; f32x2 wn, ws; };       // base[parity] = LDS address of the NW corner; wn = {nw, ne}, ws = {sw, se}

// NWY > 0 selects the WORKGROUP-SHARED form (round 5): the workgroup's NWY x NWP waves -- NWY stacked in y (2 pixel rows each),
// NWP along the plane axis (DP planes each) -- stage ONE box per source, of 2 NWY + 3 rows, and share it: the DMA instructions
// of a step are dealt out over the waves and one s_barrier per step orders "my part has landed" / "everybody is done with the
// other buffer".  Fewer staged rows per voxel (7 rows per 4 pixel rows instead of 2 x 5; one box for NWP x DP planes), and a
// wave's register and LDS footprint is that of DP planes while the box and the ref view's per-pixel work are amortised
// over NWP x DP.  NWY = 0: every wave stages its own box and never meets a barrier (WV_WAVES waves stacked in y).
template <int GEO, int NSRC, int CT, int DP, int AR, int NWY = 0, int NWP = 1, int WPS = 0>
__global__ __launch_bounds__(NWY > 0 ? 64 * NWY * NWP : 64 * WV_WAVES,
                             NWY > 0 ? WPS : (NSRC <= 2 && DP <= 4 ? SMVS_WAVES_PER_SIMD : NSRC <= 2 ? (AR == AR_FUSED ? SMVS_WPS_DP8_FUSED : SMVS_WPS_DP8) : (NSRC <= 4 && DP <= 2 ? SMVS_WPS_NSRC34_DP2 : NSRC <= 4 ? SMVS_WPS_NSRC34_DP4 : 2)))
void costvol_dma_kernel(const CostVolParams p)
{
    // Staging layout of one source box and channel pair: [row][channel of the pair][column] dwords, row pitch 2*BW; filled
    // by 16-byte LDS-DMA chunks (4 columns of one channel row), chunk k of the box at byte 16 k.
    constexpr bool SHARED = NWY > 0;
    constexpr int NW = SHARED ? NWY * NWP : WV_WAVES;        // waves per workgroup
    constexpr int NY = SHARED ? NWY : WV_WAVES;              // of which stacked in y
    // box width: 44 columns hold 4-5 planes of a 5-view 1536-wide tile (1.7 columns of parallax per plane at the steepest view);
    // a box shared by 8 planes takes 52
    constexpr int BW = SHARED && DP * NWP >= 8 && NSRC > 2 ? SMVS_BOX_W8 : DM_BW, R = SHARED ? 2 * NWY + 3 : DM_R, C4 = BW / 4;
    constexpr int SLOTS = R * 2 * C4;                        // 16-byte chunks per source box and channel pair
    constexpr int NI = (SLOTS + 63) / 64;                    // DMA instructions per source box and channel pair
    constexpr int SRC_DW = NI * 256;                         // dwords per source box, padded to whole DMA instructions
    constexpr int BUF_DW = NSRC * SRC_DW;
    constexpr int ZPAD_DW = 3 * BW + 4;                      // always-zero dwords a dropped tap reads (offsets 0 .. 3*BW+1)
    // ONE_BASE: a tap keeps one LDS address (buffer 0) and the odd steps add the buffer pitch -- one VALU add per tap and odd
    // step for NSRC * DP fewer registers; every buffer is then followed by its own zero cells, so that a dropped tap's address
    // moves with the others.  Taken where registers decide the occupancy (the shared form, 3+ sources).
    constexpr bool ONE_BASE = SMVS_ONE_BASE >= 0 ? (SHARED && SMVS_ONE_BASE) : (SHARED && WPS >= 3 && NSRC > 2);
    constexpr int BUFP_DW = ONE_BASE ? BUF_DW + ZPAD_DW : BUF_DW;       // buffer pitch
    constexpr int TILE_DW = ONE_BASE ? DM_NBUF * BUFP_DW : DM_NBUF * BUF_DW + ZPAD_DW;
    constexpr int NSTEP = CT / 2;
    static_assert(BW % 4 == 0 && CT % 2 == 0 && 2 * DP + 2 <= 63 && DP * NSRC <= 32, "chunks / steps / vmcnt bookkeeping / tap mask");
    __shared__ __attribute__((aligned(16))) uint32_t tile_all[SHARED ? 1 : WV_WAVES][(TILE_DW + 3) & ~3];
    __shared__ __attribute__((aligned(16))) int xch[SHARED ? NW : 1][NSRC][4];   // shared form: every wave's tap extents
#ifdef SMVS_LDS_PAD
    __shared__ float lds_pad[SMVS_LDS_PAD / 4];            // profiling builds only: caps the workgroups per CU
    if (p.B < 0) lds_pad[threadIdx.x] = 0.0f;
#endif
    // one wave = one 32 x 2 pixel patch x ONE group of DP planes (p.dch == DP): no loop over groups, so
    // nothing of the geometry phase stays live across the channel-pair loop
    uint32_t L = xcd_remap(blockIdx.x, gridDim.x);
    int xtile, ytile, dchunk;
    if (p.chunk_major && p.group_rows > 1) {
        // plane chunk fastest, then GROUPS of group_rows tile rows walked column by column: the workgroups an XCD runs together
        // (32 CUs x 2-3) form a block of tiles that is about as tall as it is wide, so the rows and columns neighbouring boxes
        // share are requested while they are in that XCD's L2 (launch_order())
        dchunk = L % p.dct; L /= p.dct;
        const uint32_t tiles = (uint32_t)p.xt * (uint32_t)p.yt, t = L % tiles;
        L /= tiles;
        const uint32_t per = (uint32_t)p.group_rows * (uint32_t)p.xt, g = t / per, r = t - g * per;
        const uint32_t rows = min((uint32_t)p.group_rows, (uint32_t)p.yt - g * (uint32_t)p.group_rows);
        xtile = r / rows;
        ytile = g * p.group_rows + (r - xtile * rows);
    } else {
        if (p.chunk_major) { dchunk = L % p.dct; L /= p.dct; xtile = L % p.xt; L /= p.xt; }     // see launch_order()
        else               { xtile = L % p.xt; L /= p.xt; dchunk = L % p.dct; L /= p.dct; }
        ytile = L % p.yt;
        L /= p.yt;
    }
    const int b = L;

    const int H = p.H, W = p.W;
    const int HW = H * W;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wy = SHARED ? wave % NY : wave;               // position in the workgroup's stack of row pairs
    const int wp = SHARED ? wave / NY : 0;                  // ... and along the plane axis
    uint32_t* tile = tile_all[SHARED ? 0 : wave];
    const uint32_t tile_lds = __builtin_amdgcn_readfirstlane(lds_addr(tile));
    const int x = xtile * WV_TX + (lane & (WV_TX - 1));
    const int y = (ytile * NY + wy) * WV_TY + (lane >> 5);
    const bool active = (x < W) && (y < H);
    const int pix = min(y, H - 1) * W + min(x, W - 1);
    const int dg = p.d_begin + (dchunk * NWP + wp) * DP;
    const int np = min(DP, p.d_end - dg);                    // may be <= 0 in the shared form (a wave past the last plane: it only stages)

    if constexpr (!SHARED) {
        if ((ytile * WV_WAVES + wave) * WV_TY >= H) return;  // whole wave below the image (no barriers used)
    }

    // zero cells behind each buffer: a tap whose footprint misses the image reads these, so it
    // contributes 0 * weight exactly like four masked gathers (0, or NaN for a NaN coordinate)
    if constexpr (ONE_BASE) {
        for (int i = lane; i < ZPAD_DW; i += 64) tile[BUF_DW + i] = tile[BUFP_DW + BUF_DW + i] = 0u;     // (every wave writes the same zeros)
    } else {
        for (int i = lane; i < ZPAD_DW; i += 64) tile[DM_NBUF * BUF_DW + i] = 0u;
    }

    const float fV = (float)p.V;
    const float rV = p.rV;
    const float half_wm1 = (float)((W - 1) * 0.5);
    const float half_hm1 = (float)((H - 1) * 0.5);
    const float r_half_wm1 = p.r_half_wm1, r_half_hm1 = p.r_half_hm1;

    const cgeo_t geo_b = as_cgeo((GEO == 0) ? p.geo + (size_t)b * p.V * RPC_LEN
                                            : p.geo + (size_t)b * (p.V - 1) * 16);
    const double fx = (double)min(x, W - 1), fy = (double)min(y, H - 1);
    const uint32_t pix4 = (uint32_t)pix * 4u;

    SMVS_T(const unsigned long long t_start = now(); unsigned long long t_vm = 0, t_dma = 0, t_st = 0;)
    constexpr int PC_LPR = (DP * PC_PER_CUBIC * 8 + 63) / 64 + (DP * PC_PER_CUBIC * 8 % 64 ? 1 : 0);      // 64-byte lines per coefficient run of DP planes
    const bool pc_maybe = GEO == 0 && p.pc != nullptr && p.depth_is_4d != HEIGHT_GENERATED && !(SMVS_ABLATE & 4) && np > 0;     // (np <= 0: shared form, a wave past the last plane only stages)
    const int dg0 = min(dg, p.d_end - 1);
    // heights of the group's planes (tail planes shadow the last one; they are never stored)
    float hf[DP];
    if (p.depth_is_4d == HEIGHT_GENERATED) {
        HeightPix hpx;
        hg_prepare(p.hg, b, min(y, H - 1), min(x, W - 1), hpx);
#pragma unroll
        for (int pl = 0; pl < DP; ++pl) hf[pl] = hg_height(p.hg, hpx, min(dg + pl, p.d_end - 1));
    } else {
#pragma unroll
        for (int pl = 0; pl < DP; ++pl) {
            const int d = min(dg + pl, p.d_end - 1);
            hf[pl] = p.depth_is_4d ? p.depth[((size_t)b * p.D + d) * HW + pix] : p.depth[(size_t)b * p.D + d];
        }
    }

    // Plane-coefficient path (smvs_device.h, "plane-constant heights"): the coefficient runs of the wave's planes -- one run
    // of DP planes per (source, cubic) -- and the line that holds the planes' heights are pulled into the scalar cache now,
    // while the height loads above are in flight.
    const cgeo_t pc_co = as_cgeo(p.pc) + pc_header_doubles((size_t)p.B * p.D, p.B);      // coefficient area, pc_offset()
    if constexpr (GEO == 0) {
        if (pc_maybe) {
            // (heights: 8 doubles from wherever in a line the group starts; scales: 24 doubles; then one wait per source view)
            scalar_touch4<2>(as_cgeo(p.pc) + (size_t)b * p.D + dg0, as_cgeo(p.pc) + pc_heights_doubles((size_t)p.B * p.D) + (size_t)b * PC_SCALES,
                             as_cgeo(p.pc) + pc_heights_doubles((size_t)p.B * p.D) + (size_t)b * PC_SCALES + 16, as_cgeo(p.pc) + (size_t)b * p.D + dg0);
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
                scalar_touch4<PC_LPR>(pc_co + pc_offset(b, s, 0, dg0, NSRC, p.D), pc_co + pc_offset(b, s, 1, dg0, NSRC, p.D),
                                      pc_co + pc_offset(b, s, 2, dg0, NSRC, p.D), pc_co + pc_offset(b, s, 3, dg0, NSRC, p.D));
        }
    }

    bool use_pc = false;
    SMVS_T(const unsigned long long t_chk = now();)
    // ---- A: taps of the group's planes -----------------------------------------------------------
    TapD tap[DP][NSRC];
    uint32_t txy[DP][NSRC];
    uint32_t okmask = 0;
    int lo_x[NSRC], hi_x[NSRC], lo_y[NSRC], hi_y[NSRC];
#pragma unroll
    for (int s = 0; s < NSRC; ++s) { lo_x[s] = lo_y[s] = INT_MAX; hi_x[s] = hi_y[s] = INT_MIN; }

    double lat[DP], lon[DP];
#pragma unroll
    for (int pl = 0; pl < DP; ++pl) lat[pl] = lon[pl] = 0.0;
    RpcInv ref_n, src_n[NSRC];
    ref_n.a = ref_n.b = ref_n.h = 0.0;
#pragma unroll
    for (int s = 0; s < NSRC; ++s) src_n[s].a = src_n[s].b = src_n[s].h = 0.0;
    if (GEO == 0 && !(SMVS_ABLATE & 4)) {
        // The 3 reciprocal scales of every view, in VGPRs (18 SGPRs would not survive the coefficient loads).
        if (pc_maybe) {
            // the fold kernel divided already (smvs_device.h, PC_SCALES): scalar loads out of the workspace, copied to VGPRs
            const cgeo_t sc = launder(as_cgeo(p.pc)) + pc_heights_doubles((size_t)p.B * p.D) + (size_t)b * PC_SCALES;
            ref_n.a = to_vgpr(sc[0]); ref_n.b = to_vgpr(sc[1]); ref_n.h = to_vgpr(sc[2]);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) { src_n[s].a = to_vgpr(sc[3 * s + 3]); src_n[s].b = to_vgpr(sc[3 * s + 4]); src_n[s].h = to_vgpr(sc[3 * s + 5]); }
        } else {
            // no workspace: every lane forms the nine reciprocals itself from scalar-loaded scales (recip_scale: 5 instructions each, no
            // dependence between them).  Until round 6 lane 3v+k did one IEEE division and the quotients travelled through LDS: a
            // global load -> ~35-instruction division -> LDS round trip chain of ~10 000 clocks at the start of every wave.
            const cgeo_t gr = launder(geo_b);
            ref_n.a = recip_scale(gr[I_SAMP_SCALE]); ref_n.b = recip_scale(gr[I_LINE_SCALE]); ref_n.h = recip_scale(gr[I_H_SCALE]);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const cgeo_t gs = launder(geo_b) + (size_t)(s + 1) * RPC_LEN;
                src_n[s].a = recip_scale(gs[I_LAT_SCALE]); src_n[s].b = recip_scale(gs[I_LON_SCALE]); src_n[s].h = recip_scale(gs[I_H_SCALE]);
            }
        }
    }
    SMVS_T(const unsigned long long t_ref = now();)
    // PQ planes per pass: every source coefficient is fetched into SGPRs once for all of them
    constexpr int PQ = DP < SMVS_O2P_PLANES ? DP : SMVS_O2P_PLANES;
    static_assert(DP % PQ == 0, "planes per pass");
    // Planes and source views for the heights hh.  PC: every plane's height is the one its coefficients were folded for (the bivariate
    // source cubics, smvs_device.h); else the trivariate chain.
    auto planes_and_sources = [&](auto pc_tag, const float (&hh)[DP]) __attribute__((always_inline)) {
        constexpr bool PC = decltype(pc_tag)::value;
        okmask = 0;
#pragma unroll
        for (int s = 0; s < NSRC; ++s) { lo_x[s] = lo_y[s] = INT_MAX; hi_x[s] = hi_y[s] = INT_MIN; }
        if (GEO == 0 && !(SMVS_ABLATE & 4)) {
            // ref view, image -> ground: plane-invariant part once per pixel, Horner in the height per plane.  (Formed inside, so that the
            // 24 registers of the pixel part die before the source passes: a wave that has to redo its planes forms them again.)
            P2OPix px;
            p2o_pixel(geo_b, ref_n, fx, fy, px);
            p2o_planes<DP>(launder(geo_b), ref_n, px, hh, lat, lon);
        }
#pragma unroll
        for (int pq = 0; pq < DP; pq += PQ) {
            const cgeo_t geo_d = launder(geo_b);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                float gxs[PQ], gys[PQ];
                if (SMVS_ABLATE & 4) {
#pragma unroll
                    for (int u = 0; u < PQ; ++u) {
                        gxs[u] = ((float)fx + 0.37f + 0.011f * hh[pq + u] * (float)(s + 1)) / half_wm1 - 1.0f;
                        gys[u] = ((float)fy + 0.21f) / half_hm1 - 1.0f;
                    }
                } else if (GEO == 0) {
                    double samp[PQ], line[PQ];
                    if constexpr (PC) {
                        // planes per pass of the bivariate chain: 4 where the instance has 256 registers (8 planes per wave, 2 waves per SIMD), 2 in the
                        // 168-register instances (9 monomial registers pairs per plane in flight: 4 planes spilled 20 registers to scratch)
                        constexpr int PQW = DP >= 8 ? SMVS_PC_PLANES : 2, PQC = PQ < PQW ? PQ : PQW;
#pragma unroll
                        for (int v = 0; v < PQ; v += PQC)
                            o2p_pc_xn<PQC>(geo_d + (size_t)(s + 1) * RPC_LEN, src_n[s], lat + pq + v, lon + pq + v,
                                           pc_co + pc_offset(b, s, 0, dg + pq + v, NSRC, p.D), (size_t)p.D * PC_PER_CUBIC, samp + v, line + v);
                    } else {
                        double hd[PQ];
#pragma unroll
                        for (int u = 0; u < PQ; ++u) hd[u] = (double)hh[pq + u];
                        o2p_xn<PQ>(geo_d + (size_t)(s + 1) * RPC_LEN, src_n[s], lat + pq, lon + pq, hd, samp, line);
                    }
#pragma unroll
                    for (int u = 0; u < PQ; ++u) {
                        gxs[u] = div_half_int((float)samp[u], half_wm1, r_half_wm1) - 1.0f;
                        gys[u] = div_half_int((float)line[u], half_hm1, r_half_hm1) - 1.0f;
                    }
                } else {
                    const cgeo_t P = geo_d + s * 16;
#pragma unroll
                    for (int u = 0; u < PQ; ++u) {
                        const double hd = (double)hh[pq + u];
                        const double rx = fma(P[1], fy, P[0] * fx) + P[2];
                        const double ry = fma(P[5], fy, P[4] * fx) + P[6];
                        const double rz = fma(P[9], fy, P[8] * fx) + P[10];
                        const double X = fma(rx, hd, P[3]), Y = fma(ry, hd, P[7]), Z = fma(rz, hd, P[11]);
                        gxs[u] = (float)((X / Z) / ((W - 1) * 0.5) - 1.0);
                        gys[u] = (float)((Y / Z) / ((H - 1) * 0.5) - 1.0);
                    }
                }
#pragma unroll
                for (int u = 0; u < PQ; ++u) {
                    const int pl = pq + u;
                    // same arithmetic as tap_from_grid (ATen unnormalise, floor, weights)
                    const float px_ = fmaf(gxs[u] + 1.0f, (float)W * 0.5f, -0.5f);
                    const float py_ = fmaf(gys[u] + 1.0f, (float)H * 0.5f, -0.5f);
                    const float xw = floorf(px_), yn = floorf(py_);
                    const float w = px_ - xw, e = 1.0f - w, n = py_ - yn, so = 1.0f - n;
                    // A footprint that misses the image reads the zero cells: 0 * weight = 0 for any finite
                    // weight, NaN for the NaN weights of a NaN / infinite coordinate -- what four masked
                    // gathers contribute.  v_cvt_i32_f32 saturates and maps NaN to 0, so the unsigned tests
                    // reject every far-away coordinate (a NaN one passes with NaN weights: NaN either way).
                    if constexpr (AR == AR_FUSED) {
                        // the variance's constant factor rides on the weights (smvs_device.h, "fused arithmetic")
                        const float sk = so * p.kw, nk = n * p.kw;
                        tap[pl][s].wn.x = sk * e; tap[pl][s].wn.y = sk * w;
                        tap[pl][s].ws.x = nk * e; tap[pl][s].ws.y = nk * w;
                    } else {
                        tap[pl][s].wn.x = so * e; tap[pl][s].wn.y = so * w;
                        tap[pl][s].ws.x = n * e;  tap[pl][s].ws.y = n * w;
                    }
                    const int ix0 = cvt_i32_sat(xw), iy0 = cvt_i32_sat(yn);
                    const bool ok = ((uint32_t)(ix0 + 1) <= (uint32_t)W) && ((uint32_t)(iy0 + 1) <= (uint32_t)H);
                    txy[pl][s] = (uint32_t)((iy0 + 1) * (2 * BW) + (ix0 + 1)); // dword index relative to image corner (-1,-1); used only if ok
                    if (ok) okmask |= 1u << (pl * NSRC + s);
                    if (ok && active && pl < np) {
                        lo_x[s] = min(lo_x[s], ix0); hi_x[s] = max(hi_x[s], ix0);
                        lo_y[s] = min(lo_y[s], iy0); hi_y[s] = max(hi_y[s], iy0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // Plane-constant heights?  SPECULATED: where folded coefficients are at hand the wave runs the bivariate chain on the heights
    // they were folded FOR (wave-uniform values out of the workspace, already in the scalar cache) and only afterwards compares
    // them with its own 64 x DP heights -- by then the (B,D,H,W) loads issued at the top have long landed, so their HBM latency
    // (~4 000 clocks under full write load when asked behind the ref pixel's work, 10 000 when asked at once:
    // profiles/r06_plane_coef_transport.txt) is off the wave's critical path.  A single differing lane (per-voxel hypotheses, a
    // jittered pixel, a NaN, a stale workspace) and the wave redoes its planes on the trivariate chain with its own heights: the
    // result never depends on trusting the caller.
    bool geometry_done = false;
    if constexpr (GEO == 0) {
        if (pc_maybe) {
            const cgeo_t hdr = launder(as_cgeo(p.pc)) + (size_t)b * p.D + dg0;
            float hs[DP];
#pragma unroll
            for (int pl = 0; pl < DP; ++pl) hs[pl] = (float)hdr[min(pl, p.d_end - 1 - dg0)];
            planes_and_sources(std::true_type(), hs);
            bool eq = true;
#pragma unroll
            for (int pl = 0; pl < DP; ++pl) eq = eq && (hf[pl] == hs[pl]);
            use_pc = __ballot(!eq) == 0ull;
            geometry_done = use_pc;
        }
    }
    if (!geometry_done) planes_and_sources(std::false_type(), hf);

    if constexpr (GEO == 0 && DP > 1) {
        // A group cut short by the end of the sweep: its tail planes carry the last plane's height, but a pass of the
        // bivariate chain read the records of the planes BEHIND the sweep for them -- give them the last plane's taps
        // (what the trivariate chain computes for them; they are never stored).  Wave-uniform, taken by tail groups only.
        if (use_pc && np < DP) {
#pragma unroll
            for (int pl = 1; pl < DP; ++pl)
                if (pl >= np) {
#pragma unroll
                    for (int s = 0; s < NSRC; ++s) {
                        tap[pl][s] = tap[pl - 1][s];
                        txy[pl][s] = txy[pl - 1][s];
                        const uint32_t bit = (okmask >> ((pl - 1) * NSRC + s)) & 1u;
                        okmask = (okmask & ~(1u << (pl * NSRC + s))) | (bit << (pl * NSRC + s));
                    }
                }
        }
    }

    SMVS_T(const unsigned long long t_geo = now();)
    // ---- B: the wave's bounding box per source -------------------------------------------------------
    // The box origin is moved left onto a chunk grid that the image edge it may touch falls on (column 0, or column W
    // when the last column a tap reads lies within a chunk of the right edge and W is not a multiple of 4): a 16-byte
    // chunk is then inside the image row or outside it as a whole, and outside chunks deposit zeros = zero padding.
    // For W % 4 == 0 every chunk is 16-byte aligned in memory as well.
    int bx0[NSRC], by0[NSRC], bw[NSRC], bh[NSRC];
    bool fits = true;
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        int a0 = lo_x[s], a1 = hi_x[s], b0 = lo_y[s], b1 = hi_y[s];
        wave_minmax4(a0, a1, b0, b1);
        if constexpr (SHARED) {
            if (lane == 0) { xch[wave][s][0] = a0; xch[wave][s][1] = a1; xch[wave][s][2] = b0; xch[wave][s][3] = b1; }
        }
        lo_x[s] = a0; hi_x[s] = a1; lo_y[s] = b0; hi_y[s] = b1;
    }
    if constexpr (SHARED) {
        // the workgroup's box = the union of its waves' extents (wave-uniform values, read back as broadcasts)
        __syncthreads();
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            int a0 = INT_MAX, a1 = INT_MIN, b0 = INT_MAX, b1 = INT_MIN;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                a0 = min(a0, xch[w][s][0]); a1 = max(a1, xch[w][s][1]);
                b0 = min(b0, xch[w][s][2]); b1 = max(b1, xch[w][s][3]);
            }
            lo_x[s] = __builtin_amdgcn_readfirstlane(a0); hi_x[s] = __builtin_amdgcn_readfirstlane(a1);
            lo_y[s] = __builtin_amdgcn_readfirstlane(b0); hi_y[s] = __builtin_amdgcn_readfirstlane(b1);
        }
    }
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        const int a0 = lo_x[s], a1 = hi_x[s], b0 = lo_y[s], b1 = hi_y[s];
        const bool empty = a1 < a0;
        const int al = (a1 + 4 >= W) ? (W & 3) : 0;
        const int ax = a0 - ((a0 - al) & 3);
        bx0[s] = empty ? 0 : ax; by0[s] = empty ? 0 : b0;
        bw[s] = empty ? 0 : a1 + 2 - ax;                   // columns from the aligned origin
        bh[s] = empty ? 0 : b1 - b0 + 2;
        fits = fits && (bw[s] <= BW) && (bh[s] <= R) && !(bx0[s] < 0 && (bx0[s] & 3) != 0);
    }

    BufRsrc rs[NSRC];
#pragma unroll
    for (int s = 0; s < NSRC; ++s)
        rs[s] = make_rsrc(p.src[s] + (size_t)b * CT * HW, (uint32_t)CT * (uint32_t)HW * 4u);
    const BufRsrc rref = make_rsrc(p.ref + (size_t)b * CT * HW, (uint32_t)CT * (uint32_t)HW * 4u);
    const size_t ostride = (size_t)p.D_out * HW;             // floats between channels of the output

    if (fits) {
        // DMA lane map.  Chunk k of a source box = 4 columns of (row, channel) with k = (row * 2 + channel) * C4 + column / 4;
        // lane l of DMA instruction j carries chunk 64 j + l to LDS byte 16 (64 j + l) of the box.  Chunks outside the box
        // rows, the image, or right of the last column any tap reads get an out-of-range offset = zeros.
        // Shared form: the NSRC * NI instructions of a step are dealt out over the workgroup's waves, instruction i = s * NI + j
        // to wave i % NW; a wave keeps the offsets, destination and descriptor of its own KD instructions only.
        constexpr int NDMA = NSRC * NI;
        constexpr int KD = SHARED ? (NDMA + NW - 1) / NW : 1;
        uint32_t vo[SHARED ? 1 : NSRC][SHARED ? 1 : NI];
        uint32_t vk[KD], dstk[KD];
        bool usek[KD];
        BufRsrc rsk[KD];
        if constexpr (SHARED) {
#pragma unroll
            for (int k = 0; k < KD; ++k) {
                const int i = k * NW + wave;                        // wave-uniform
                const int sk = min(i / NI, NSRC - 1), j = i - (i / NI) * NI;
                int bxs = bx0[0], bys = by0[0], bws = bw[0], bhs = bh[0];
                const float* sp = p.src[0];
#pragma unroll
                for (int t = 1; t < NSRC; ++t)
                    if (sk == t) { bxs = bx0[t]; bys = by0[t]; bws = bw[t]; bhs = bh[t]; sp = p.src[t]; }
                const int slot = 64 * j + lane;
                const int row = slot / (2 * C4), rem = slot - row * (2 * C4);
                const int ch = rem >= C4 ? 1 : 0, col = (rem - ch * C4) * 4;
                const int rel = (row * W + col) * 4 + ch * HW * 4;
                const int gx = bxs + col, gy = bys + row;
                const bool valid = (row < bhs) && (col < bws) && ((uint32_t)gy < (uint32_t)H) && (gx >= 0) && (gx + 4 <= W);
                vk[k] = valid ? (uint32_t)(rel + (bys * W + bxs) * 4) : SMVS_OOB;
                if (SMVS_ABLATE & 64) vk[k] = SMVS_OOB;
                if (SMVS_ABLATE & 128) vk[k] = valid ? (vk[k] & 0xfff0u) : SMVS_OOB;
                usek[k] = (i < NDMA) && (64 * j < bhs * 2 * C4);
                dstk[k] = (uint32_t)(sk * SRC_DW * 4 + j * 1024);
                rsk[k] = make_rsrc(sp + (size_t)b * CT * HW, (uint32_t)CT * (uint32_t)HW * 4u);
            }
        } else {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int slot = 64 * j + lane;
            const int row = slot / (2 * C4), rem = slot - row * (2 * C4);
            const int ch = rem >= C4 ? 1 : 0, col = (rem - ch * C4) * 4;
            const int rel = (row * W + col) * 4 + ch * HW * 4;
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const int gx = bx0[s] + col, gy = by0[s] + row;
                const bool valid = (row < bh[s]) && (col < bw[s]) && ((uint32_t)gy < (uint32_t)H) && (gx >= 0) && (gx + 4 <= W);
                vo[s][j] = valid ? (uint32_t)(rel + (by0[s] * W + bx0[s]) * 4) : SMVS_OOB;
                if (SMVS_ABLATE & 64) vo[s][j] = SMVS_OOB;
                if (SMVS_ABLATE & 128) vo[s][j] = valid ? (vo[s][j] & 0xfff0u) : SMVS_OOB;
            }
        }
        }
#pragma unroll
        for (int pl = 0; pl < DP; ++pl)
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const bool ok = (okmask >> (pl * NSRC + s)) & 1u;
                const int box0 = s * SRC_DW - ((by0[s] + 1) * (2 * BW) + bx0[s] + 1);      // wave-uniform
                const uint32_t a = tile_lds + 4u * (uint32_t)((int)txy[pl][s] + box0);
                const uint32_t z = tile_lds + 4u * (uint32_t)(ONE_BASE ? BUF_DW : DM_NBUF * BUF_DW);
                tap[pl][s].base[0] = ok ? a : z;
                tap[pl][s].base[1] = ONE_BASE ? 0u : ok ? a + 4u * (uint32_t)BUF_DW : z;
            }
        uint32_t ovo[DP];                                 // per-plane byte offset of this pixel inside one channel volume
#pragma unroll
        for (int pl = 0; pl < DP; ++pl)
            ovo[pl] = (active && pl < np && !(SMVS_ABLATE & 1)) ? (uint32_t)((SMVS_ABLATE & 16) ? ((dg + pl) & 3) : (dg + pl - p.d_begin + p.d_out_off)) * (uint32_t)HW * 4u + pix4
                                          : SMVS_OOB;       // inactive lanes / tail planes: store dropped by the range check

        // (s, j) are unrolled loop indices but not constant expressions: dispatch to the immediate-offset variant
        auto dma_at = [&](int s, int j, uint32_t buf, uint32_t voff, int so) {
#define SMVS_DMA_CASE(S, J) \
    case (S) * 8 + (J): dma_x4_to_lds_at<((S) * SRC_DW * 4 + (J) * 1024)>(rs[(S) < NSRC ? (S) : 0], buf, voff, so); break;
            static_assert(NI <= 8 && NSRC <= 7, "instruction dispatch");
            switch (s * 8 + j) {
                SMVS_DMA_CASE(0, 0) SMVS_DMA_CASE(0, 1) SMVS_DMA_CASE(0, 2) SMVS_DMA_CASE(0, 3) SMVS_DMA_CASE(0, 4) SMVS_DMA_CASE(0, 5) SMVS_DMA_CASE(0, 6) SMVS_DMA_CASE(0, 7)
                SMVS_DMA_CASE(1, 0) SMVS_DMA_CASE(1, 1) SMVS_DMA_CASE(1, 2) SMVS_DMA_CASE(1, 3) SMVS_DMA_CASE(1, 4) SMVS_DMA_CASE(1, 5) SMVS_DMA_CASE(1, 6) SMVS_DMA_CASE(1, 7)
                SMVS_DMA_CASE(2, 0) SMVS_DMA_CASE(2, 1) SMVS_DMA_CASE(2, 2) SMVS_DMA_CASE(2, 3) SMVS_DMA_CASE(2, 4) SMVS_DMA_CASE(2, 5) SMVS_DMA_CASE(2, 6) SMVS_DMA_CASE(2, 7)
                SMVS_DMA_CASE(3, 0) SMVS_DMA_CASE(3, 1) SMVS_DMA_CASE(3, 2) SMVS_DMA_CASE(3, 3) SMVS_DMA_CASE(3, 4) SMVS_DMA_CASE(3, 5) SMVS_DMA_CASE(3, 6) SMVS_DMA_CASE(3, 7)
                SMVS_DMA_CASE(4, 0) SMVS_DMA_CASE(4, 1) SMVS_DMA_CASE(4, 2) SMVS_DMA_CASE(4, 3) SMVS_DMA_CASE(4, 4) SMVS_DMA_CASE(4, 5) SMVS_DMA_CASE(4, 6) SMVS_DMA_CASE(4, 7)
                SMVS_DMA_CASE(5, 0) SMVS_DMA_CASE(5, 1) SMVS_DMA_CASE(5, 2) SMVS_DMA_CASE(5, 3) SMVS_DMA_CASE(5, 4) SMVS_DMA_CASE(5, 5) SMVS_DMA_CASE(5, 6) SMVS_DMA_CASE(5, 7)
                SMVS_DMA_CASE(6, 0) SMVS_DMA_CASE(6, 1) SMVS_DMA_CASE(6, 2) SMVS_DMA_CASE(6, 3) SMVS_DMA_CASE(6, 4) SMVS_DMA_CASE(6, 5) SMVS_DMA_CASE(6, 6) SMVS_DMA_CASE(6, 7)
                default: break;
            }
#undef SMVS_DMA_CASE
        };
        int ni[NSRC];                                      // instructions that carry rows of the box (wave-uniform)
#pragma unroll
        for (int s = 0; s < NSRC; ++s) ni[s] = (bh[s] * 2 * C4 + 63) >> 6;
        auto issue_dma = [&](int st) {
            if (SMVS_ABLATE & 2) return;
            const uint32_t buf = tile_lds + (uint32_t)((st & 1) * BUFP_DW * 4);
            const int choff = 2 * st * HW * 4;
            if constexpr (SHARED) {
#pragma unroll
                for (int k = 0; k < KD; ++k)
                    if (usek[k]) dma_x4_to_lds(rsk[k], buf + dstk[k], vk[k], choff);      // wave-uniform
            } else {
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        // destination = buf + a compile-time offset, formed inside the asm by one s_add into M0
                        if (j < ni[s]) dma_at(s, j, buf, vo[s][j], choff);        // wave-uniform
                    }
                }
            }
        };

        SMVS_T(const unsigned long long t_setup = now();)
        // prologue: pair 0
        // ref feature pairs run two steps ahead of their use, in registers
        f32x2 ref0, ref1;
        ref0.x = llvm_raw_buffer_load_f32(rref.v, (int)pix4, 0, 0);
        ref0.y = llvm_raw_buffer_load_f32(rref.v, (int)pix4, HW * 4, 0);
        ref1.x = llvm_raw_buffer_load_f32(rref.v, (int)pix4, (NSTEP > 1 ? 2 : 0) * HW * 4, 0);
        ref1.y = llvm_raw_buffer_load_f32(rref.v, (int)pix4, (NSTEP > 1 ? 3 : 1) * HW * 4, 0);
        issue_dma(0);

        // One step = one channel pair.  PAR (buffer parity) is a compile-time constant so that the
        // staging buffer enters every LDS read as an immediate offset.
        auto step = [&](int st, auto par_tag) {
            constexpr int PAR = decltype(par_tag)::value;
            // Issue order behind DMA(st): the two ref loads of step st+1 and the 2*DP stores of step
            // st-1 (every step issues all of them; lanes/planes without output carry an out-of-range
            // offset).  vmcnt retires in order, so DMA(st) has landed once at most that many
            // operations are outstanding.
            const f32x2 refc = ref0;
            SMVS_T(const unsigned long long tw0 = now();)
            if (SMVS_ABLATE & 256) { if (st == 0) wait_vmcnt<0>(); }      // profiling: the landing of DMA(st) is not waited for
            else if (st == 0) wait_vmcnt<0>();
            else if (st + 1 < NSTEP) wait_vmcnt<2 * DP + 2>();
            else wait_vmcnt<2 * DP>();
            // shared form: my part of DMA(st) has landed -> so has everybody's, and everybody has finished step st-1, whose
            // buffer DMA(st+1) is about to overwrite.  (A wave's LDS reads of step st-1 have all returned: its last unit
            // waited for lgkmcnt(0).)
            if constexpr (SHARED) { if (!(SMVS_ABLATE & 512)) asm volatile("s_barrier" ::: "memory"); }
            SMVS_T(const unsigned long long tw1 = now(); t_vm += tw1 - tw0;)
            ref0 = ref1;
            if (st + 1 < NSTEP) {
                issue_dma(st + 1);
                SMVS_T(t_dma += now() - tw1;)
                const int nx = (st + 2 < NSTEP) ? 2 * st + 4 : 0;   // dummy reload keeps the count constant
                ref1.x = llvm_raw_buffer_load_f32(rref.v, (int)pix4, nx * HW * 4, 0);
                ref1.y = llvm_raw_buffer_load_f32(rref.v, (int)pix4, (nx + 1) * HW * 4, 0);
            }
            const f32x2 refsq = refc * refc;
            // one descriptor per channel pair: base = channel 2*st of the output, second channel
            // through the scalar offset
            const BufRsrc ro = make_rsrc(p.out + ((size_t)b * CT + 2 * st) * ostride, (uint32_t)(2 * ostride * 4));
            const int och1 = (int)(ostride * 4);
            // Software pipeline over UNITS = (plane, group of US sources): the taps of unit u+1 are in flight (ds_read2_b32
            // from inline asm, in-order return, counted lgkmcnt) while unit u is being accumulated.  Two sources per unit
            // when the source count is even, one otherwise; a plane's sum / sum of squares run across its units.  The packed
            // arithmetic is the hand-ordered blocks of smvs_device.h (pk_bilinear / pk_accumulate / pk_finish / pk_variance:
            // no instruction reads a packed result in the slot right behind its producer); a plane's last instruction
            // (var = meansq - mean^2) and its two stores are issued behind the bilinear block of the NEXT unit.
            constexpr int US = (NSRC % 2 == 0) ? 2 : 1, UPP = NSRC / US, NU = DP * UPP;
            f32x2 cv[2][US][4];
            f32x2 accS[2], accT[2];                         // partial sums of a plane between its units (3+ sources)
            f32x2 S, T;                                     // mean^2 and meansq of the plane whose stores are pending
            const f32x2 rvv = {rV, fV};
            auto read_unit = [&](int u) {
                const int pl = u / UPP, s0 = (u % UPP) * US;
                if (SMVS_ABLATE & 8) {
#pragma unroll
                    for (int k = 0; k < US; ++k) cv[u & 1][k][0] = cv[u & 1][k][1] = cv[u & 1][k][2] = cv[u & 1][k][3] = tap[pl][s0 + k].wn;
                    return;
                }
#pragma unroll
                for (int k = 0; k < US; ++k)
                    lds_read_tap_planar<BW>(ONE_BASE ? tap[pl][s0 + k].base[0] + (uint32_t)(PAR * BUFP_DW * 4) : tap[pl][s0 + k].base[PAR],
                                            cv[u & 1][k][0], cv[u & 1][k][1], cv[u & 1][k][2], cv[u & 1][k][3]);
            };
            auto store_plane = [&](int pl, f32x2 var) {
                SMVS_T(const unsigned long long ts0 = now();)
                llvm_raw_buffer_store_f32(var.x, ro.v, (int)ovo[pl], 0, STORE_AUX);
                llvm_raw_buffer_store_f32(var.y, ro.v, (int)ovo[pl], och1, STORE_AUX);
                SMVS_T(t_st += now() - ts0;)
            };
            if constexpr (AR == AR_FUSED && !(SMVS_ABLATE & (8 | 32))) {
                // Fused arithmetic: differences to the ref feature straight out of the bilinear chains (weights and ref
                // pre-scaled), variance in 3 (2 sources) / 1 (1 source) / 2 S + 1 (S sources) packed operations; the tail
                // of a unit rides in the gaps of the next unit's bilinear block (smvs_device.h).
                const f32x2 kv = {p.kw, p.kw};
                const f32x2 rk = pk_scale_lo(refc, kv);
                read_unit(0);
                if constexpr (US == 2) {
                    f32x2 ab[2][2], var;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        if (u + 1 < NU) read_unit(u + 1);
                        f32x2 (&c)[US][4] = cv[u & 1];
                        if (u + 1 < NU) lds_wait<8>(c[0][0], c[0][1], c[0][2], c[0][3], c[1][0], c[1][1], c[1][2], c[1][3]);
                        else            lds_wait<0>(c[0][0], c[0][1], c[0][2], c[0][3], c[1][0], c[1][1], c[1][2], c[1][3]);
                        const int pl = u / UPP, s0 = (u % UPP) * US, kp = (u + UPP - 1) % UPP;     // kp: position of the previous unit in its plane
                        f32x2 (&n)[2] = ab[u & 1];
                        f32x2 (&o)[2] = ab[(u & 1) ^ 1];
#define SMVS_FB_CALL c[0][0], c[0][1], c[0][2], c[0][3], tap[pl][s0].wn, tap[pl][s0].ws, c[1][0], c[1][1], c[1][2], c[1][3], tap[pl][s0 + 1].wn, tap[pl][s0 + 1].ws, rk
                        if (u == 0) pk_fbil2(n[0], n[1], SMVS_FB_CALL);
                        else if constexpr (UPP == 1) { pk_fbil2_var3(n[0], n[1], var, o[0], o[1], SMVS_FB_CALL); store_plane(pl - 1, var); }
                        else if (kp == 0) pk_fbil2_acc0(n[0], n[1], S, T, o[0], o[1], SMVS_FB_CALL);
                        else if (kp < UPP - 1) pk_fbil2_acc(n[0], n[1], S, T, o[0], o[1], SMVS_FB_CALL);
                        else { pk_fbil2_fin(n[0], n[1], var, S, T, o[0], o[1], rvv, SMVS_FB_CALL); store_plane(pl - 1, var); }
#undef SMVS_FB_CALL
                    }
                    f32x2 (&l)[2] = ab[(NU - 1) & 1];
                    if constexpr (UPP == 1) store_plane(DP - 1, pk_var3_tail(l[0], l[1]));
                    else                    store_plane(DP - 1, pk_fin_tail(S, T, l[0], l[1], rvv));
                } else {
                    // odd source counts: one source per unit, compiler-scheduled packed operations (same operation sequence
                    // as fused_variance<NSRC>)
                    f32x2 dS = (f32x2)(0.0f), dT = (f32x2)(0.0f);
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        if (u + 1 < NU) read_unit(u + 1);
                        f32x2 (&c)[US][4] = cv[u & 1];
                        f32x2 d0, d1, d2, d3;
                        d0 = d1 = d2 = d3 = (f32x2)(0.0f);
                        if (u + 1 < NU) lds_wait<4>(c[0][0], c[0][1], c[0][2], c[0][3], d0, d1, d2, d3);
                        else            lds_wait<0>(c[0][0], c[0][1], c[0][2], c[0][3], d0, d1, d2, d3);
                        const int pl = u / UPP, k = u % UPP;
                        const TapD& t = tap[pl][k];
                        f32x2 d = __builtin_elementwise_fma(c[0][0], (f32x2)(t.wn.x), -rk);
                        d = __builtin_elementwise_fma(c[0][1], (f32x2)(t.wn.y), d);
                        d = __builtin_elementwise_fma(c[0][2], (f32x2)(t.ws.x), d);
                        d = __builtin_elementwise_fma(c[0][3], (f32x2)(t.ws.y), d);
                        if (k == 0) { dS = d; dT = d * d; }
                        else { dS = dS + d; dT = __builtin_elementwise_fma(d, d, dT); }
                        if (k == UPP - 1) {
                            f32x2 var;
                            if constexpr (NSRC == 1) var = dT;
                            else { const f32x2 pq = dS * rV; var = __builtin_elementwise_fma(-pq, dS, dT); }
                            asm volatile("s_nop 0" : "+v"(var));       // the packed result is not read in the slot behind its producer
                            store_plane(pl, var);
                        }
                    }
                }
                return;
            }
            read_unit(0);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (u + 1 < NU) read_unit(u + 1);
                // reads per unit = 4*US; everything older than the next unit's reads has returned
                f32x2 (&c)[US][4] = cv[u & 1];
                if (SMVS_ABLATE & 8) {
                } else if constexpr (US == 1) {
                    f32x2 d0, d1, d2, d3;
                    d0 = d1 = d2 = d3 = (f32x2)(0.0f);
                    if (u + 1 < NU) lds_wait<4>(c[0][0], c[0][1], c[0][2], c[0][3], d0, d1, d2, d3);
                    else            lds_wait<0>(c[0][0], c[0][1], c[0][2], c[0][3], d0, d1, d2, d3);
                } else {
                    if (u + 1 < NU) lds_wait<8>(c[0][0], c[0][1], c[0][2], c[0][3], c[1][0], c[1][1], c[1][2], c[1][3]);
                    else            lds_wait<0>(c[0][0], c[0][1], c[0][2], c[0][3], c[1][0], c[1][1], c[1][2], c[1][3]);
                }
                const int pl = u / UPP, k = u % UPP, s0 = k * US;
                f32x2 a, b;
                if (SMVS_ABLATE & 32) {
                    if (u > 0 && k == 0) store_plane(pl - 1, c[0][0]);
                    continue;
                }
                if constexpr (US == 2) pk_bilinear2(a, b, c[0][0], c[0][1], c[0][2], c[0][3], tap[pl][s0].wn, tap[pl][s0].ws,
                                                    c[1][0], c[1][1], c[1][2], c[1][3], tap[pl][s0 + 1].wn, tap[pl][s0 + 1].ws);
                else                   pk_bilinear1(a, c[0][0], c[0][1], c[0][2], c[0][3], tap[pl][s0].wn, tap[pl][s0].ws);
                if (u > 0 && k == 0) store_plane(pl - 1, pk_variance<0>(T, S));      // the previous plane leaves under this unit
                const f32x2& sin = (k == 0) ? refc : accS[(k - 1) & 1];
                const f32x2& tin = (k == 0) ? refsq : accT[(k - 1) & 1];
                if (k == UPP - 1) {
                    if constexpr (US == 2) pk_finish2(S, T, sin, tin, a, b, rvv);
                    else                   pk_finish1(S, T, sin, tin, a, rvv);
                } else {
                    if constexpr (US == 2) pk_accumulate2(accS[k & 1], accT[k & 1], sin, tin, a, b);
                    else                   pk_accumulate1(accS[k & 1], accT[k & 1], sin, tin, a);
                }
            }
            if (SMVS_ABLATE & 32) store_plane(DP - 1, cv[0][0][1]);
            else store_plane(DP - 1, pk_variance<1>(T, S));
        };
        static_assert(NSTEP % 2 == 0, "two steps per loop iteration");
        for (int st = 0; st < NSTEP; st += 2) {
            step(st, std::integral_constant<int, 0>());
            step(st + 1, std::integral_constant<int, 1>());
        }
#ifdef SMVS_TIMING
        {
            const unsigned long long t_end = now();
            if (lane == 0) {
                atomicAdd(&smvs_timing[0], t_geo - t_start); atomicAdd(&smvs_timing[1], t_setup - t_geo);
                atomicAdd(&smvs_timing[2], t_end - t_setup); atomicAdd(&smvs_timing[3], t_vm);
                atomicAdd(&smvs_timing[5], t_dma); atomicAdd(&smvs_timing[4], t_st); atomicAdd(&smvs_timing[7], 1ull);
                atomicAdd(&smvs_timing[8], t_chk - t_start); atomicAdd(&smvs_timing[9], t_ref - t_chk); atomicAdd(&smvs_timing[10], t_geo - t_ref);
            }
        }
#endif
    } else {
        // ---- fallback: direct gathers for this plane group (box larger than the staged tile).
        //      Rare and wave-uniform; the source taps are rebuilt plane by plane from the ground point
        //      of phase A in a rolled loop (same float64 values as the staged path would have used).
        SMVS_T(if (lane == 0) atomicAdd(&smvs_timing[6], 1ull);)      // waves that took the fallback
        const float* refp = p.ref + (size_t)b * CT * HW + pix;
        float* outp = p.out + (size_t)b * CT * p.D_out * HW + pix;
#pragma unroll 1
        for (int pl = 0; pl < np; ++pl) {
            const int d = dg + pl;
            float hfp = hf[0]; double latp = lat[0], lonp = lon[0];
#pragma unroll
            for (int k = 1; k < DP; ++k) if (pl == k) { hfp = hf[k]; latp = lat[k]; lonp = lon[k]; }
            const double h = (double)hfp;
            const cgeo_t geo_d = launder(geo_b);
            Tap tp[NSRC];
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                if (GEO == 0) {
                    double samp, line;
                    if (use_pc) {
                        o2p_pc_xn<1>(geo_d + (size_t)(s + 1) * RPC_LEN, src_n[s], &latp, &lonp,
                                     pc_co + pc_offset(b, s, 0, d, NSRC, p.D), (size_t)p.D * PC_PER_CUBIC, &samp, &line);
                    } else {
                        o2p_xn<1>(geo_d + (size_t)(s + 1) * RPC_LEN, src_n[s], &latp, &lonp, &h, &samp, &line);
                    }
                    tp[s] = tap_from_pixel((float)samp, (float)line, H, W, half_wm1, half_hm1);
                } else {
                    const cgeo_t P = geo_d + s * 16;
                    const double rx = fma(P[1], fy, P[0] * fx) + P[2];
                    const double ry = fma(P[5], fy, P[4] * fx) + P[6];
                    const double rz = fma(P[9], fy, P[8] * fx) + P[10];
                    const double X = fma(rx, h, P[3]), Y = fma(ry, h, P[7]), Z = fma(rz, h, P[11]);
                    tp[s] = tap_from_grid((float)((X / Z) / ((W - 1) * 0.5) - 1.0),
                                          (float)((Y / Z) / ((H - 1) * 0.5) - 1.0), H, W);
                }
            }
            float* od = outp + (size_t)(d - p.d_begin + p.d_out_off) * HW;
#pragma unroll 1
            for (int c = 0; c < CT; ++c) {
                const float r = refp[(size_t)c * HW];
                if constexpr (AR == AR_FUSED) {
                    const float rk = r * p.kw;
                    float df[NSRC];
#pragma unroll
                    for (int s = 0; s < NSRC; ++s) df[s] = fused_tap_diff(rs[s], tp[s], c * HW * 4, p.kw, rk);
                    if (active) od[(size_t)c * ostride] = fused_variance<NSRC>(df, rV);
                    continue;
                }
                float sum = r;
                float sq = r * r;
#pragma unroll
                for (int s = 0; s < NSRC; ++s) {
                    const float wv = tap_fetch(rs[s], tp[s], c * HW * 4);
                    sum = sum + wv;
                    sq = sq + wv * wv;
                }
                const float m = div_by_views(sum, fV, rV);
                const float q = div_by_views(sq, fV, rV);
                if (active) od[(size_t)c * ostride] = q - m * m;
            }
        }
    }
}

// Kernel choice.  The staged kernel serves 2-8 views at C = 8/16/32 (one channel volume of the output < 2 GiB);
// everything else (and SMVS_COSTVOL_DIRECT=1 in tuning builds) takes the direct-gather kernel.  Both produce
// identical bits.  Planes per wave (DP): a whole sweep is cut into groups of 8 (2-3 views) or 4 planes (the taps of a
// group live in registers; with 3-4 sources the LDS tiles allow two workgroups per CU, i.e. 256 VGPRs per lane), the
// plane-at-a-time launches of the pred loop get DP = 1 so that no float64 work is spent on planes that are not stored.
enum { K_DIRECT = 0, K_DMA = 2 };

static int kernel_choice()
{
    return tune_int("SMVS_COSTVOL_DIRECT", 0) == 1 ? K_DIRECT : K_DMA;      // A/B switch (tuning builds only)
}

// Order of the workgroups of one band of rows (an XCD sweeps a run of consecutive workgroups, xcd_remap): x tile fastest, then
// plane chunk -- every plane chunk walks the band's source rows once more and finds them in that XCD's L2 (4 MiB) as long as the
// band's staged rows fit: rows x sources x channels x W x 4 B = 1 MB at the metric shape (traffic 1.007 x algorithmic).  At
// 5 views x 1536 columns the band is 3.9 MB: every chunk missed and the features came from HBM once per plane chunk (FETCH_SIZE
// 1.58 GB against 0.755 GB for the 8-plane shard, 12 GB for the 64-plane sweep; profiles/r04_cfg4_summary.txt).  Above 1.5 MB
// the plane chunk runs fastest instead: all plane chunks of an x tile are in flight together and share ~0.1 MB.
#ifndef SMVS_CHUNK_MAJOR
#define SMVS_CHUNK_MAJOR (-1)         // A/B switch of profiling builds: 0 / 1 force an order
#endif
#ifndef SMVS_GROUP_ROWS
#define SMVS_GROUP_ROWS 8            // tile rows per group of the chunk-major order (shared-box form; A/B switch of profiling builds)
#endif
static int launch_order(const CostVolParams& p, int rows, int nsrc)
{
    if (SMVS_CHUNK_MAJOR >= 0) return SMVS_CHUNK_MAJOR;
    return (long long)rows * nsrc * p.C * p.W * 4 > 1536 * 1024 ? 1 : 0;
}

template <int GEO, int NSRC, int DP, int AR>
static hipError_t launch_staged(CostVolParams p, hipStream_t st)
{
    const int nd = p.d_end - p.d_begin;
    p.xt = (p.W + WV_TX - 1) / WV_TX;
    p.yt = (p.H + WV_TY * WV_WAVES - 1) / (WV_TY * WV_WAVES);
    p.dch = DP;
    p.dct = (nd + DP - 1) / DP;
    p.chunk_major = launch_order(p, WV_TY * WV_WAVES + 3, NSRC);
    const long long nb = (long long)p.xt * p.yt * p.dct * p.B;
    if (nb >= (1ll << 31)) return hipErrorInvalidValue;
    dim3 blk(64 * WV_WAVES), grd((unsigned)nb);
#ifdef SMVS_ONLY_BENCH
    // profiling builds: only the instances the headline bench launches (seconds instead of a minute to compile)
    if constexpr (GEO == 0 && (NSRC == 2 || NSRC == 4) && DP >= 2) { if (p.C == 32) hipLaunchKernelGGL((costvol_dma_kernel<GEO, NSRC, 32, DP, AR>), grd, blk, 0, st, p); }
    return hipGetLastError();
#else
    switch (p.C) {
    case 8:  hipLaunchKernelGGL((costvol_dma_kernel<GEO, NSRC, 8, DP, AR>), grd, blk, 0, st, p); break;
    case 16: hipLaunchKernelGGL((costvol_dma_kernel<GEO, NSRC, 16, DP, AR>), grd, blk, 0, st, p); break;
    default: hipLaunchKernelGGL((costvol_dma_kernel<GEO, NSRC, 32, DP, AR>), grd, blk, 0, st, p); break;
    }
    return hipGetLastError();
#endif
}

// Shared-box form: a workgroup = NWY x NWP waves = 32 x 2 NWY pixels x DP NWP planes
template <int GEO, int NSRC, int DP, int AR, int NWY, int NWP, int WPS>
static hipError_t launch_shared(CostVolParams p, hipStream_t st)
{
    const int nd = p.d_end - p.d_begin;
    p.xt = (p.W + WV_TX - 1) / WV_TX;
    p.yt = (p.H + WV_TY * NWY - 1) / (WV_TY * NWY);
    p.dch = DP;
    p.dct = (nd + DP * NWP - 1) / (DP * NWP);
    p.chunk_major = launch_order(p, WV_TY * NWY + 3, NSRC);
    p.group_rows = SMVS_GROUP_ROWS;
    const long long nb = (long long)p.xt * p.yt * p.dct * p.B;
    if (nb >= (1ll << 31)) return hipErrorInvalidValue;
    dim3 blk(64 * NWY * NWP), grd((unsigned)nb);
#ifdef SMVS_ONLY_BENCH
    if (p.C == 32) hipLaunchKernelGGL((costvol_dma_kernel<GEO, NSRC, 32, DP, AR, NWY, NWP, WPS>), grd, blk, 0, st, p);
    return hipGetLastError();
#else
    switch (p.C) {
    case 8:  hipLaunchKernelGGL((costvol_dma_kernel<GEO, NSRC, 8, DP, AR, NWY, NWP, WPS>), grd, blk, 0, st, p); break;
    case 16: hipLaunchKernelGGL((costvol_dma_kernel<GEO, NSRC, 16, DP, AR, NWY, NWP, WPS>), grd, blk, 0, st, p); break;
    default: hipLaunchKernelGGL((costvol_dma_kernel<GEO, NSRC, 32, DP, AR, NWY, NWP, WPS>), grd, blk, 0, st, p); break;
    }
    return hipGetLastError();
#endif
}

// shared-box configuration of the 3-4 source sweeps (4+ planes): planes per wave, waves in y, waves along planes, waves per SIMD
#ifndef SMVS_N34_SHARED
#define SMVS_N34_SHARED 1
#endif
#ifndef SMVS_N34_DP
#define SMVS_N34_DP 4
#endif
#ifndef SMVS_N34_NWY
#define SMVS_N34_NWY 2
#endif
#ifndef SMVS_N34_NWP
#define SMVS_N34_NWP 2
#endif
#ifndef SMVS_N34_WPS
#define SMVS_N34_WPS 2
#endif
// ... and of the 1-2 source sweeps that divide into eights at C = 32 (A/B switch; 0 = every wave its own box)
#ifndef SMVS_N2_SHARED
#define SMVS_N2_SHARED 0
#endif
#ifndef SMVS_N2_DP
#define SMVS_N2_DP 8
#endif
#ifndef SMVS_N2_NWY
#define SMVS_N2_NWY 2
#endif
#ifndef SMVS_N2_NWP
#define SMVS_N2_NWP 1
#endif
#ifndef SMVS_N2_WPS
#define SMVS_N2_WPS 2
#endif

template <int GEO, int NSRC, int AR>
static hipError_t launch_ct(CostVolParams p, hipStream_t st)
{
    const int nd = p.d_end - p.d_begin;
    {
        // one channel volume of the output < 2 GiB: the store descriptor spans two of them (num_records is 32-bit)
        // and 2^31 is the offset that marks a dropped store; 2 <= W,H < 65535: packed tap coordinates and the
        // exact-division argument of div_half_int
        const bool staged_ok = (p.C == 8 || p.C == 16 || p.C == 32) && p.W >= 2 && p.H >= 2 && p.W < 65535 && p.H < 65535 &&
                               (long long)p.D_out * p.H * p.W * 4 < (1ll << 31);
        if (kernel_choice() != K_DIRECT && staged_ok) {
            if (nd == 1) return launch_staged<GEO, NSRC, 1, AR>(p, st);
            // 6-8 views: 4 planes x 5-7 sources of tap state need more than 256 registers (one wave per SIMD) -> 2 planes per wave
            if (nd == 2 || NSRC > 4) return launch_staged<GEO, NSRC, 2, AR>(p, st);
            // 8 planes per wave when the sweep divides into eights (48 / 32 / 8 / 64-plane sweeps): half the staging DMA per
            // voxel and the ref view's plane-invariant part amortised over twice the planes outweigh the drop to two
            // waves per SIMD (219 VGPRs) -- measured 0.699 vs 0.717 ms at the metric shape; at C = 8 (float64-bound) 4 planes per
            // wave stay faster (0.053 vs 0.061 ms)
#if SMVS_N2_SHARED
            if constexpr (NSRC <= 2 && GEO == 0) { if (nd % 8 == 0 && p.C == 32) return launch_shared<GEO, NSRC, SMVS_N2_DP, AR, SMVS_N2_NWY, SMVS_N2_NWP, SMVS_N2_WPS>(p, st); }
#endif
#if SMVS_N34_SHARED
            if constexpr (NSRC > 2 && NSRC <= 4) { if (nd % (SMVS_N34_DP * SMVS_N34_NWP) == 0) return launch_shared<GEO, NSRC, SMVS_N34_DP, AR, SMVS_N34_NWY, SMVS_N34_NWP, SMVS_N34_WPS>(p, st); }
#endif
#if SMVS_DP8
            if constexpr (NSRC <= 2 && (GEO == 0 || SMVS_DP8_HOMO)) { if (nd % 8 == 0 && p.C >= SMVS_DP8_MINC) return launch_staged<GEO, NSRC, 8, AR>(p, st); }
#endif
            if constexpr (NSRC > 2 && NSRC <= 4 && SMVS_NSRC34_DP == 2) return launch_staged<GEO, NSRC, 2, AR>(p, st);
            if constexpr (NSRC <= 4) return launch_staged<GEO, NSRC, 4, AR>(p, st);
        }
    }
    p.xt = (p.W + TILE_X - 1) / TILE_X;
    p.yt = (p.H + TILE_Y - 1) / TILE_Y;
    p.dch = nd < 8 ? nd : 8;
    p.dct = (nd + p.dch - 1) / p.dch;
    const long long nblocks = (long long)p.xt * p.yt * p.dct * p.B;
    if (nblocks >= (1ll << 31)) return hipErrorInvalidValue;
    dim3 blk(TILE_X, TILE_Y);
#ifndef SMVS_ONLY_BENCH
    switch (p.C) {
    case 8:  hipLaunchKernelGGL((costvol_fwd_kernel<GEO, NSRC, 8, AR>), dim3(nblocks), blk, 0, st, p); break;
    case 16: hipLaunchKernelGGL((costvol_fwd_kernel<GEO, NSRC, 16, AR>), dim3(nblocks), blk, 0, st, p); break;
    case 32: hipLaunchKernelGGL((costvol_fwd_kernel<GEO, NSRC, 32, AR>), dim3(nblocks), blk, 0, st, p); break;
    default: hipLaunchKernelGGL((costvol_fwd_kernel<GEO, NSRC, 0, AR>), dim3(nblocks), blk, 0, st, p); break;
    }
#endif
    return hipGetLastError();
}

template <int GEO, int AR>
static hipError_t launch_nsrc(const CostVolParams& p, hipStream_t st)
{
    switch (p.V - 1) {
    case 1: return launch_ct<GEO, 1, AR>(p, st);
    case 2: return launch_ct<GEO, 2, AR>(p, st);
    case 3: return launch_ct<GEO, 3, AR>(p, st);
    case 4: return launch_ct<GEO, 4, AR>(p, st);
    case 5: return launch_ct<GEO, 5, AR>(p, st);
    case 6: return launch_ct<GEO, 6, AR>(p, st);
    case 7: return launch_ct<GEO, 7, AR>(p, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace smvs
