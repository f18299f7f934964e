// conv_wgrad.hip -- weight (and bias) gradient of the 3x3, stride-1, pad-1 convolutions of the ConvGRU cells (training path).
//
// Replaces, for gate_conv / output_conv of ConvGRUCell2 (/root/reference/modules/module.py:13-14, :24, :47 under
// loss.backward(), /root/reference/train.py:284), what autograd gets from MIOpen on this image: per call an im2col, two to four
// layout transposes, an implicit-GEMM weight-gradient kernel and a col2im / reduction -- ~40 % of the training step's convolution
// time for ~700 calls per step (profiles/r04_train_step.txt).
//     dW[co][ci][ky][kx] = sum_{b,y,x} dY[b][co][y][x] * X[b][ci][y+ky-1][x+kx-1]        db[co] = sum dY[b][co][y][x]
//
// One WAVE owns (a pair of input channels) x (a group of 8 output channels) x (a 64-pixel-wide column strip) x (a range of rows)
// and keeps its 2 x 8 x 9 = 144 partial sums in registers, lane = column: per row it loads the new south row of the two input
// channels (3 shifted dwords each: the 3x3 window slides down through registers, every X row is fetched once per wave) and the 8
// gradient values, and issues 144 FMAs -- 14 coalesced loads per 144 FMAs, no LDS.  Columns outside the image carry an
// out-of-range offset (the buffer range check returns 0 = zero padding), rows outside are skipped wave-uniformly.  At the end the
// 144 sums are reduced over the 64 lanes on the DPP network and lane 63 adds them to dW with float atomics (the caller zeroes
// dW): the association differs from MIOpen's like any split-K GEMM's does.
// The bias gradient rides along in the waves of input-channel pair 0.
#include <type_traits>

#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

// Generalised in round 4 to every 3x3 layer of the RED regulariser: the WINDOW tensor (`x`: Cin channels, read through the 3x3 taps at
// stride S) and the GRADIENT-side tensor (`dy`: Cout channels, one value per position of the H x W grid) swap roles for the transposed
// convolutions -- conv (stride 1 / 2): window = layer input, grid = output gradient, dw (Cout,Cin,3,3); ConvTranspose2d (stride 2 or 1,
// pad 1): window = output gradient, grid = layer input, weight (Cin_layer, Cout_layer, 3, 3) = the same index formula.
constexpr int WGRAD_LIST_MAX = 64;
struct WgradParams {
    const float* x; const float* dy; float* dw; float* db;
    const float* x2; int CA;            // window tensor given as two operands: channels [0, CA) in x (B,CA,..), the rest in x2 (B,Cin-CA,..); x2 null: all in x (CA = Cin)
    int B, Cin, Cout, H, W;             // H, W: the grid; the window tensor is (B, Cin, S*H, S*W)
    int ncp, ncog, nxs, nrc, rows;      // input-channel pairs, output-channel groups of 8, column strips, row chunks, rows per chunk
    int nbc, bchunk;                    // batch chunks; a wave walks samples [bc * bchunk, min(B, (bc + 1) * bchunk)) with its sums in registers
    // The batch as a LIST of nlist tensors of Bper samples each (B = nlist * Bper): the planes of a training step's plane loop, whose
    // activations and output gradients are separate allocations -- one launch per layer and step instead of one per layer and plane
    // (a launch on these shapes is a 13 us latency floor; ~1 300 per training step of the 48/32/8 cascade).
    int nlist, Bper;
    const float* xl[WGRAD_LIST_MAX]; const float* x2l[WGRAD_LIST_MAX]; const float* dyl[WGRAD_LIST_MAX];
};

// Four registers -> one: the sums over the 64 lanes of a, b, c, d end up in lanes 15 (a), 31 (c), 47 (b), 63 (d) of the result.
// v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves / odd-even 16-lane rows between two registers, so one swap +
// one add folds two registers into one whose halves (rows) carry the partial sums of different values; the last four steps run
// inside the 16-lane rows on the DPP network.  152 values cost 38 x (3 swaps + 3 adds + 4 DPP adds) instead of 152 x 6 DPP adds,
// and leave 38 atomics with four active lanes instead of 152 with one.
__device__ __forceinline__ float reduce4_rows(float a, float b, float c, float d)
{
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %1\n\t"          // a = {a.lo, b.lo}, b = {a.hi, b.hi}
                 "v_permlane32_swap_b32 %2, %3\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32 %0, %0, %1\n\t"                  // lanes 0-31: sum pairs of a, lanes 32-63: of b
                 "v_add_f32 %2, %2, %3\n\t"                  // likewise c | d
                 "s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %2\n\t"          // %0 = {ab.row0, cd.row0, ab.row2, cd.row2}, %2 = {ab.row1, cd.row1, ab.row3, cd.row3}
                 "s_nop 1\n\t"
                 "v_add_f32 %0, %0, %2\n\t"                  // row 0: a, row 1: c, row 2: b, row 3: d -- 16 partial sums each
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\ts_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    return a;
}

#ifndef SMVS_WGRAD2_XCD
#define SMVS_WGRAD2_XCD 0               // sharers adjacent + one contiguous run of workgroups per XCD (the 3-D kernel's order).  Measured in round 5 on the casred graphed step, alternating: 58.6 / 58.1 against 59.0 / 58.2 ms -- inside the spread; off
#endif

template <int S>
__global__ __launch_bounds__(256)
void conv3x3_wgrad_kernel(const WgradParams p)
{
    const int lane = threadIdx.x & 63;
#if SMVS_WGRAD2_XCD
    // (round 5, from the 3-D kernel below) the waves that read the same rows -- every channel pair wants a gradient row, every output
    // group a window row -- run fastest in the unit order and every XCD gets one contiguous run of workgroups: sharers meet in one L2
    int unit = (int)xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);   // one wave = one unit
    unit = __builtin_amdgcn_readfirstlane(unit);
    const int total = p.ncp * p.ncog * p.nxs * p.nrc * p.nbc;
    if (unit >= total) return;
    const int cp = unit % p.ncp; unit /= p.ncp;
    const int cog = unit % p.ncog; unit /= p.ncog;
    const int rc = unit % p.nrc; unit /= p.nrc;
    const int xs = unit % p.nxs;
    const int bc = unit / p.nxs;
#else
    int unit = blockIdx.x * 4 + (threadIdx.x >> 6);                  // one wave = one unit
    unit = __builtin_amdgcn_readfirstlane(unit);
    const int total = p.ncp * p.ncog * p.nxs * p.nrc * p.nbc;
    if (unit >= total) return;
    // row chunk fastest, then column strip, channel pair, output group, batch: neighbouring waves share X / dY rows in L2
    const int rc = unit % p.nrc; unit /= p.nrc;
    const int xs = unit % p.nxs; unit /= p.nxs;
    const int cp = unit % p.ncp; unit /= p.ncp;
    const int cog = unit % p.ncog;
    const int bc = unit / p.ncog;
#endif
    const int b0 = bc * p.bchunk, b1 = min(p.B, b0 + p.bchunk);
    const int H = p.H, W = p.W, HW = H * W;
    const int HX = S * H, WX = S * W, HWX = HX * WX;                  // the window tensor's plane
    const int x = xs * 64 + lane;
    const int y0 = rc * p.rows, y1 = min(y0 + p.rows, H);
    const int ci0 = 2 * cp;
    const bool two = ci0 + 1 < p.Cin;                                 // odd channel counts: the last pair has one member
    // column offsets of the three taps inside a row (bytes), out-of-range where the column is outside the image
    uint32_t cx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int xx = S * x - 1 + k;
        cx[k] = (x < W && xx >= 0 && xx < WX) ? (uint32_t)xx * 4u : SMVS_OOB;
    }
    const uint32_t cy = x < W ? (uint32_t)x * 4u : SMVS_OOB;
    const bool inA = ci0 < p.CA;                                      // (CA is even when there is a second operand: a pair never straddles)
    const int nco = min(8, p.Cout - cog * 8);

    float acc[2][8][9];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[c][j][k] = 0.0f;
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.0f;

    for (int bb = b0; bb < b1; ++bb) {
    const int li = p.nlist ? bb / p.Bper : 0, b = p.nlist ? bb % p.Bper : bb;              // (wave-uniform)
    const float* xb = p.nlist ? p.xl[li] : p.x;
    const float* x2b = p.nlist ? p.x2l[li] : p.x2;
    const float* dyb = p.nlist ? p.dyl[li] : p.dy;
    const BufRsrc rx = make_rsrc(inA ? xb + ((size_t)b * p.CA + ci0) * HWX : x2b + ((size_t)b * (p.Cin - p.CA) + (ci0 - p.CA)) * HWX,
                                 (uint32_t)((two ? 2 : 1) * HWX) * 4u);
    const BufRsrc ry = make_rsrc(dyb + ((size_t)b * p.Cout + cog * 8) * HW, (uint32_t)(nco * HW) * 4u);
    // window rows: win[c][r][k] = X[ci0 + c][y - 1 + r][x - 1 + k]; rows outside the image are zeros.  Every load is UNCONDITIONAL
    // (a row outside the image / past the chunk reads through a descriptor with zero records: the range check returns 0) -- a load
    // under a branch makes the compiler wait with vmcnt(0) at the join, and the prefetch below would hide nothing (mfma_conv.h).
    float win[2][3][3];
    auto load_row = [&](int yy, float (&dst)[2][3]) {                 // yy: row of the window tensor
        i32x4 r = rx.v;
        r.z = (yy >= 0 && yy < HX) ? r.z : 0;                         // wave-uniform scalar select
        const int so = (yy >= 0 && yy < HX) ? yy * WX * 4 : 0;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) dst[c][k] = llvm_raw_buffer_load_f32(r, (int)cx[k], c * HWX * 4 + so, 0);
    };
    auto load_g = [&](int yy, float (&dst)[8]) {
        i32x4 r = ry.v;
        r.z = yy < y1 ? r.z : 0;
        const int so = yy < y1 ? yy * W * 4 : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) dst[j] = llvm_raw_buffer_load_f32(r, (int)cy, j * HW * 4 + so, 0);     // channels beyond Cout: out of range = 0
    };
    // Grid row y reads window rows S*y - 1 .. S*y + 1.  Before iteration y the window holds rows (., S*y - 1, S*y) in slots 1, 2 for
    // S = 1 (one new row per iteration) resp. row S*y - 1 in slot 2 for S = 2 (two new rows per iteration).
    // The rows of iteration y+1 and its gradient row are requested BEFORE iteration y's arithmetic: un-prefetched, a wave pays a
    // full memory latency per row (measured: ~2 us per row on the full-resolution shapes, a 13 us floor on the small ones).
    float new_n[S][2][3], g_n[8];
    if (S == 1) {
        float r0[2][3], r1[2][3];
        load_row(y0 - 1, r0);
        load_row(y0, r1);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) { win[c][1][k] = r0[c][k]; win[c][2][k] = r1[c][k]; win[c][0][k] = 0.0f; }
        load_row(y0 + 1, new_n[0]);
    } else {
        float r0[2][3];
        load_row(S * y0 - 1, r0);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) { win[c][2][k] = r0[c][k]; win[c][0][k] = win[c][1][k] = 0.0f; }
#pragma unroll
        for (int q = 0; q < S; ++q) load_row(S * y0 + q, new_n[q]);
    }
    load_g(y0, g_n);
    for (int y = y0; y < y1; ++y) {
        float fresh[S][2][3], g[8];
#pragma unroll
        for (int q = 0; q < S; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) fresh[q][c][k] = new_n[q][c][k];
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = g_n[j];
        // next iteration's rows (past the chunk: not needed -> a row index outside the tensor)
#pragma unroll
        for (int q = 0; q < S; ++q) load_row(y + 1 < y1 ? (S == 1 ? y + 2 : S * (y + 1) + q) : HX, new_n[q]);
        load_g(y + 1, g_n);
        __builtin_amdgcn_sched_barrier(0);
        // slide the window by S rows
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (S == 1) { win[c][0][k] = win[c][1][k]; win[c][1][k] = win[c][2][k]; win[c][2][k] = fresh[0][c][k]; }
                else { win[c][0][k] = win[c][2][k]; win[c][1][k] = fresh[0][c][k]; win[c][2][k] = fresh[1][c][k]; }
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc[c][j][r * 3 + k] = fmaf(g[j], win[c][r][k], acc[c][j][r * 3 + k]);
            bsum[j] += g[j];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    }   // samples of this wave
    // reduce over the lanes four values at a time; lanes 15 / 31 / 47 / 63 publish values 4m + {0, 2, 1, 3}
    // value index: (c * 8 + j) * 9 + k for the weight gradients, 144 + j for the bias gradient
    auto value = [&](int i) -> float { return i < 144 ? acc[i / 72][(i % 72) / 9][i % 9] : bsum[i - 144]; };
    const int q = lane >> 4;
    const int sel = q == 1 ? 2 : q == 2 ? 1 : q;
    const bool publisher = (lane & 15) == 15;
#pragma unroll
    for (int m = 0; m < 38; ++m) {
        const float v = reduce4_rows(value(4 * m), value(4 * m + 1), value(4 * m + 2), value(4 * m + 3));
        const int i = 4 * m + sel;
        if (publisher) {
            if (i < 144) {
                const int c = i / 72, j = (i % 72) / 9, k = i % 9;
                if (j < nco && (c == 0 || two)) unsafeAtomicAdd(p.dw + ((size_t)(cog * 8 + j) * p.Cin + ci0 + c) * 9 + k, v);
            } else {
                const int j = i - 144;
                if (p.db && cp == 0 && j < nco) unsafeAtomicAdd(p.db + cog * 8 + j, v);
            }
        }
    }
}


// ---- 3x3x3 layers of CostRegNet (training path of --model casmvs / ucs) -------------------------------------------------------------
// Weight gradient of Conv3d (stride 1 / 2, pad 1) and ConvTranspose3d (stride 2, pad 1, output_padding 1) of
// /root/reference/modules/module.py:324-410 under loss.backward() (/root/reference/train.py:284):
//     dW[g][c][kd][ky][kx] = sum_{b,d,y,x} G[b][g][d][y][x] * Xw[b][c][S d + kd - 1][S y + ky - 1][S x + kx - 1]
// (G = the tensor on the D x H x W grid, Xw = the one read through the taps: layer input / output gradient, swapped for the
// transposed layers exactly like the 2-D kernel above).  On this image MIOpen answers these calls with
// naive_conv_ab_nonpacked_wrw_ncdhw (one thread per weight, double accumulation) and a batched-GEMM fallback: 1.35 s + 0.64 s of the
// 1.98 s training step of the 48/32/8 cascade at the 768x384 tile (profiles/r05_train_step_casmvs.txt).
// The 2-D scheme with a depth loop around it: a wave owns (a pair of window channels) x (8 grid channels) x (a 64-column strip) x
// (a range of rows) x (a range of grid planes) x ONE depth tap kd -- 144 sums in registers; for every grid plane d of its range it
// slides the 3x3 window down the rows of window plane S d + kd - 1 (planes outside the volume are skipped wave-uniformly).  Channel
// strides are those of the (B,C,D,H,W) volumes, so nothing is copied or transposed.
// Explicitly counted row pipeline (round 5, SMVS_WGRAD3_COUNTED; measured: no gain, off).  With plain loads the compiler closes every
// iteration of the row loop with s_waitcnt vmcnt(0) -- the prefetched row is a loop-carried value -- so a row of memory latency is covered
// by ONE row of arithmetic (72 v_pk_fma_f32).  Loads issued through inline assembly are invisible to
// the compiler's wait insertion; the kernel waits itself, counted: two register sets, each requested two iterations before its use.
__device__ __forceinline__ void wg_load(float& dst, const i32x4& r, uint32_t voff, int soff)
{
    asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=&v"(dst) : "v"(voff), "s"(r), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void wg_wait() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// the counted wait that releases a register set: the registers are operands of the wait itself, so that no read of them can be scheduled
// above it (an empty pinning statement next to an operand-less wait is not enough: the copies were hoisted over the wait)
template <int N>
__device__ __forceinline__ void wg_wait_set(float (&w)[2][3], float (&g)[8])
{
    asm volatile("s_waitcnt vmcnt(%14)"
                 : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[1][2]),
                   "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7])
                 : "n"(N) : "memory");
}
// The counted wait AND the hand-over of a set's 14 registers in ONE statement: the set is read nowhere else, so its registers live from the
// load statements to this one and nothing can be scheduled (or copied) in between that reads them before the wait.
template <int N>
__device__ __forceinline__ void wg_wait_take(const float (&w)[2][3], const float (&g)[8], float (&fw)[1][2][3], float (&fg)[8])
{
    asm volatile("s_waitcnt vmcnt(%28)\n\t"
                 "v_mov_b32 %0, %14\n\tv_mov_b32 %1, %15\n\tv_mov_b32 %2, %16\n\tv_mov_b32 %3, %17\n\tv_mov_b32 %4, %18\n\tv_mov_b32 %5, %19\n\t"
                 "v_mov_b32 %6, %20\n\tv_mov_b32 %7, %21\n\tv_mov_b32 %8, %22\n\tv_mov_b32 %9, %23\n\tv_mov_b32 %10, %24\n\tv_mov_b32 %11, %25\n\t"
                 "v_mov_b32 %12, %26\n\tv_mov_b32 %13, %27"
                 : "=&v"(fw[0][0][0]), "=&v"(fw[0][0][1]), "=&v"(fw[0][0][2]), "=&v"(fw[0][1][0]), "=&v"(fw[0][1][1]), "=&v"(fw[0][1][2]),
                   "=&v"(fg[0]), "=&v"(fg[1]), "=&v"(fg[2]), "=&v"(fg[3]), "=&v"(fg[4]), "=&v"(fg[5]), "=&v"(fg[6]), "=&v"(fg[7])
                 : "v"(w[0][0]), "v"(w[0][1]), "v"(w[0][2]), "v"(w[1][0]), "v"(w[1][1]), "v"(w[1][2]),
                   "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3]), "v"(g[4]), "v"(g[5]), "v"(g[6]), "v"(g[7]), "n"(N)
                 : "memory");
}
__device__ __forceinline__ void wg_wait_rows(float (&a)[2][3], float (&b)[2][3])
{
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[0][2]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[1][2]),
                   "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[0][2]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[1][2])
                 :: "memory");
}

struct Wgrad3Params {
    const float* x; const float* dy; float* dw;
    int B, Cin, Cout, D, H, W;          // D, H, W: the grid; the window tensor is (B, Cin, S*D, S*H, S*W)
    int ncp, ncog, nxs, nrc, rows;      // window-channel pairs, grid-channel groups of 8, column strips, row chunks, rows per chunk
    int ndc, dchunk;                    // chunks of grid planes, planes per chunk
    float* part;                        // null: sums go to dw with float atomics; else (groups, waves per group, 144) partial sums for conv3d_wgrad_fold
};

#ifndef SMVS_WGRAD3_ROTATE
#define SMVS_WGRAD3_ROTATE 1            // stride 1: four rotating row slots, loop unrolled by four (0: the sliding window, A/B)
#endif
#ifndef SMVS_WGRAD3_GC1
#define SMVS_WGRAD3_GC1 1               // single-channel grid tensors (the `prob` layer) on the one-grid-channel instance (0: the 8-channel group, A/B)
#endif
#ifndef SMVS_WGRAD3_XCD
#define SMVS_WGRAD3_XCD 1               // sharers of a row adjacent in the unit order + one contiguous run of workgroups per XCD (0: the first order, A/B)
#endif
#ifndef SMVS_WGRAD3_COUNTED
#define SMVS_WGRAD3_COUNTED 0           // 1: stride 1 on the explicitly counted two-set row pipeline below.  Built and measured in round 5 (tests green): 10.4 against 9.9 ms per 5 casmvs steps -- row latency is not what bounds the kernel (two-ahead through the compiler: 9.9 as well); off
#endif

// GC: grid channels a wave carries -- 8, or 1 for the single-channel `prob` layer (round 5: with 8 it ran seven padding channels, a quarter
// of the step's weight-gradient row iterations for one eighth of the useful work)
template <int S, int GC>
__global__ __launch_bounds__(256)
void conv3d_wgrad_kernel(const Wgrad3Params p)
{
    const int lane = threadIdx.x & 63;
    // Which waves read the same rows: a grid row (8 channels of one (b, d, y-range, x-strip)) is wanted by every window-channel pair and
    // depth tap -- ncp x 3 waves, 24 at stage 2's conv0 -- a window row by the three depth taps of neighbouring planes and every grid
    // group.  Round 5: with the row chunk fastest and workgroups dealt round-robin over the 8 XCDs those sharers sat on different XCDs and
    // each L2 fetched its own copy (2.2 GB of L2 misses per launch for 0.23 GB of tensors: the kernel ran at the fabric's rate).  Now depth
    // tap, channel pair and grid group run fastest and every XCD gets one contiguous run of the workgroup order (xcd_remap): the sharers
    // are neighbours on one XCD, in flight together.
    const uint32_t wg = SMVS_WGRAD3_XCD ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    long long unit = (long long)wg * 4 + (threadIdx.x >> 6);           // one wave = one unit
    const long long total = (long long)p.ncp * p.ncog * p.nxs * p.nrc * 3 * p.ndc * p.B;
    if (unit >= total) return;
#if SMVS_WGRAD3_XCD
    const int kd = __builtin_amdgcn_readfirstlane((int)(unit % 3)); unit /= 3;
    const int cp = __builtin_amdgcn_readfirstlane((int)(unit % p.ncp)); unit /= p.ncp;
    const int cog = __builtin_amdgcn_readfirstlane((int)(unit % p.ncog)); unit /= p.ncog;
    const int rc = __builtin_amdgcn_readfirstlane((int)(unit % p.nrc)); unit /= p.nrc;
    const int xs = __builtin_amdgcn_readfirstlane((int)(unit % p.nxs)); unit /= p.nxs;
    const int dc = __builtin_amdgcn_readfirstlane((int)(unit % p.ndc));
    const int b = __builtin_amdgcn_readfirstlane((int)(unit / p.ndc));
#else
    // (first form) row chunk fastest, then column strip, depth tap, plane chunk, channel pair, output group, batch
    const int rc = __builtin_amdgcn_readfirstlane((int)(unit % p.nrc)); unit /= p.nrc;
    const int xs = __builtin_amdgcn_readfirstlane((int)(unit % p.nxs)); unit /= p.nxs;
    const int kd = __builtin_amdgcn_readfirstlane((int)(unit % 3)); unit /= 3;
    const int dc = __builtin_amdgcn_readfirstlane((int)(unit % p.ndc)); unit /= p.ndc;
    const int cp = __builtin_amdgcn_readfirstlane((int)(unit % p.ncp)); unit /= p.ncp;
    const int cog = __builtin_amdgcn_readfirstlane((int)(unit % p.ncog));
    const int b = __builtin_amdgcn_readfirstlane((int)(unit / p.ncog));
#endif
    const int D = p.D, H = p.H, W = p.W, HW = H * W;
    const int DX = S * D, HX = S * H, WX = S * W, HWX = HX * WX;
    const size_t csx = (size_t)DX * HWX, csy = (size_t)D * HW;        // channel strides (elements)
    const int x = xs * 64 + lane;
    const int y0 = rc * p.rows, y1 = min(y0 + p.rows, H);
    const int dz0 = dc * p.dchunk, dz1 = min(dz0 + p.dchunk, D);
    const int ci0 = 2 * cp;
    const bool two = ci0 + 1 < p.Cin;
    uint32_t cx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int xx = S * x - 1 + k;
        cx[k] = (x < W && xx >= 0 && xx < WX) ? (uint32_t)xx * 4u : SMVS_OOB;
    }
    const uint32_t cy = x < W ? (uint32_t)x * 4u : SMVS_OOB;
    const int nco = min(8, p.Cout - cog * 8);
    const int ch1 = two ? (int)(csx * 4) : 0;                         // byte offset of the pair's second channel (odd counts: the first again, not published)
    int gch[GC];                                                       // byte offsets of the 8 grid channels (beyond Cout: the last one again, not published)
#pragma unroll
    for (int j = 0; j < GC; ++j) gch[j] = (int)((size_t)min(j, nco - 1) * csy * 4);

    float acc[2][GC][9];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < GC; ++j)
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[c][j][k] = 0.0f;

    for (int d = dz0; d < dz1; ++d) {
    const int zd = S * d + kd - 1;                                    // window plane of this grid plane and depth tap
    if (zd < 0 || zd >= DX) continue;
    const BufRsrc rx = make_rsrc(p.x + ((size_t)b * p.Cin + ci0) * csx + (size_t)zd * HWX, (uint32_t)(ch1 + HWX * 4));
    const BufRsrc ry = make_rsrc(p.dy + ((size_t)b * p.Cout + cog * 8) * csy + (size_t)d * HW, (uint32_t)(gch[GC - 1] + HW * 4));
    float win[2][3][3];
    auto load_row = [&](int yy, float (&dst)[2][3]) {                 // yy: row of the window plane; every load unconditional (see the 2-D kernel)
        i32x4 r = rx.v;
        r.z = (yy >= 0 && yy < HX) ? r.z : 0;
        const int so = (yy >= 0 && yy < HX) ? yy * WX * 4 : 0;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) dst[c][k] = llvm_raw_buffer_load_f32(r, (int)cx[k], c * ch1 + so, 0);
    };
    auto load_g = [&](int yy, float (&dst)[GC]) {
        i32x4 r = ry.v;
        r.z = yy < y1 ? r.z : 0;
        const int so = yy < y1 ? yy * W * 4 : 0;
#pragma unroll
        for (int j = 0; j < GC; ++j) dst[j] = llvm_raw_buffer_load_f32(r, (int)cy, gch[j] + so, 0);
    };
    // The rows of iteration y: window row y + 1 (S = 1) / rows 2y, 2y + 1 (S = 2) and grid row y.
    auto fma_row = [&](const float (&fresh)[S][2][3], const float (&g)[GC]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                if (S == 1) { win[c][0][k] = win[c][1][k]; win[c][1][k] = win[c][2][k]; win[c][2][k] = fresh[0][c][k]; }
                else { win[c][0][k] = win[c][2][k]; win[c][1][k] = fresh[0][c][k]; win[c][2][k] = fresh[1][c][k]; }
            }
#pragma unroll
        for (int j = 0; j < GC; ++j)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int k = 0; k < 3; ++k) acc[c][j][r * 3 + k] = fmaf(g[j], win[c][r][k], acc[c][j][r * 3 + k]);
    };
    if constexpr (!(S == 1 && GC == 8 && SMVS_WGRAD3_COUNTED)) {
        if (S == 1) {
            float r0[2][3], r1[2][3];
            load_row(y0 - 1, r0);
            load_row(y0, r1);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) { win[c][1][k] = r0[c][k]; win[c][2][k] = r1[c][k]; win[c][0][k] = 0.0f; }
        } else {
            float r0[2][3];
            load_row(S * y0 - 1, r0);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) { win[c][2][k] = r0[c][k]; win[c][0][k] = win[c][1][k] = 0.0f; }
        }
    }
    if constexpr (S == 1 && GC == 8 && SMVS_WGRAD3_COUNTED) {
        // (every load of this path goes through wg_load: one compiler-visible load whose value is first used inside the row loop would
        //  put the compiler's own vmcnt(0) INTO the loop)
        {
            float r0[2][3], r1[2][3];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int yy = y0 - 1 + q;
                i32x4 r = rx.v;
                r.z = (yy >= 0 && yy < HX) ? r.z : 0;
                const int so = (yy >= 0 && yy < HX) ? yy * WX * 4 : 0;
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int k = 0; k < 3; ++k) wg_load(q ? r1[c][k] : r0[c][k], r, cx[k], c * ch1 + so);
            }
            wg_wait_rows(r0, r1);
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) { win[c][1][k] = r0[c][k]; win[c][2][k] = r1[c][k]; win[c][0][k] = 0.0f; }
        }
        // Two sets of 6 + 8 registers; set (y - y0) & 1 holds iteration y's rows.  A step hands its set over to the arithmetic's registers
        // inside the counted wait (wg_wait_take) and requests the set again, for iteration y + 2, BEFORE its arithmetic: a request has two
        // rows of arithmetic to land in.  In flight at a wait: the other set's 14 loads.
        float sw[2][2][3], sg[2][8];
        auto issue = [&](int yi, float (&w)[2][3], float (&g)[8]) __attribute__((always_inline)) {
            i32x4 rw = rx.v, rg = ry.v;
            const bool vw = yi < y1 && yi + 1 < HX, vg = yi < y1;     // (yi + 1 >= 0 always: yi >= y0 >= 0)
            rw.z = vw ? rw.z : 0; rg.z = vg ? rg.z : 0;               // rows outside the plane / past the chunk: zero records = zeros
            const int sow = vw ? (yi + 1) * WX * 4 : 0, sog = vg ? yi * W * 4 : 0;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) wg_load(w[c][k], rw, cx[k], c * ch1 + sow);
#pragma unroll
            for (int j = 0; j < 8; ++j) wg_load(g[j], rg, cy, gch[j] + sog);
        };
        auto step = [&](int y, float (&w)[2][3], float (&g)[8]) __attribute__((always_inline)) {
            float fw[1][2][3], fg[8];
            wg_wait_take<14>(w, g, fw, fg);
            issue(y + 2, w, g);
            if (y < y1) fma_row(fw, fg);                              // (wave-uniform: an odd chunk's last half-step multiplies nothing)
        };
        issue(y0, sw[0], sg[0]);
        issue(y0 + 1, sw[1], sg[1]);
        for (int y = y0; y < y1; y += 2) {
            step(y, sw[0], sg[0]);
            step(y + 1, sw[1], sg[1]);
        }
        wg_wait<0>();                                                 // nothing of ours in flight when the next plane starts
    } else if constexpr (S == 1 && SMVS_WGRAD3_ROTATE) {
        // Rotating window (round 5): FOUR row slots per window channel -- a step reads slots t, t+1, t+2 (mod 4) as window rows 0..2 and
        // the row of the NEXT step lands in slot t+3 while it runs -- and two sets of grid values; the loop is unrolled by four so every
        // slot index is a constant: no window slide, no hand-over copies (the first form spent ~50 of its ~125 vector instructions per row
        // on those moves).
        float rw[2][4][3], gs[2][GC];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) { rw[c][0][k] = win[c][1][k]; rw[c][1][k] = win[c][2][k]; }      // rows y0 - 1, y0 (the prologue's)
        auto load_slot = [&](int yy, int slot) __attribute__((always_inline)) {      // window row yy (outside the plane / past the chunk: zeros) -> slot
            i32x4 r = rx.v;
            const bool ok = yy >= 0 && yy < HX;
            r.z = ok ? r.z : 0;
            const int so = ok ? yy * WX * 4 : 0;
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int k = 0; k < 3; ++k) rw[c][slot][k] = llvm_raw_buffer_load_f32(r, (int)cx[k], c * ch1 + so, 0);
        };
        load_slot(y0 + 1, 2);
        load_g(y0, gs[0]);
        auto four_steps = [&](const int y, auto guarded) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int yy = y + u;
                load_slot(yy + 1 < y1 ? yy + 2 : HX, (u + 3) & 3);
                load_g(yy + 1, gs[(u + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                if (!decltype(guarded)::value || yy < y1) {           // (guarded: the last, partial group of a chunk -- wave-uniform)
#pragma unroll
                    for (int j = 0; j < GC; ++j)
#pragma unroll
                        for (int c = 0; c < 2; ++c)
#pragma unroll
                            for (int r = 0; r < 3; ++r)
#pragma unroll
                                for (int k = 0; k < 3; ++k)
                                    acc[c][j][r * 3 + k] = fmaf(gs[u & 1][j], rw[c][(u + r) & 3][k], acc[c][j][r * 3 + k]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // (whole groups without the guards -- four_steps(y, std::false_type{}) -- want 270 registers: 62 spilled at two waves per SIMD)
        for (int y = y0; y < y1; y += 4) four_steps(y, std::true_type{});
    } else {
        float new_n[S][2][3], g_n[GC];
#pragma unroll
        for (int q = 0; q < S; ++q) load_row(S == 1 ? y0 + 1 : S * y0 + q, new_n[q]);
        load_g(y0, g_n);
        for (int y = y0; y < y1; ++y) {
            float fresh[S][2][3], g[GC];
#pragma unroll
            for (int q = 0; q < S; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int k = 0; k < 3; ++k) fresh[q][c][k] = new_n[q][c][k];
#pragma unroll
            for (int j = 0; j < GC; ++j) g[j] = g_n[j];
#pragma unroll
            for (int q = 0; q < S; ++q) load_row(y + 1 < y1 ? (S == 1 ? y + 2 : S * (y + 1) + q) : HX, new_n[q]);
            load_g(y + 1, g_n);
            __builtin_amdgcn_sched_barrier(0);
            fma_row(fresh, g);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    }   // grid planes of this wave
    // reduce over the lanes four values at a time; lanes 15 / 31 / 47 / 63 publish values 4m + {0, 2, 1, 3}; value index (c * 8 + j) * 9 + k
    auto value = [&](int i) -> float { return (i % 72) / 9 < GC ? acc[i / 72][(i % 72) / 9 < GC ? (i % 72) / 9 : 0][i % 9] : 0.0f; };
    const int q = lane >> 4;
    const int sel = q == 1 ? 2 : q == 2 ? 1 : q;
    const bool publisher = (lane & 15) == 15;
#pragma unroll
    for (int m = 0; m < 36; ++m) {
        if (((4 * m) % 72) / 9 >= GC && ((4 * m + 3) % 72) / 9 >= GC) continue;      // (GC = 1: only the groups that hold channel 0's sums)
        const float v = reduce4_rows(value(4 * m), value(4 * m + 1), value(4 * m + 2), value(4 * m + 3));
        const int i = 4 * m + sel;
        if (publisher) {
            const int c = i / 72, j = (i % 72) / 9, k = i % 9;
            if (p.part) {
                // group = (cog, cp, kd): the waves that sum into the same 144 weights; wave index inside it = (b, dc, xs, rc)
                const size_t grp = ((size_t)cog * p.ncp + cp) * 3 + kd;
                const size_t wv = (((size_t)b * p.ndc + dc) * p.nxs + xs) * p.nrc + rc;
                p.part[(grp * ((size_t)p.B * p.ndc * p.nxs * p.nrc) + wv) * 144 + i] = v;
            } else if (j < nco && (c == 0 || two)) {
                unsafeAtomicAdd(p.dw + ((size_t)(cog * 8 + j) * p.Cin + ci0 + c) * 27 + kd * 9 + k, v);
            }
        }
    }
}

// Second stage of the two-stage form: dw += the sum of a group's per-wave partial sums.  Round 5: with float atomics straight from the
// waves, the ~100-400 waves of a group hammer the same five cache lines of dw (144 floats) -- the serialised L2 atomics bounded the
// kernel (more, shorter waves made it SLOWER: 9.9 -> 14.1 ms per 5 steps at 4x the waves), and the sums depended on the arrival order.
// One block per group, thread = one of the 144 weights, partial sums read coalesced and added in wave order: deterministic.
__global__ __launch_bounds__(192)
void conv3d_wgrad_fold_kernel(const Wgrad3Params p)
{
    const int grp = blockIdx.x, i = threadIdx.x;
    if (i >= 144) return;
    const int kd = grp % 3, cp = (grp / 3) % p.ncp, cog = grp / (3 * p.ncp);
    const size_t nper = (size_t)p.B * p.ndc * p.nxs * p.nrc;
    const float* q = p.part + (size_t)grp * nper * 144 + i;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    size_t w = 0;
    for (; w + 4 <= nper; w += 4) { s0 += q[w * 144]; s1 += q[(w + 1) * 144]; s2 += q[(w + 2) * 144]; s3 += q[(w + 3) * 144]; }
    for (; w < nper; ++w) s0 += q[w * 144];
    const int c = i / 72, j = (i % 72) / 9, k = i % 9;
    const int co = cog * 8 + j, ci = 2 * cp + c;
    if (co < p.Cout && ci < p.Cin) p.dw[((size_t)co * p.Cin + ci) * 27 + kd * 9 + k] += (s0 + s1) + (s2 + s3);
}

}  // namespace smvs

// list: nlist > 0 tensors of `B` samples each (x / x2 / dy ignored), else one tensor of B samples
static int wgrad_launch(const float* x, const float* dy, float* dw, float* db, int B, int Cin, int Cout, int H, int W, int stride, void* stream,
                        const float* x2 = nullptr, int CA = 0, int nlist = 0, const float* const* xl = nullptr, const float* const* x2l = nullptr,
                        const float* const* dyl = nullptr)
{
    using namespace smvs;
    if (!dw || (!nlist && (!x || !dy)) || (nlist && (!xl || !dyl))) return fail(SMVS_ERR_ARG, "null pointer argument");
    const bool second = nlist ? x2l != nullptr : x2 != nullptr;
    if (second && (CA < 2 || CA >= Cin || (CA & 1))) return fail(SMVS_ERR_ARG, "two-operand window: the first operand needs an even channel count in [2, Cin)");
    if (B < 1 || Cin < 1 || Cout < 1 || H < 1 || W < 1 || nlist < 0 || nlist > WGRAD_LIST_MAX) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if (stride != 1 && stride != 2) return fail(SMVS_ERR_ARG, "stride must be 1 or 2");
    if ((long long)2 * stride * stride * H * W * 4 >= (1ll << 31) || (long long)8 * H * W * 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "plane too large");
    WgradParams p{};
    p.x = x; p.dy = dy; p.dw = dw; p.db = db; p.Cin = Cin; p.Cout = Cout; p.H = H; p.W = W;
    p.x2 = x2; p.CA = second ? CA : Cin;
    p.nlist = nlist; p.Bper = B; p.B = nlist ? nlist * B : B;
    for (int i = 0; i < nlist; ++i) {
        if (!xl[i] || !dyl[i] || (second && !x2l[i])) return fail(SMVS_ERR_ARG, "null pointer in the tensor list");
        p.xl[i] = xl[i]; p.x2l[i] = second ? x2l[i] : nullptr; p.dyl[i] = dyl[i];
    }
    p.ncp = (Cin + 1) / 2; p.ncog = (Cout + 7) / 8; p.nxs = (W + 63) / 64;
    // Work per wave: whole samples while that leaves >= ~2048 waves (the window prologue and the lane reduction are paid once per wave),
    // then single samples cut into row chunks: long enough to amortise the reduction (~3 rows of arithmetic), short enough for ~2048 waves.
    const long long base = (long long)p.ncp * p.ncog * p.nxs;
    int bchunk = p.B, rows = H;
    while (bchunk > 1 && base * ((p.B + bchunk - 1) / bchunk) < 2048) bchunk = (bchunk + 1) / 2;
    const int nbc = (p.B + bchunk - 1) / bchunk;
    while (bchunk == 1 && rows > 16 && base * nbc * ((H + rows - 1) / rows) < 2048) rows = (rows + 1) / 2;
    p.bchunk = bchunk; p.nbc = nbc;
    p.rows = rows; p.nrc = (H + rows - 1) / rows;
    const long long units = base * p.nrc * p.nbc;
    if (units >= (1ll << 31)) return fail(SMVS_ERR_ARG, "too many work units");
    if (stride == 1) hipLaunchKernelGGL(conv3x3_wgrad_kernel<1>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    else             hipLaunchKernelGGL(conv3x3_wgrad_kernel<2>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "conv3x3_wgrad launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

extern "C" SMVS_EXPORT int smvs_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* db,
                                              int B, int Cin, int Cout, int H, int W, void* stream)
{
    return wgrad_launch(x, dy, dw, db, B, Cin, Cout, H, W, 1, stream);
}

extern "C" SMVS_EXPORT int smvs_conv3x3_wgrad_strided(const float* window, const float* grid, float* dw, float* dgrid_sum,
                                                      int B, int Cwin, int Cgrid, int H, int W, int stride, void* stream)
{
    return wgrad_launch(window, grid, dw, dgrid_sum, B, Cwin, Cgrid, H, W, stride, stream);
}

extern "C" SMVS_EXPORT int smvs_conv3x3_wgrad_cat(const float* xA, int CA, const float* xB, int CB, const float* dy, float* dw, float* db,
                                                  int B, int Cout, int H, int W, void* stream)
{
    if (CB > 0 && !xB) return smvs::fail(SMVS_ERR_ARG, "null pointer argument");
    return wgrad_launch(xA, dy, dw, db, B, CA + CB, Cout, H, W, 1, stream, CB > 0 ? xB : nullptr, CA);
}

// The same sums over a LIST of n tensors of Bper samples each (host arrays of device pointers; win2 null or n pointers): the planes of
// a training step.  dw / dgrid_sum are ACCUMULATED into (zeroed by the caller), so lists longer than one launch's table just continue.
extern "C" SMVS_EXPORT int smvs_conv3x3_wgrad_list(const float* const* win, const float* const* win2, const float* const* grid, int n,
                                                   float* dw, float* dgrid_sum, int Bper, int CA, int CB, int Cgrid, int H, int W, int stride,
                                                   void* stream)
{
    if (!win || !grid || n < 1) return smvs::fail(SMVS_ERR_ARG, "empty tensor list");
    for (int i = 0; i < n; i += smvs::WGRAD_LIST_MAX) {
        const int m = n - i < smvs::WGRAD_LIST_MAX ? n - i : smvs::WGRAD_LIST_MAX;
        const int rc = wgrad_launch(nullptr, nullptr, dw, dgrid_sum, Bper, CA + CB, Cgrid, H, W, stride, stream, nullptr, CA, m, win + i,
                                    (win2 && CB > 0) ? win2 + i : nullptr, grid + i);
        if (rc) return rc;
    }
    return SMVS_OK;
}

// Weight gradient of a 3x3x3 / pad 1 layer of the 3-D regulariser, ACCUMULATED into dw (zeroed by the caller):
//   window (B, Cwin, S*D, S*H, S*W), grid (B, Cgrid, D, H, W), dw (Cgrid, Cwin, 3, 3, 3), stride S in {1, 2}
// nn.Conv3d (stride S): window = layer input, grid = output gradient; nn.ConvTranspose3d (stride 2, output_padding 1): window = output
// gradient, grid = layer input (its weight is (Cin_layer, Cout_layer, 3,3,3) = the same index formula).
// Replaces MIOpen's weight-gradient kernels under /root/reference/modules/module.py:324-410, 546-577.
static void wgrad3_split(smvs::Wgrad3Params& p, bool two_stage)
{
    using namespace smvs;
    // Work per wave.  The chip holds 2048 of these waves at a time (218 VGPRs: two per SIMD), a launch runs in ROUNDS of that many and a
    // round lasts as long as one wave: pick the split into chunks of grid planes x row chunks that minimises
    //     rounds x (rows a wave walks + ~6 rows of prologue per plane + ~12 rows of lane reduction per wave)
    // Two-stage form only; with atomics straight from the waves (no workspace) the first form stays: halve until >= 2048 waves, because
    // every further wave of a group lengthens the chain of serialised atomics on the group's five cache lines.
    const int D = p.D, H = p.H;
    const long long base = (long long)p.ncp * p.ncog * p.nxs * 3 * p.B;
    int dchunk = D, rows = H;
    if (two_stage) {
        const long long slots = tune_int("SMVS_WGRAD3_SLOTS", 2048);
        long long best = -1;
        for (int dcs = 1; dcs <= D; ++dcs) {
            const int nd = (D + dcs - 1) / dcs;
            if ((D + nd - 1) / nd != dcs) continue;                        // (equivalent splits: keep the balanced one)
            for (int nr = 1; nr <= (dcs == 1 ? (H + 15) / 16 : 1); ++nr) { // row chunks only below one plane per wave
                const int rw = (H + nr - 1) / nr;
                if ((H + rw - 1) / rw != nr) continue;
                const long long units = base * nd * nr;
                const long long cost = ((units + slots - 1) / slots) * ((long long)dcs * (rw + 6) + 12);
                if (best < 0 || cost < best) { best = cost; dchunk = dcs; rows = rw; }
            }
        }
    } else {
        while (dchunk > 1 && base * ((D + dchunk - 1) / dchunk) < 2048) dchunk = (dchunk + 1) / 2;
        const int nd = (D + dchunk - 1) / dchunk;
        while (dchunk == 1 && rows > 16 && base * nd * ((H + rows - 1) / rows) < 2048) rows = (rows + 1) / 2;
    }
    p.dchunk = dchunk; p.ndc = (D + dchunk - 1) / dchunk; p.rows = rows; p.nrc = (H + rows - 1) / rows;
}

// floats of workspace the two-stage form of smvs_conv3d_wgrad wants for this shape (0: unsupported arguments)
extern "C" SMVS_EXPORT size_t smvs_conv3d_wgrad_workspace_floats(int B, int Cwin, int Cgrid, int D, int H, int W)
{
    using namespace smvs;
    if (B < 1 || Cwin < 1 || Cgrid < 1 || D < 1 || H < 1 || W < 1) return 0;
    Wgrad3Params p{};
    p.B = B; p.Cin = Cwin; p.Cout = Cgrid; p.D = D; p.H = H; p.W = W;
    p.ncp = (Cwin + 1) / 2; p.ncog = (Cgrid + 7) / 8; p.nxs = (W + 63) / 64;
    wgrad3_split(p, true);
    return (size_t)p.ncp * p.ncog * 3 * ((size_t)B * p.ndc * p.nxs * p.nrc) * 144;
}

extern "C" SMVS_EXPORT int smvs_conv3d_wgrad(const float* window, const float* grid, float* dw, float* workspace, size_t workspace_floats,
                                             int B, int Cwin, int Cgrid, int D, int H, int W, int stride, void* stream)
{
    using namespace smvs;
    if (!window || !grid || !dw) return fail(SMVS_ERR_ARG, "null pointer argument");
    if (B < 1 || Cwin < 1 || Cgrid < 1 || D < 1 || H < 1 || W < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    if (stride != 1 && stride != 2) return fail(SMVS_ERR_ARG, "stride must be 1 or 2");
    const long long S = stride, vol_g = (long long)D * H * W, vol_w = vol_g * S * S * S, plane_w = (long long)H * W * S * S;
    if ((vol_w + plane_w) * 4 >= (1ll << 31) || (7 * vol_g + (long long)H * W) * 4 >= (1ll << 31))
        return fail(SMVS_ERR_ARG, "volume too large for the 32-bit channel offsets of conv3d_wgrad (window %lld, grid %lld elements per channel)", vol_w, vol_g);
    Wgrad3Params p{};
    p.x = window; p.dy = grid; p.dw = dw; p.B = B; p.Cin = Cwin; p.Cout = Cgrid; p.D = D; p.H = H; p.W = W;
    p.ncp = (Cwin + 1) / 2; p.ncog = (Cgrid + 7) / 8; p.nxs = (W + 63) / 64;
    const size_t need = workspace ? smvs_conv3d_wgrad_workspace_floats(B, Cwin, Cgrid, D, H, W) : 0;
    if (workspace && workspace_floats < need) return fail(SMVS_ERR_ARG, "workspace too small: %zu < %zu floats", workspace_floats, need);
    wgrad3_split(p, workspace != nullptr);
    p.part = workspace;
    const long long base = (long long)p.ncp * p.ncog * p.nxs * 3 * B;
    const int ndc = p.ndc;
    const long long units = base * ndc * p.nrc;
    if ((units + 3) / 4 >= (1ll << 31)) return fail(SMVS_ERR_ARG, "too many work units");
    const dim3 grd((unsigned)((units + 3) / 4));
    if (Cgrid == 1 && SMVS_WGRAD3_GC1) {
        if (stride == 1) hipLaunchKernelGGL((conv3d_wgrad_kernel<1, 1>), grd, dim3(256), 0, (hipStream_t)stream, p);
        else             hipLaunchKernelGGL((conv3d_wgrad_kernel<2, 1>), grd, dim3(256), 0, (hipStream_t)stream, p);
    } else {
        if (stride == 1) hipLaunchKernelGGL((conv3d_wgrad_kernel<1, 8>), grd, dim3(256), 0, (hipStream_t)stream, p);
        else             hipLaunchKernelGGL((conv3d_wgrad_kernel<2, 8>), grd, dim3(256), 0, (hipStream_t)stream, p);
    }
    if (workspace) hipLaunchKernelGGL(conv3d_wgrad_fold_kernel, dim3((unsigned)(p.ncp * p.ncog * 3)), dim3(192), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "conv3d_wgrad launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}
