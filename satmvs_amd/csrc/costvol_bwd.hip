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
// per voxel-channel (1 for the reference gradient + 4 per source tap) = 20 ms at the 3-view 768x384x64 shape.  Three
// reductions BEFORE the atomics, none of which changes what is summed (only the order, like any atomic scatter):
//   * a lane owns its pixel for a chunk of DCH planes: the reference gradient is summed over the chunk in a
//     register -- one atomic per DCH planes;
//   * neighbouring lanes of a wave (a PX x 64/PX patch of ref pixels) hit neighbouring cells: where the lane to the
//     east has its north-west cell ON this lane's north-east cell (decided once per tap, not per channel), this lane
//     hands its two east contributions over the DPP network (wave_shr:1) and the neighbour folds them into its west
//     ones: 2 atomics per tap instead of 4 inside a run of such lanes.  (With PX < 64 a lane likewise hands its two
//     south contributions to the lane one patch row below, ds_bpermute: one atomic per interior tap -- measured, no
//     gain over the single-row patch, which ships);
//   * consecutive planes of a lane whose taps fall into the SAME cell (small parallax per plane: cascade stages 2-3,
//     coarse stage 1) are summed in registers and flushed when the cell changes.
// Taps that touch the image border (some corner outside) keep the plain per-corner path; bits of the loss are
// unaffected, gradients differ from round 2 by float32 summation order only (tests: reference-captured gradients,
// tests/golden/grad.npz and train_step.npz, and autograd of the torch composite).
#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

constexpr int MAX_SRC = 7;
constexpr int TILE_X = 64, TILE_Y = 4;

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

#ifndef SMVS_BWD_PX
#define SMVS_BWD_PX 64                 // patch width of a wave: 64 = one image row (east hand-over only).  32 (32 x 2) and 16 (16 x 4) add the
                                       // south hand-over: measured 9.35 / 11.2 ms against 9.33 ms at the metric shape (fewer atomics, but
                                       // shorter row segments per gather and store), so the single row ships
#endif
constexpr uint32_t TAP_DROPPED = 0x80000000u;      // = SMVS_OOB: a load through it returns 0
constexpr uint32_t TAP_PARTIAL = 0xC0000000u;      // | (y0+1) << 15 | (x0+1): some corner lies outside the image

__device__ __forceinline__ float dpp_from_prev_lane(float v)
{
    // wave_shr:1 -- lane i receives lane i-1's value, lane 0 keeps `old` (0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ uint32_t dpp_from_prev_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)TAP_DROPPED, (int)v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ uint32_t dpp_from_next_lane(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }   // wave_shl:1

__device__ __forceinline__ float lane_from(float v, int src_lane) { return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane << 2, __builtin_bit_cast(int, v))); }
__device__ __forceinline__ uint32_t lane_from(uint32_t v, int src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v); }

// PX = patch width of a wave (64: one image row; 32: 32 x 2; 16: 16 x 4); a workgroup of 4 waves covers 64 x 4 pixels
template <int GEO, int NSRC, int DCH, int PX>
__global__ __launch_bounds__(TILE_X * TILE_Y, 3)
void costvol_bwd_kernel(const CostVolBwdParams p)
{
    static_assert(DCH * NSRC <= 32, "tap flag masks");
    constexpr int PY = 64 / PX;
    uint32_t L = xcd_remap(blockIdx.x, gridDim.x);
    const int xtile = L % p.xt; L /= p.xt;
    const int dchunk = L % p.dct; L /= p.dct;
    const int ytile = L % p.yt;
    const int b = L / p.yt;
    const int lane = threadIdx.x;                           // blockDim = (64, 4): threadIdx.y = wave
    const int wv_ = threadIdx.y;
    const int lx = lane % PX, ly = lane / PX;                // position inside the wave's PX x PY patch
    const int x = xtile * TILE_X + (wv_ % (TILE_X / PX)) * PX + lx;
    const int y = ytile * TILE_Y + (wv_ / (TILE_X / PX)) * PY + ly;
    if (ytile * TILE_Y + (wv_ / (TILE_X / PX)) * PY >= p.H) return;      // whole wave below the image (wave-uniform)
    const bool active = x < p.W && y < p.H;

    const int H = p.H, W = p.W, C = p.C, D = p.D;
    const int HW = H * W;
    const int pix = min(y, H - 1) * W + min(x, W - 1);
    const int d0 = dchunk * DCH, d1 = min(d0 + DCH, D);
    const float half_wm1 = (float)((W - 1) * 0.5), half_hm1 = (float)((H - 1) * 0.5);
    const float fV = (float)p.V, rV = __fdiv_rn(1.0f, fV), two_over_v = 2.0f / fV;

    BufRsrc rs[NSRC];
#pragma unroll
    for (int s = 0; s < NSRC; ++s) rs[s] = make_rsrc(p.src[s] + (size_t)b * C * HW, (uint32_t)C * (uint32_t)HW * 4u);

    // ---- A: taps of the chunk's planes (float64 chain), once for all channels ------------------------------------
    uint32_t tb[DCH][NSRC];                                 // full tap: byte offset of its north-west cell | partial | dropped
    float tw[DCH][NSRC][4];                                 // nw, ne, sw, se
    uint32_t take = 0, give = 0;                            // bit d*NSRC+s: fold the west lane's east pair in / hand mine to the east lane
    uint32_t take_n = 0, give_s = 0, north_east = 0;        // same towards south; north_east: the lane above still owned its east pair
    uint32_t any_partial = 0, any_take = 0, any_take_n = 0; // wave-uniform: some lane of the wave has such a tap
    {
        const cgeo_t geo_b = as_cgeo((GEO == 0) ? p.geo + (size_t)b * p.V * RPC_LEN : p.geo + (size_t)b * (p.V - 1) * 16);
        RpcInv ref_n, src_n[NSRC];
        if (GEO == 0) {
            ref_n = rpc_inv_image(geo_b);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) src_n[s] = rpc_inv_ground(geo_b + (size_t)(s + 1) * RPC_LEN);
        }
        const double fx = (double)min(x, W - 1), fy = (double)min(y, H - 1);
#pragma unroll
        for (int k = 0; k < DCH; ++k) {
            const int d = min(d0 + k, D - 1);
            const float hf = p.depth_is_4d ? p.depth[((size_t)b * D + d) * HW + pix] : p.depth[(size_t)b * D + d];
            const double h = (double)hf;
            const cgeo_t geo_d = launder(geo_b);
            double lat = 0.0, lon = 0.0;
            if (GEO == 0) { rpc_photo2obj(geo_d, ref_n, fx, fy, h, lat, lon); pin(lat); pin(lon); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                Tap t;
                if (GEO == 0) {
                    double samp, line;
                    rpc_obj2photo(launder(geo_d) + (size_t)(s + 1) * RPC_LEN, src_n[s], lat, lon, h, samp, line);
                    t = tap_from_pixel((float)samp, (float)line, H, W, half_wm1, half_hm1);
                } else {
                    const cgeo_t P = geo_d + s * 16;
                    const double rx = fma(P[1], fy, P[0] * fx) + P[2];
                    const double ry = fma(P[5], fy, P[4] * fx) + P[6];
                    const double rz = fma(P[9], fy, P[8] * fx) + P[10];
                    const double X = fma(rx, h, P[3]), Y = fma(ry, h, P[7]), Z = fma(rz, h, P[11]);
                    t = tap_from_grid((float)((X / Z) / ((W - 1) * 0.5) - 1.0), (float)((Y / Z) / ((H - 1) * 0.5) - 1.0), H, W);
                }
                tw[k][s][0] = t.nw; tw[k][s][1] = t.ne; tw[k][s][2] = t.sw; tw[k][s][3] = t.se;
                const bool in = active && d0 + k < d1;
                const bool v0 = t.o_nw != SMVS_OOB, v1 = t.o_ne != SMVS_OOB, v2 = t.o_sw != SMVS_OOB, v3 = t.o_se != SMVS_OOB;
                uint32_t e = TAP_DROPPED;
                if (in && v0 && v1 && v2 && v3) e = t.o_nw;
                else if (in && (v0 || v1 || v2 || v3)) {
                    e = TAP_PARTIAL | ((uint32_t)(t.y0 + 1) << 15) | (uint32_t)(t.x0 + 1);   // -1 <= y0 <= H-1, -1 <= x0 <= W-1
                }
                tb[k][s] = e;
                const uint32_t bit = 1u << (k * NSRC + s);
                const bool full = (e & TAP_DROPPED) == 0;
                const uint32_t pe = dpp_from_prev_lane(e);
                const bool tk = full && lx > 0 && (pe & TAP_DROPPED) == 0 && pe + 4u == e;
                const uint32_t nt = dpp_from_next_lane(tk ? 1u : 0u);
                const bool gv = lx < PX - 1 && nt;
                if (tk) take |= bit;
                if (gv) give |= bit;
                if (PY > 1) {
                    const uint32_t ue = lane_from(e, lane - PX);                      // lanes of the first patch row read garbage: masked by ly > 0
                    const bool tn = full && ly > 0 && (ue & TAP_DROPPED) == 0 && ue + 4u * (uint32_t)W == e;
                    const uint32_t st = lane_from(tn ? 1u : 0u, lane + PX);
                    const uint32_t uk = lane_from(gv ? 0u : 1u, lane - PX);
                    if (tn) take_n |= bit;
                    if (ly < PY - 1 && st) give_s |= bit;
                    if (tn && uk) north_east |= bit;
                    if (__builtin_amdgcn_ballot_w64(tn) != 0) any_take_n |= bit;
                }
                if (__builtin_amdgcn_ballot_w64((e & TAP_PARTIAL) == TAP_PARTIAL) != 0) any_partial |= bit;
                if (__builtin_amdgcn_ballot_w64(tk) != 0) any_take |= bit;
                __builtin_amdgcn_sched_barrier(0);          // one view at a time: its 80 coefficients leave the SGPRs before the next view's arrive
            }
        }
    }
    any_partial = __builtin_amdgcn_readfirstlane(any_partial);
    any_take = __builtin_amdgcn_readfirstlane(any_take);
    any_take_n = __builtin_amdgcn_readfirstlane(any_take_n);

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

    for (int c = 0; c < C; ++c) {
        // Re-materialise the per-tap words every channel: otherwise every lane mask derived from them (full / partial /
        // take / give, ~100 of them) is hoisted out of the loop as an SGPR pair and spilled (measured: 640 SGPR spills,
        // 256 VGPRs, one wave per SIMD).
#pragma unroll
        for (int k = 0; k < DCH; ++k)
#pragma unroll
            for (int s = 0; s < NSRC; ++s) asm volatile("" : "+v"(tb[k][s]));
        asm volatile("" : "+v"(take), "+v"(give), "+v"(take_n), "+v"(give_s), "+v"(north_east));
        const float r = refp[(size_t)c * HW];
        const int choff = c * HW * 4;
        float gref = 0.0f;
        uint32_t rkey[NSRC];                                // run of planes whose tap sits in the same cell: key + 4 sums
        float racc[NSRC][4];
        bool live1[NSRC], live2[NSRC], live3[NSRC];         // the run still owns its north-east / south-west / south-east cell on some plane
#pragma unroll
        for (int s = 0; s < NSRC; ++s) { rkey[s] = TAP_DROPPED; racc[s][0] = racc[s][1] = racc[s][2] = racc[s][3] = 0.0f; live1[s] = live2[s] = live3[s] = false; }
        auto flush = [&](int s) {
            if ((rkey[s] & TAP_DROPPED) == 0) {
                float* q = p.grad_src[s] + ((size_t)b * C + c) * HW + (rkey[s] >> 2);
                unsafeAtomicAdd(q, racc[s][0]);
                if (live1[s]) unsafeAtomicAdd(q + 1, racc[s][1]);
                if (live2[s]) unsafeAtomicAdd(q + W, racc[s][2]);
                if (live3[s]) unsafeAtomicAdd(q + W + 1, racc[s][3]);
            }
        };
#pragma unroll
        for (int k = 0; k < DCH; ++k) {
            const float g = (active && d0 + k < d1) ? gp[((size_t)c * D + k) * HW] * two_over_v : 0.0f;
            float wv[NSRC];
            float sum = r;
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const uint32_t e = tb[k][s], bit = 1u << (k * NSRC + s);
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
                float t = a0 * tw[k][s][0];
                t = fmaf(a1, tw[k][s][1], t);
                t = fmaf(a2, tw[k][s][2], t);
                t = fmaf(a3, tw[k][s][3], t);
                wv[s] = t;
                sum = sum + t;
            }
            const float m = div_by_views(sum, fV, rV);
            gref = fmaf(g, r - m, gref);
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const uint32_t e = tb[k][s], bit = 1u << (k * NSRC + s);
                const float gw = g * (wv[s] - m);
                float c0 = gw * tw[k][s][0], c1 = gw * tw[k][s][1], c2 = gw * tw[k][s][2], c3 = gw * tw[k][s][3];
                // contributions funnel EAST, then SOUTH (cells: east lane's NW/SW = my NE/SE; south lane's NW/NE = my SW/SE)
                const bool ge = (give & bit) != 0, gs = PY > 1 && (give_s & bit) != 0;
                if (any_take & bit) {                       // wave-uniform: somewhere in the wave an east pair moves one lane on
                    const float p1 = dpp_from_prev_lane(c1), p3 = dpp_from_prev_lane(c3);
                    if (take & bit) { c0 += p1; c2 += p3; }
                }
                if (ge) { c1 = 0.0f; c3 = 0.0f; }
                if (PY > 1 && (any_take_n & bit)) {         // wave-uniform: somewhere a south pair moves one patch row down
                    const float n2 = lane_from(c2, lane - PX), n3 = lane_from(c3, lane - PX);
                    if (take_n & bit) { c0 += n2; c1 += n3; }
                }
                if (gs) { c2 = 0.0f; c3 = 0.0f; }
                if ((e & TAP_DROPPED) == 0) {               // full tap: extend the run or start a new one
                    if (e != rkey[s]) {
                        flush(s);
                        rkey[s] = e; racc[s][0] = racc[s][1] = racc[s][2] = racc[s][3] = 0.0f; live1[s] = live2[s] = live3[s] = false;
                    }
                    racc[s][0] += c0; racc[s][1] += c1; racc[s][2] += c2; racc[s][3] += c3;
                    live1[s] = live1[s] || !ge || (north_east & bit) != 0;
                    live2[s] = live2[s] || !gs;
                    live3[s] = live3[s] || (!ge && !gs);
                } else if ((e & TAP_PARTIAL) == TAP_PARTIAL) {
                    float* plane = p.grad_src[s] + ((size_t)b * C + c) * HW;
                    const float cc[4] = {c0, c1, c2, c3};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint32_t o = corner_off(e, q);
                        if (o != SMVS_OOB) unsafeAtomicAdd(plane + (o >> 2), cc[q]);
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NSRC; ++s) flush(s);
        if (active) unsafeAtomicAdd(grefp + (size_t)c * HW, gref);
    }
}

template <int GEO, int NSRC>
static hipError_t launch_bwd_n(CostVolBwdParams p, hipStream_t st)
{
    // planes per lane: the taps of a chunk live in registers (5 per tap)
    constexpr int DCH = NSRC <= 2 ? 8 : NSRC <= 4 ? 4 : 2;
    p.dch = DCH < p.D ? DCH : p.D;
    p.dct = (p.D + DCH - 1) / DCH;
    const long long nb = (long long)p.xt * p.yt * p.dct * p.B;
    if (nb >= (1ll << 31)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((costvol_bwd_kernel<GEO, NSRC, DCH, SMVS_BWD_PX>), dim3((unsigned)nb), dim3(TILE_X, TILE_Y), 0, st, p);
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
    p.B = B; p.V = n_src + 1; p.C = C; p.D = D; p.H = H; p.W = W; p.depth_is_4d = depth_is_4d;
    p.xt = (W + TILE_X - 1) / TILE_X; p.yt = (H + TILE_Y - 1) / TILE_Y;
    hipError_t e = geo_kind == 0 ? launch_bwd<0>(p, (hipStream_t)stream) : launch_bwd<1>(p, (hipStream_t)stream);
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "costvol_bwd launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}
