// smvs_device.h -- device-side building blocks of the RPC plane-sweep engine (gfx950 only).
//
//   rpc_*      float64 rational-cubic camera model, forward (ground -> image) and pre-fitted
//              inverse (image + height -> ground); reference semantics:
//              /root/reference/modules/warping.py:183-307.
//   Tap        the reference's bug-compatible bilinear sampler: grid normalised with the
//              align_corners=True formula, sampled by grid_sample(align_corners=False)
//              (/root/reference/modules/warping.py:350-359).  Rounding order follows
//              ATen's CPU kernel so the float32 result is bit-identical to oracle/oracle.c.
//   BufRsrc    raw buffer descriptor of one batch item's (C,H,W) feature block; out-of-image taps
//              are given an out-of-range offset and the hardware range check returns 0 for them
//              for free -- grid_sample's padding_mode='zeros' without a select per channel.
//
// Compile with -ffp-contract=off: every FMA in here is explicit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace smvs {

// ---- 170-vector layout (tools/RPCCore.py:8-28) -----------------------------------------------
enum : int {
    I_LINE_OFF = 0, I_SAMP_OFF = 1, I_LAT_OFF = 2, I_LON_OFF = 3, I_H_OFF = 4,
    I_LINE_SCALE = 5, I_SAMP_SCALE = 6, I_LAT_SCALE = 7, I_LON_SCALE = 8, I_H_SCALE = 9,
    I_LNUM = 10, I_LDEN = 30, I_SNUM = 50, I_SDEN = 70,
    I_LATNUM = 90, I_LATDEN = 110, I_LONNUM = 130, I_LONDEN = 150, RPC_LEN = 170
};

// Wave-uniform camera parameters are read through the constant address space so that every
// access is a scalar load (s_load_*) into SGPRs; a generic pointer would turn them into
// per-lane flat loads.
typedef const double __attribute__((address_space(4))) * cgeo_t;

__device__ __forceinline__ cgeo_t as_cgeo(const double* p) { return (cgeo_t)(uintptr_t)p; }

// Re-materialise a uniform pointer in SGPRs so loads through it cannot be hoisted out of the
// enclosing loop (see costvol.hip: hoisting 80*V coefficients spills ~570 SGPRs).
__device__ __forceinline__ cgeo_t launder(cgeo_t p)
{
    uintptr_t v = (uintptr_t)p;
    asm volatile("" : "+s"(v));
    return (cgeo_t)v;
}

// Four cubics sharing one monomial set: n0/d0 and n1/d1 are the two rational outputs.
// k0..k3 point at 20 wave-uniform coefficients each (scalar loads).  Monomial order is RPC00B
//   1,L,P,H,LP,LH,PH,LL,PP,HH,PLH,LLL,LPP,LHH,LLP,PPP,PHH,LLH,PPH,HHH   (warping.py:183-207)
// built by the reference's product chain; every term is acc = fma(monomial, coeff, acc), i.e. the
// SGPR coefficient is a MULTIPLICAND of v_fmac_f64 and the accumulator stays in VGPRs.  (A nested
// Horner form needs 16 fewer multiplies but puts the coefficients in the ADDEND slot, which on
// gfx950 costs two v_mov_b32 per FMA to copy them out of SGPRs -- measured 1318 vs 57 moves per
// kernel -- so the flat form is the cheaper one here.)
__device__ __forceinline__ void cubic4(cgeo_t k0, cgeo_t k1, cgeo_t k2, cgeo_t k3,
                                       double P, double L, double H,
                                       double& n0, double& d0, double& n1, double& d1)
{
    double a0 = k0[0], a1 = k1[0], a2 = k2[0], a3 = k3[0];
#define SMVS_ACC(i, t)                                                         \
    a0 = fma((t), k0[i], a0); a1 = fma((t), k1[i], a1);                        \
    a2 = fma((t), k2[i], a2); a3 = fma((t), k3[i], a3);
    const double LP = L * P, LH = L * H, PH = P * H, LL = L * L, PP = P * P, HH = H * H;
    SMVS_ACC(1, L)  SMVS_ACC(2, P)  SMVS_ACC(3, H)
    SMVS_ACC(4, LP) SMVS_ACC(5, LH) SMVS_ACC(6, PH)
    SMVS_ACC(7, LL) SMVS_ACC(8, PP) SMVS_ACC(9, HH)
    SMVS_ACC(10, P * LH) SMVS_ACC(11, L * LL) SMVS_ACC(12, L * PP) SMVS_ACC(13, L * HH)
    SMVS_ACC(14, L * LP) SMVS_ACC(15, P * PP) SMVS_ACC(16, P * HH) SMVS_ACC(17, L * LH)
    SMVS_ACC(18, P * PH) SMVS_ACC(19, H * HH)
#undef SMVS_ACC
    n0 = a0; d0 = a1; n1 = a2; d1 = a3;
}

// The same four cubics at TWO points at once (two height planes of one pixel): every coefficient is
// loaded into SGPRs once and feeds eight FMAs instead of four, which halves the scalar-load batches
// (and their s_waitcnt stalls) per voxel and doubles the independent FMA chains.
struct Quad { double n0, d0, n1, d1; };

__device__ __forceinline__ void cubic4x2(cgeo_t k0, cgeo_t k1, cgeo_t k2, cgeo_t k3,
                                         double Pa, double La, double Ha, double Pb, double Lb, double Hb,
                                         Quad& qa, Quad& qb)
{
    double a0 = k0[0], a1 = k1[0], a2 = k2[0], a3 = k3[0];
    double b0 = a0, b1 = a1, b2 = a2, b3 = a3;
#define SMVS_ACC2(i, ta, tb)                                                   \
    { const double c0 = k0[i], c1 = k1[i], c2 = k2[i], c3 = k3[i];             \
      const double ua = (ta), ub = (tb);                                       \
      a0 = fma(ua, c0, a0); a1 = fma(ua, c1, a1); a2 = fma(ua, c2, a2); a3 = fma(ua, c3, a3); \
      b0 = fma(ub, c0, b0); b1 = fma(ub, c1, b1); b2 = fma(ub, c2, b2); b3 = fma(ub, c3, b3); }
    const double LPa = La * Pa, LHa = La * Ha, PHa = Pa * Ha, LLa = La * La, PPa = Pa * Pa, HHa = Ha * Ha;
    const double LPb = Lb * Pb, LHb = Lb * Hb, PHb = Pb * Hb, LLb = Lb * Lb, PPb = Pb * Pb, HHb = Hb * Hb;
    SMVS_ACC2(1, La, Lb)   SMVS_ACC2(2, Pa, Pb)   SMVS_ACC2(3, Ha, Hb)
    SMVS_ACC2(4, LPa, LPb) SMVS_ACC2(5, LHa, LHb) SMVS_ACC2(6, PHa, PHb)
    SMVS_ACC2(7, LLa, LLb) SMVS_ACC2(8, PPa, PPb) SMVS_ACC2(9, HHa, HHb)
    SMVS_ACC2(10, Pa * LHa, Pb * LHb) SMVS_ACC2(11, La * LLa, Lb * LLb) SMVS_ACC2(12, La * PPa, Lb * PPb)
    SMVS_ACC2(13, La * HHa, Lb * HHb) SMVS_ACC2(14, La * LPa, Lb * LPb) SMVS_ACC2(15, Pa * PPa, Pb * PPb)
    SMVS_ACC2(16, Pa * HHa, Pb * HHb) SMVS_ACC2(17, La * LHa, Lb * LHb) SMVS_ACC2(18, Pa * PHa, Pb * PHb)
    SMVS_ACC2(19, Ha * HHa, Hb * HHb)
#undef SMVS_ACC2
    qa.n0 = a0; qa.d0 = a1; qa.n1 = a2; qa.d1 = a3;
    qb.n0 = b0; qb.d0 = b1; qb.n1 = b2; qb.d1 = b3;
}

// n / d in float64 without the IEEE special-case scaffolding (v_div_scale / v_div_fmas /
// v_div_fixup): hardware reciprocal seed, two Newton steps, one residual correction.  |d| is a
// cubic with leading coefficient 1 on normalised arguments, i.e. ~1: no overflow/denormal cases to
// patch.  Error <= 1 ulp (typically correctly rounded); the stated geodesy tolerance is 1e-12 deg.
__device__ __forceinline__ double fast_div(double n, double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return fma(fma(-d, q, n), r, q);
}

// Two quotients n0/d0, n1/d1 with ONE reciprocal: r = 1/(d0*d1) (hardware seed + two Newton steps), q0 = (n0*d1)*r,
// q1 = (n1*d0)*r.  9 instructions + one v_rcp_f64 (a 16-clock transcendental) instead of 14 + two; no residual step,
// so the quotients carry <= 3 ulp (7e-16 relative: 2e-14 deg on a latitude, 5e-13 px on a pixel coordinate; tolerance
// 1e-12 deg / 1e-8 px).  Used by the staged cost-volume kernel, where the float64 chain is the largest VALU block; the
// flat projector (smvs_rpc_project) keeps fast_div.  The denominators are cubics with constant term 1 on normalised
// arguments (~1), so d0*d1 cannot overflow or vanish.
__device__ __forceinline__ void div_pair(double n0, double d0, double n1, double d1, double& q0, double& q1)
{
    const double dd = d0 * d1;
    double r = __builtin_amdgcn_rcp(dd);
    r = fma(fma(-dd, r, 1.0), r, r);
    r = fma(fma(-dd, r, 1.0), r, r);
    q0 = (n0 * d1) * r;
    q1 = (n1 * d0) * r;
}

// 1 / d for a scale of the camera model (|d| between 1e-4 and 1e5: no special cases): hardware seed (24 bits) + two Newton steps.
// Over 2^24 denominators the result equals the IEEE quotient bit for bit (tools/ubench_rcp.hip); where it should ever differ it
// does so by one unit in the last place, i.e. 1e-16 of a normalised coordinate.  Used wherever the staged cost-volume kernel and the
// fold kernel need the reciprocal scales (both call THIS function, so a wave computes the same bits with or without a workspace):
// five independent-ish instructions per scale instead of a ~35-instruction IEEE division behind a global load and an LDS round trip.
__device__ __forceinline__ double recip_scale(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}

// Loop-invariant part of one view's normalisation: only the three reciprocal scales a direction
// needs are kept live (6 SGPRs per view); offsets and forward scales are re-read with the
// coefficient block (scalar cache).  Reciprocals are taken once, in float64.
struct RpcInv { double a, b, h; };

__device__ __forceinline__ RpcInv rpc_inv_image(cgeo_t r)    // for photo -> object (ref view)
{
    RpcInv n;
    n.a = recip_scale(r[I_SAMP_SCALE]); n.b = recip_scale(r[I_LINE_SCALE]); n.h = recip_scale(r[I_H_SCALE]);
    return n;
}

__device__ __forceinline__ RpcInv rpc_inv_ground(cgeo_t r)   // for object -> photo (source views)
{
    RpcInv n;
    n.a = recip_scale(r[I_LAT_SCALE]); n.b = recip_scale(r[I_LON_SCALE]); n.h = recip_scale(r[I_H_SCALE]);
    return n;
}

// image (samp, line) + height -> ground (lat, lon); RPC_Photo2Obj, warping.py:255-307.
__device__ __forceinline__ void rpc_photo2obj(cgeo_t r, const RpcInv& n,
                                              double samp, double line, double hei,
                                              double& lat, double& lon)
{
    const double s = (samp - r[I_SAMP_OFF]) * n.a;
    const double l = (line - r[I_LINE_OFF]) * n.b;
    const double h = (hei - r[I_H_OFF]) * n.h;
    double an, ad, on, od;
    cubic4(r + I_LATNUM, r + I_LATDEN, r + I_LONNUM, r + I_LONDEN, s, l, h, an, ad, on, od);
    lat = fma(fast_div(an, ad), r[I_LAT_SCALE], r[I_LAT_OFF]);
    lon = fma(fast_div(on, od), r[I_LON_SCALE], r[I_LON_OFF]);
}

// ground (lat, lon, h) -> image (samp, line); RPC_Obj2Photo, warping.py:218-252.
__device__ __forceinline__ void rpc_obj2photo(cgeo_t r, const RpcInv& n,
                                              double lat, double lon, double hei,
                                              double& samp, double& line)
{
    const double p = (lat - r[I_LAT_OFF]) * n.a;
    const double l = (lon - r[I_LON_OFF]) * n.b;
    const double h = (hei - r[I_H_OFF]) * n.h;
    double sn, sd, ln, ld;
    cubic4(r + I_SNUM, r + I_SDEN, r + I_LNUM, r + I_LDEN, p, l, h, sn, sd, ln, ld);
    samp = fma(fast_div(sn, sd), r[I_SAMP_SCALE], r[I_SAMP_OFF]);
    line = fma(fast_div(ln, ld), r[I_LINE_SCALE], r[I_LINE_OFF]);
}

// Two-plane forms (same pixel, heights ha / hb): identical arithmetic per plane.
__device__ __forceinline__ void rpc_photo2obj_x2(cgeo_t r, const RpcInv& n, double samp, double line,
                                                 double ha, double hb, double& lat_a, double& lon_a,
                                                 double& lat_b, double& lon_b)
{
    const double s = (samp - r[I_SAMP_OFF]) * n.a;
    const double l = (line - r[I_LINE_OFF]) * n.b;
    const double za = (ha - r[I_H_OFF]) * n.h, zb = (hb - r[I_H_OFF]) * n.h;
    Quad qa, qb;
    cubic4x2(r + I_LATNUM, r + I_LATDEN, r + I_LONNUM, r + I_LONDEN, s, l, za, s, l, zb, qa, qb);
    lat_a = fma(fast_div(qa.n0, qa.d0), r[I_LAT_SCALE], r[I_LAT_OFF]);
    lon_a = fma(fast_div(qa.n1, qa.d1), r[I_LON_SCALE], r[I_LON_OFF]);
    lat_b = fma(fast_div(qb.n0, qb.d0), r[I_LAT_SCALE], r[I_LAT_OFF]);
    lon_b = fma(fast_div(qb.n1, qb.d1), r[I_LON_SCALE], r[I_LON_OFF]);
}

__device__ __forceinline__ void rpc_obj2photo_x2(cgeo_t r, const RpcInv& n,
                                                 double lat_a, double lon_a, double ha,
                                                 double lat_b, double lon_b, double hb,
                                                 double& samp_a, double& line_a, double& samp_b, double& line_b)
{
    const double pa = (lat_a - r[I_LAT_OFF]) * n.a, la = (lon_a - r[I_LON_OFF]) * n.b, za = (ha - r[I_H_OFF]) * n.h;
    const double pb = (lat_b - r[I_LAT_OFF]) * n.a, lb = (lon_b - r[I_LON_OFF]) * n.b, zb = (hb - r[I_H_OFF]) * n.h;
    Quad qa, qb;
    cubic4x2(r + I_SNUM, r + I_SDEN, r + I_LNUM, r + I_LDEN, pa, la, za, pb, lb, zb, qa, qb);
    samp_a = fma(fast_div(qa.n0, qa.d0), r[I_SAMP_SCALE], r[I_SAMP_OFF]);
    line_a = fma(fast_div(qa.n1, qa.d1), r[I_LINE_SCALE], r[I_LINE_OFF]);
    samp_b = fma(fast_div(qb.n0, qb.d0), r[I_SAMP_SCALE], r[I_SAMP_OFF]);
    line_b = fma(fast_div(qb.n1, qb.d1), r[I_LINE_SCALE], r[I_LINE_OFF]);
}

// ---- rational cubics in nested (Horner) form, coefficients in SGPRs -------------------------------------
//   poly = A + H*(B + H*(C + H*c19))
//   A = (c0 + P(c2 + P(c8 + P c15))) + L((c1 + P(c4 + P c12)) + L((c7 + P c14) + L c11))
//   B = (c3 + P(c6 + P c18)) + L((c5 + P c10) + L c17)        C = c9 + L c13 + P c16
// (monomial order 1,L,P,H,LP,LH,PH,LL,PP,HH,PLH,LLL,LPP,LHH,LLP,PPP,PHH,LLH,PPH,HHH, warping.py:183-207):
// 19 FMAs per cubic and no monomial products (the flat form costs 16 products + 76 FMAs per four cubics).
// A VALU instruction reads at most one SGPR operand, and the six innermost steps (c8 + P c15, ...) combine two
// coefficients: one of each pair is copied to a VGPR first (v_mov_b64), shared by the N points evaluated
// together.  N points = N planes of one pixel: every scalar-loaded coefficient then feeds N FMAs.
// Same polynomial as the reference's sum of 20 products, different summation order: the results differ by
// rounding only (~1e-16 relative; tolerance 1e-12 deg / 1e-8 px, tests/test_hip_parity.py).
__device__ __forceinline__ double to_vgpr(double s)
{
    double v;
    asm("v_mov_b64 %0, %1" : "=v"(v) : "s"(s));
    return v;
}

// A(P,L), B(P,L), C(P,L) of one cubic at N points
template <int N>
__device__ __forceinline__ void cubic_abc_xn(cgeo_t kp, const double* P, const double* L, double* A, double* B, double* C)
{
    // all 19 coefficients requested before the first is used: one wait per cubic (the compiler's own order of use put two)
    double k[19];
#pragma unroll
    for (int j = 0; j < 19; ++j) k[j] = kp[j];
    __builtin_amdgcn_sched_barrier(0);
    const double v4 = to_vgpr(k[4]), v5 = to_vgpr(k[5]), v6 = to_vgpr(k[6]);
    const double v7 = to_vgpr(k[7]), v8 = to_vgpr(k[8]), v9 = to_vgpr(k[9]);
#pragma unroll
    for (int u = 0; u < N; ++u) {
        double t = fma(L[u], k[11], fma(P[u], k[14], v7));
        t = fma(L[u], t, fma(P[u], fma(P[u], k[12], v4), k[1]));
        A[u] = fma(L[u], t, fma(P[u], fma(P[u], fma(P[u], k[15], v8), k[2]), k[0]));
        t = fma(L[u], k[17], fma(P[u], k[10], v5));
        B[u] = fma(L[u], t, fma(P[u], fma(P[u], k[18], v6), k[3]));
        C[u] = fma(P[u], k[16], fma(L[u], k[13], v9));
    }
}

// Pin a value at this point of the program: it must be computed before, and cannot be re-derived after.  Stops
// the optimiser from sinking a finished computation down to its first use (which keeps its operands alive
// all the way there).
__device__ __forceinline__ void pin(double& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }

// RPC_Photo2Obj (warping.py:255-307) of one ref pixel, split into its plane-invariant part (A, B, C of the
// four cubics in the normalised (samp, line): 64 FMAs per pixel) and the per-plane part (Horner in the
// normalised height: 12 FMAs + two quotients).  inv = {1/SAMP_SCALE, 1/LINE_SCALE, 1/HEIGHT_SCALE}.
struct P2OPix { double A[4], B[4], C[4]; };

__device__ __forceinline__ void p2o_pixel(cgeo_t r, const RpcInv& n, double samp, double line, P2OPix& o)
{
    const double P = (samp - r[I_SAMP_OFF]) * n.a;
    const double L = (line - r[I_LINE_OFF]) * n.b;
    // one cubic at a time: the pointer is re-materialised so that only 20 coefficients are in SGPRs at once
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        cubic_abc_xn<1>(launder(r) + I_LATNUM + 20 * i, &P, &L, &o.A[i], &o.B[i], &o.C[i]);
        pin(o.A[i]); pin(o.B[i]); pin(o.C[i]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void p2o_plane(cgeo_t r, const RpcInv& n, const P2OPix& o, double hei, double& lat, double& lon)
{
    const double z = (hei - r[I_H_OFF]) * n.h;
    double q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = fma(z, fma(z, fma(z, r[I_LATNUM + 20 * i + 19], o.C[i]), o.B[i]), o.A[i]);
    double qa, qo;
    div_pair(q[0], q[1], q[2], q[3], qa, qo);
    lat = fma(qa, r[I_LAT_SCALE], r[I_LAT_OFF]);
    lon = fma(qo, r[I_LON_SCALE], r[I_LON_OFF]);
}

// The same for N planes of the pixel: the nine wave-uniform constants (height offset, the four c19, output scales and
// offsets) are fetched once for all of them -- p2o_plane in a loop re-reads them per plane behind three dependent
// scalar-load waits each.
template <int N>
__device__ __forceinline__ void p2o_planes(cgeo_t r, const RpcInv& n, const P2OPix& o, const float* hei, double* lat, double* lon)
{
    const double h_off = r[I_H_OFF];
    const double c19[4] = {r[I_LATNUM + 19], r[I_LATDEN + 19], r[I_LONNUM + 19], r[I_LONDEN + 19]};
    const double lat_s = r[I_LAT_SCALE], lon_s = r[I_LON_SCALE];
    const double lat_o = to_vgpr(r[I_LAT_OFF]), lon_o = to_vgpr(r[I_LON_OFF]);      // one copy for the N planes
#pragma unroll
    for (int u = 0; u < N; ++u) {
        const double z = ((double)hei[u] - h_off) * n.h;
        double q[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = fma(z, fma(z, fma(z, c19[i], o.C[i]), o.B[i]), o.A[i]);
        double qa, qo;
        div_pair(q[0], q[1], q[2], q[3], qa, qo);
        lat[u] = fma(qa, lat_s, lat_o);
        lon[u] = fma(qo, lon_s, lon_o);
        pin(lat[u]); pin(lon[u]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// RPC_Obj2Photo (warping.py:218-252) at N ground points at once (N planes of one pixel), one cubic at a time.
// n = {1/LAT_SCALE, 1/LONG_SCALE, 1/HEIGHT_SCALE}.
template <int N>
__device__ __forceinline__ void o2p_xn(cgeo_t r, const RpcInv& n, const double* lat, const double* lon, const double* hei,
                                       double* samp, double* line)
{
    double P[N], L[N], H[N], q[4][N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
        P[u] = (lat[u] - r[I_LAT_OFF]) * n.a;
        L[u] = (lon[u] - r[I_LON_OFF]) * n.b;
        H[u] = (hei[u] - r[I_H_OFF]) * n.h;
    }
    constexpr int base[4] = {I_SNUM, I_SDEN, I_LNUM, I_LDEN};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        const cgeo_t k = launder(r) + base[i];
        double A[N], B[N], C[N];
        cubic_abc_xn<N>(k, P, L, A, B, C);
#pragma unroll
        for (int u = 0; u < N; ++u) {
            q[i][u] = fma(H[u], fma(H[u], fma(H[u], k[19], C[u]), B[u]), A[u]);
            pin(q[i][u]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    const cgeo_t rr = launder(r);
    const double so = to_vgpr(rr[I_SAMP_OFF]), lo = to_vgpr(rr[I_LINE_OFF]);     // one copy for the N points
#pragma unroll
    for (int u = 0; u < N; ++u) {
        double qs, ql;
        div_pair(q[0][u], q[1][u], q[2][u], q[3][u], qs, ql);
        samp[u] = fma(qs, rr[I_SAMP_SCALE], so);
        line[u] = fma(ql, rr[I_LINE_SCALE], lo);
    }
}

// ---- plane-constant heights: the source cubics collapse to bivariate ones (round 6) ----------------------
// Stage 1 of every cascade hands the warp PLANE-CONSTANT heights (networks/casred.py:138-149,
// modules/depth_range.py:23-42; modules/warping.py:329-332 accepts the (B,D) form).  With the normalised height H
// of a plane fixed, a source view's cubic sum_i c_i m_i(P,L,H) is a cubic in (P,L) alone with 10 coefficients
//   k0 (1)   = c0 + H c3 + H^2 c9 + H^3 c19     k1 (L)   = c1 + H c5 + H^2 c13     k2 (P)   = c2 + H c6 + H^2 c16
//   k3 (LP)  = c4 + H c10                       k4 (LL)  = c7 + H c17              k5 (PP)  = c8 + H c18
//   LLL: c11        LPP: c12        LLP: c14        PPP: c15                        (plane-invariant: read from the RPC itself)
// (monomial order 1,L,P,H,LP,LH,PH,LL,PP,HH,PLH,LLL,LPP,LHH,LLP,PPP,PHH,LLH,PPH,HHH, warping.py:183-207).  A tiny
// pre-kernel (rpc_plane_coef_kernel, costvol.hip) folds k0..k5 once per (batch item, plane, source, cubic) into a
// caller-owned workspace, and a wave whose 64 x DP heights all equal their planes' heights (checked, wave-uniform
// branch) evaluates 7 monomial products + 4 x (1 move + 9 FMAs) per source and plane instead of 6/4 moves + 76 FMAs +
// the height normalisation.  Same polynomial, re-associated: coordinates move by float64 rounding only (~1e-13 px, like
// the Horner form vs the reference's order).  Per-voxel heights (stages 2-3, jittered planes) fail the check and take
// the trivariate chain above.
// Workspace layout (doubles): [0, pc_heights_doubles(B D)) the planes' heights, (b, d) at b D + d; the views' reciprocal scales (below);
// then, from pc_header_doubles(B D, B) on, the folded
// coefficients as [b][source][cubic: SNUM, SDEN, LNUM, LDEN][d][6]: for one (source, cubic) the planes are contiguous,
// so the N planes of an evaluation pass are ONE run of 6 N doubles (three s_load_dwordx16 at N = 4) and a wave's DP
// planes are 4 n_src runs of 48 DP bytes.
// How the coefficients reach the FMAs (measured, profiles/r06_plane_coef_transport.txt): as SGPR multiplicands through the
// scalar cache.  The 64 planes' coefficients (25 KB at 3 views) do not fit the CU's 16 KB scalar cache next to the RPC
// vectors, so every workgroup finds its plane chunk's lines cold, and a cold line costs ~170 clocks whether it is
// requested back to back or by a dependent use -- the wave therefore pulls its 4 n_src runs in with blocks of
// back-to-back touches (scalar_touch) instead of stalling at 16-48 dependent loads.  The alternatives lose: a
// coalesced vector load of the block + two v_readlane_b32 per coefficient costs 0.65 ms against 0.58 for the trivariate
// chain (VALU-written SGPRs are slow to consume); broadcast LDS reads cost the LDS pipe what the FMAs save.
// Behind the heights, per batch item: the 3 reciprocal scales of every view (PC_SCALES doubles per batch item: view v at 3 v --
// 1/SAMP_SCALE, 1/LINE_SCALE, 1/HEIGHT_SCALE for the ref view, 1/LAT_SCALE, 1/LONG_SCALE, 1/HEIGHT_SCALE for a source view),
// IEEE divisions done once by the fold kernel: a wave that has the workspace reads them as scalars instead of dividing itself and
// passing the quotients through LDS (a 9 000-clock latency chain at the start of every wave: global load -> division -> LDS round trip).
enum : int { PC_PER_CUBIC = 6, PC_SCALES = 24 };
__host__ __device__ constexpr size_t pc_heights_doubles(size_t planes) { return (planes + 7) & ~(size_t)7; }
__host__ __device__ constexpr size_t pc_header_doubles(size_t planes, int B) { return pc_heights_doubles(planes) + (size_t)B * PC_SCALES; }
__host__ __device__ constexpr size_t pc_total_doubles(int B, int n_src, int D)
{
    // + 64: a group of DP planes cut short by the end of the sweep reads (and discards) up to DP - 1 records past plane D - 1,
    // and the touches of a run cover whole lines
    return pc_header_doubles((size_t)B * D, B) + (size_t)B * D * 4 * PC_PER_CUBIC * n_src + 64;
}
// doubles from the start of the coefficient area to (b, source s, cubic i, plane d)
__host__ __device__ constexpr size_t pc_offset(int b, int s, int i, int d, int n_src, int D)
{
    return ((((size_t)b * n_src + s) * 4 + i) * D + d) * PC_PER_CUBIC;
}

// Touch LINES 64-byte lines from p on (rounded down to a line) through the scalar cache, one dword each, and wait.  The
// loads carry IMMEDIATE offsets and are issued back to back, so the run costs about one scalar-cache latency; a loop
// that steps an SGPR offset between the loads costs ~200 clocks PER load -- the SALU write of the offset waits for the
// load in flight that reads it (measured: 13 000 clocks per wave for 50 lines, hits or misses alike,
// profiles/r06_plane_coef_transport.txt).  The destination register is dead; the wait inside the block is what makes
// that safe (the compiler may hand the register to anything after the statement -- and any later s_waitcnt lgkmcnt(0)
// would wait for the touches anyway: there is no such thing as a non-blocking scalar prefetch).
template <int LINES>
__device__ __forceinline__ void scalar_touch(cgeo_t p)
{
    static_assert(LINES >= 1 && LINES <= 8, "lines per run");
    const uintptr_t a = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)p >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p);      // wave-uniform by construction
    const cgeo_t p0 = (cgeo_t)(a & ~(uintptr_t)63);
    uint32_t t;
#define SMVS_TOUCH(k) "s_load_dword %0, %1, 64*" #k "\n\t"
    if constexpr (LINES == 1) asm volatile(SMVS_TOUCH(0) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p0) : "memory");
    if constexpr (LINES == 2) asm volatile(SMVS_TOUCH(0) SMVS_TOUCH(1) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p0) : "memory");
    if constexpr (LINES == 3) asm volatile(SMVS_TOUCH(0) SMVS_TOUCH(1) SMVS_TOUCH(2) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p0) : "memory");
    if constexpr (LINES == 4) asm volatile(SMVS_TOUCH(0) SMVS_TOUCH(1) SMVS_TOUCH(2) SMVS_TOUCH(3) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p0) : "memory");
    if constexpr (LINES == 5) asm volatile(SMVS_TOUCH(0) SMVS_TOUCH(1) SMVS_TOUCH(2) SMVS_TOUCH(3) SMVS_TOUCH(4) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p0) : "memory");
    if constexpr (LINES == 6) asm volatile(SMVS_TOUCH(0) SMVS_TOUCH(1) SMVS_TOUCH(2) SMVS_TOUCH(3) SMVS_TOUCH(4) SMVS_TOUCH(5) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p0) : "memory");
    if constexpr (LINES == 7) asm volatile(SMVS_TOUCH(0) SMVS_TOUCH(1) SMVS_TOUCH(2) SMVS_TOUCH(3) SMVS_TOUCH(4) SMVS_TOUCH(5) SMVS_TOUCH(6) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p0) : "memory");
    if constexpr (LINES == 8) asm volatile(SMVS_TOUCH(0) SMVS_TOUCH(1) SMVS_TOUCH(2) SMVS_TOUCH(3) SMVS_TOUCH(4) SMVS_TOUCH(5) SMVS_TOUCH(6) SMVS_TOUCH(7) "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(p0) : "memory");
#undef SMVS_TOUCH
}

// The same for four runs at once (one wait for 4 LINES lines: the four cubics of one source view)
template <int LINES>
__device__ __forceinline__ void scalar_touch4(cgeo_t pa, cgeo_t pb, cgeo_t pc, cgeo_t pd)
{
    static_assert(LINES >= 1 && LINES <= 7, "lines per run");
    auto line0 = [](cgeo_t p) {
        const uintptr_t a = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uintptr_t)p >> 32)) << 32) |
                            (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)p);
        return (cgeo_t)(a & ~(uintptr_t)63);
    };
    const cgeo_t q0 = line0(pa), q1 = line0(pb), q2 = line0(pc), q3 = line0(pd);
    uint32_t t;
#define SMVS_T4(k) "s_load_dword %0, %1, 64*" #k "\n\ts_load_dword %0, %2, 64*" #k "\n\ts_load_dword %0, %3, 64*" #k "\n\ts_load_dword %0, %4, 64*" #k "\n\t"
#define SMVS_T4_ASM(body) asm volatile(body "s_waitcnt lgkmcnt(0)" : "=&s"(t) : "s"(q0), "s"(q1), "s"(q2), "s"(q3) : "memory")
    if constexpr (LINES == 1) SMVS_T4_ASM(SMVS_T4(0));
    if constexpr (LINES == 2) SMVS_T4_ASM(SMVS_T4(0) SMVS_T4(1));
    if constexpr (LINES == 3) SMVS_T4_ASM(SMVS_T4(0) SMVS_T4(1) SMVS_T4(2));
    if constexpr (LINES == 4) SMVS_T4_ASM(SMVS_T4(0) SMVS_T4(1) SMVS_T4(2) SMVS_T4(3));
    if constexpr (LINES == 5) SMVS_T4_ASM(SMVS_T4(0) SMVS_T4(1) SMVS_T4(2) SMVS_T4(3) SMVS_T4(4));
    if constexpr (LINES == 6) SMVS_T4_ASM(SMVS_T4(0) SMVS_T4(1) SMVS_T4(2) SMVS_T4(3) SMVS_T4(4) SMVS_T4(5));
    if constexpr (LINES == 7) SMVS_T4_ASM(SMVS_T4(0) SMVS_T4(1) SMVS_T4(2) SMVS_T4(3) SMVS_T4(4) SMVS_T4(5) SMVS_T4(6));
#undef SMVS_T4_ASM
#undef SMVS_T4
}

// RPC_Obj2Photo (warping.py:218-252) at N consecutive planes of one pixel whose heights are their planes': pc -> the
// folded coefficients of THIS source at the first of the N planes, cubic 0 (wave-uniform, scalar loads); cubic i is
// `cubic_stride` doubles further (= 6 D); r = the source view's 170-vector
template <int N>
__device__ __forceinline__ void o2p_pc_xn(cgeo_t r, const RpcInv& n, const double* lat, const double* lon,
                                          cgeo_t pc, size_t cubic_stride, double* samp, double* line)
{
    double q[4][N];
    double P[N], L[N], LL[N], LP[N], PP[N], LLL[N], LPP[N], LLP[N], PPP[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
        P[u] = (lat[u] - r[I_LAT_OFF]) * n.a;
        L[u] = (lon[u] - r[I_LON_OFF]) * n.b;
        LL[u] = L[u] * L[u]; LP[u] = L[u] * P[u]; PP[u] = P[u] * P[u];
        LLL[u] = L[u] * LL[u]; LPP[u] = L[u] * PP[u]; LLP[u] = L[u] * LP[u]; PPP[u] = P[u] * PP[u];
    }
    constexpr int base[4] = {I_SNUM, I_SDEN, I_LNUM, I_LDEN};
    // one cubic at a time for all N planes: 6 N folded (one contiguous run) + 4 invariant coefficients in SGPRs (56 at
    // N = 4), then N independent chains of 1 move + 9 FMAs
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        const cgeo_t c = launder(r) + base[i];
        const cgeo_t k = launder(pc) + cubic_stride * i;
        // every coefficient of the batch requested before the first is used: ONE wait per cubic (left to itself the compiler requests
        // them in the order of use, behind three waits)
        double kk[N][PC_PER_CUBIC];
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
            for (int j = 0; j < PC_PER_CUBIC; ++j) kk[u][j] = k[PC_PER_CUBIC * u + j];
        const double c11 = c[11], c12 = c[12], c14 = c[14], c15 = c[15];
        __builtin_amdgcn_sched_barrier(0);
        double a[N];
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = to_vgpr(kk[u][0]);
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = fma(L[u], kk[u][1], a[u]);
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = fma(P[u], kk[u][2], a[u]);
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = fma(LP[u], kk[u][3], a[u]);
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = fma(LL[u], kk[u][4], a[u]);
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = fma(PP[u], kk[u][5], a[u]);
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = fma(LLL[u], c11, a[u]);
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = fma(LPP[u], c12, a[u]);
#pragma unroll
        for (int u = 0; u < N; ++u) a[u] = fma(LLP[u], c14, a[u]);
#pragma unroll
        for (int u = 0; u < N; ++u) { q[i][u] = fma(PPP[u], c15, a[u]); pin(q[i][u]); }
    }
    __builtin_amdgcn_sched_barrier(0);
    const cgeo_t rr = launder(r);
    const double so = to_vgpr(rr[I_SAMP_OFF]), lo = to_vgpr(rr[I_LINE_OFF]);     // one copy for the N points
#pragma unroll
    for (int u = 0; u < N; ++u) {
        double qs, ql;
        div_pair(q[0][u], q[1][u], q[2][u], q[3][u], qs, ql);
        samp[u] = fma(qs, rr[I_SAMP_SCALE], so);
        line[u] = fma(ql, rr[I_LINE_SCALE], lo);
    }
}

// x / d in float32, correctly rounded, for a wave-uniform d = k/2 with integer 1 <= k < 2^16 (the
// reference divides pixel coordinates by the python float (W-1)/2, warping.py:350-351): q0 = x*rd,
// r = fma(-q0, d, x) (exact), q = fma(r, rd, q0), rd = RN(1/d).  q0 + r*rd = x/d * (1 + e), |e| <
// 2^-47, while a quotient of a float by such a d is either representable or at least 2^-41
// (relative) away from every rounding boundary, so the final rounding is the IEEE one; inf/NaN
// inputs give NaN where the true quotient is inf/NaN, and both end as NaN taps.  3 instructions
// instead of the ~11 of the IEEE sequence (v_div_scale / v_rcp / v_div_fmas / v_div_fixup).
__device__ __forceinline__ float div_half_int(float x, float d, float rd)
{
    const float q0 = x * rd;
    const float r = fmaf(-q0, d, x);
    return fmaf(r, rd, q0);
}

// float -> int32 with the hardware's saturating conversion (NaN -> 0): defined for every input, unlike a C cast
__device__ __forceinline__ int cvt_i32_sat(float x)
{
    int r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

// ---- wave reductions on the DPP network (no LDS traffic, unlike ds_bpermute shuffles) ------------
// min(a), max(b), min(c), max(d) over the 64 lanes, the four reductions interleaved so that every DPP
// instruction is three instructions away from the write it reads (the DPP read-after-VALU-write hazard needs two
// wait states; the compiler's own lowering spends a v_mov and an s_nop per step).  row_shr 1,2,4,8 leave the row
// result in lane 15 of each 16-lane row (a lane whose source falls outside its row is left untouched),
// row_bcast:15 / row_bcast:31 fold the rows; the wave result ends in lane 63.
__device__ __forceinline__ void wave_minmax4(int& a, int& b, int& c, int& d)
{
#define SMVS_DPP_STEP(ctrl)                       \
    "v_min_i32_dpp %0, %0, %0 " ctrl "\n\t"       \
    "v_max_i32_dpp %1, %1, %1 " ctrl "\n\t"       \
    "v_min_i32_dpp %2, %2, %2 " ctrl "\n\t"       \
    "v_max_i32_dpp %3, %3, %3 " ctrl "\n\t"
    asm("s_nop 1\n\t"
        SMVS_DPP_STEP("row_shr:1 row_mask:0xf bank_mask:0xf")
        SMVS_DPP_STEP("row_shr:2 row_mask:0xf bank_mask:0xf")
        SMVS_DPP_STEP("row_shr:4 row_mask:0xf bank_mask:0xf")
        SMVS_DPP_STEP("row_shr:8 row_mask:0xf bank_mask:0xf")
        SMVS_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
        SMVS_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
        "s_nop 0"
        : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef SMVS_DPP_STEP
    a = __builtin_amdgcn_readlane(a, 63);
    b = __builtin_amdgcn_readlane(b, 63);
    c = __builtin_amdgcn_readlane(c, 63);
    d = __builtin_amdgcn_readlane(d, 63);
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const uint64_t u = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}

// ---- height hypotheses generated in the kernel -----------------------------------
// Stages 2 and 3 of the cascades build their hypotheses from the previous stage's height map
// (networks/casred.py:134-145 + modules/depth_range.py:4-20):
//   cur     = F.interpolate(prev, [img_h, img_w], bilinear, align_corners=False)          image resolution
//   cur_min = cur - c, cur_max = cur + c, c = float(ndepth / 2 * interval);  step = (cur_max - cur_min) / (ndepth - 1)
//   samples[d] = cur_min + d * step                                                        (B, D, img_h, img_w)
//   heights = F.interpolate(samples, [D, img_h / scale, img_w / scale], trilinear, align_corners=False)
// Generating them per pixel in the consuming kernel removes the (B,D,H,W) tensor (4 B/voxel of reads), the
// image-resolution temporaries and ~8 launches per stage.  The arithmetic below is ATen's, rounding for rounding
// (checked bit for bit against the reference on the CPU: tests/test_oracle_golden.py::test_height_hypotheses):
// every two-term interpolation is fma(w0, v0, w1 * v1); in the plane axis the resize is the identity (weight 1 / 0);
// scale 2 averages a 2x2 block with weights 1/2 (exact products), scale 1 is the identity.
struct HeightGen {
    const float* prev;          // (B, hp, wp)
    int hp, wp, ih, iw, scale;  // previous map size, image size, image size / this stage's size (1 or 2)
    float c, ndm1;              // float(ndepth / 2 * interval), float(ndepth - 1)
    // UCS-Net sampler (depth_range.py:45-86) when var != null: span prev -+ var, clamped to [rmin[b], rmax[b]]; scale == 1
    const float* var;           // (B, hp, wp) or null
    const float *rmin, *rmax;   // (B)
};

__device__ __forceinline__ float hg_interp2(float w0, float v0, float w1, float v1) { return fmaf(w0, v0, w1 * v1); }

// image-resolution pixel (Y, X) of batch item b: cur_min and step
__device__ __forceinline__ void hg_pixel(const HeightGen& g, int b, int Y, int X, float& cmin, float& step)
{
    const float sh = (float)g.hp / (float)g.ih, sw = (float)g.wp / (float)g.iw;
    const float sy = fmaxf(sh * ((float)Y + 0.5f) - 0.5f, 0.0f), sx = fmaxf(sw * ((float)X + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, g.hp - 1), x1 = min(x0 + 1, g.wp - 1);
    const float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const float* q = g.prev + (size_t)b * g.hp * g.wp;
    const float top = hg_interp2(lx0, q[y0 * g.wp + x0], lx1, q[y0 * g.wp + x1]);
    const float bot = hg_interp2(lx0, q[y1 * g.wp + x0], lx1, q[y1 * g.wp + x1]);
    const float cur = hg_interp2(ly0, top, ly1, bot);
    if (g.var) {
        // the same resize of the standard-deviation map, then the reference's masked clamps ((low - min) < 0, (high - max) > 0)
        const float* qv = g.var + (size_t)b * g.hp * g.wp;
        const float tv = hg_interp2(lx0, qv[y0 * g.wp + x0], lx1, qv[y0 * g.wp + x1]);
        const float bv = hg_interp2(lx0, qv[y1 * g.wp + x0], lx1, qv[y1 * g.wp + x1]);
        const float ev = hg_interp2(ly0, tv, ly1, bv);
        float low = cur - ev, high = cur + ev;
        const float mn = g.rmin[b], mx = g.rmax[b];
        if (low - mn < 0.0f) low = mn;
        if (high - mx > 0.0f) high = mx;
        cmin = low;
        step = __fdiv_rn(high - low, g.ndm1);
        return;
    }
    cmin = cur - g.c;
    step = __fdiv_rn((cur + g.c) - cmin, g.ndm1);
}

// per stage pixel (y, x): the 1 (scale 1) or 4 (scale 2) image-resolution pixels under it
struct HeightPix { float cmin[4], step[4]; };

__device__ __forceinline__ void hg_prepare(const HeightGen& g, int b, int y, int x, HeightPix& hp)
{
    if (g.scale == 1) {
        hg_pixel(g, b, y, x, hp.cmin[0], hp.step[0]);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) hg_pixel(g, b, 2 * y + (k >> 1), 2 * x + (k & 1), hp.cmin[k], hp.step[k]);
    }
}

__device__ __forceinline__ float hg_height(const HeightGen& g, const HeightPix& hp, int d)
{
    const float fd = (float)d;
    if (g.var) return (hp.cmin[0] + hp.step[0] * fd) + 1e-12f;          // low + step * i + eps (depth_range.py:80)
    if (g.scale == 1) return hp.cmin[0] + fd * hp.step[0];
    const float a = hp.cmin[0] + fd * hp.step[0], bb = hp.cmin[1] + fd * hp.step[1];
    const float c = hp.cmin[2] + fd * hp.step[2], dd = hp.cmin[3] + fd * hp.step[3];
    return hg_interp2(0.5f, hg_interp2(0.5f, a, 0.5f, bb), 0.5f, hg_interp2(0.5f, c, 0.5f, dd));
}

// Where a kernel takes its heights from: (B,D) planes, a (B,D,H,W) tensor, or the generator above.
enum { HEIGHT_PLANES = 0, HEIGHT_TENSOR = 1, HEIGHT_GENERATED = 2 };

// ---- raw buffer access -------------------------------------------------------------------------
typedef int32_t i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ float llvm_raw_buffer_load_f32(i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.load.f32");

struct BufRsrc { i32x4 v; };

// Descriptor over `bytes` bytes at `base` (wave-uniform).  Raw buffer, stride 0: a load is in
// range iff voffset + soffset + 4 <= bytes (on gfx950 the scalar offset IS part of the range check
// -- measured: with num_records = one plane every channel >= 1 read back 0).  `bytes` therefore
// spans the whole (C,H,W) block of one batch item, the channel plane offset rides in the scalar
// offset, and a dropped tap carries voffset = 0x80000000, out of range for any channel.
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, uint32_t bytes)
{
    const uint64_t a = (uint64_t)base;
    BufRsrc r;
    r.v.x = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)a);
    r.v.y = (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));  // stride 0, no swizzle
    r.v.z = (int32_t)__builtin_amdgcn_readfirstlane(bytes);
    r.v.w = 0x00020000;                                                    // gfx9-family raw dword access
    return r;
}

__device__ void llvm_raw_buffer_store_f32(float v, i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.store.f32");
// float atomic add through a buffer descriptor (the returned old value is ignored by every caller: the no-return form is issued);
// a lane whose offset is out of range (SMVS_OOB) is dropped by the range check -- no branch around the instruction
__device__ float llvm_raw_buffer_atomic_fadd_f32(float v, i32x4 rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.buffer.atomic.fadd.f32");

// The two helpers below write M0 (the LDS-DMA destination register) inside the asm and list it as clobbered, which is
// the correct declaration: the compiler never keeps a value in M0 across a statement that clobbers it (its own M0 users
// -- LDS-DMA intrinsics, s_sendmsg, s_movrel -- set M0 immediately before each use).  clang's generic "reserved register
// in the clobber list" warning does not apply to that use, so it is switched off for exactly these definitions.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// LDS-DMA: one dword per lane from a buffer straight into LDS at lds_byte_addr + 4*lane, no VGPR
// round trip; out-of-range lanes deposit 0.  Issued from inline asm so that hipcc does not drain
// vmcnt(0) before every LDS read (it cannot tell the two staging buffers apart): completion is
// tracked by hand with counted s_waitcnt vmcnt(N) (vmcnt retires in issue order).
__device__ __forceinline__ void dma_dword_to_lds(const BufRsrc& rs, uint32_t lds_byte_addr, uint32_t voffset,
                                                 int soffset)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds_byte_addr), "v"(voffset), "s"(rs.v), "s"(soffset) : "memory", "m0");
}

// Same, with the LDS destination = lds_base (SGPR) + a compile-time byte offset formed by ONE scalar add into
// M0.  Passing precomputed destinations instead makes the compiler keep every (source,row,half) address as a
// loop-invariant SGPR; there are more of them than SGPRs, so they end up spilled to VGPR lanes and each use
// costs a v_readlane on the VALU (measured: 20 per channel pair in the cost-volume kernel).
template <int OFF>
__device__ __forceinline__ void dma_dword_to_lds_at(const BufRsrc& rs, uint32_t lds_base, uint32_t voffset, int soffset)
{
    asm volatile("s_add_u32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dword %1, %2, %3 offen lds"
                 :: "s"(lds_base), "v"(voffset), "s"(rs.v), "s"(soffset), "n"(OFF) : "memory", "m0", "scc");
}

// 16 bytes per lane: lane l deposits the four dwords at buffer offset voffset + soffset into LDS at
// m0 + 16 * l (gfx950's b128 LDS-DMA), i.e. one instruction lays down 1 KiB of LDS from 64 independent 16-byte
// chunks.  The range check covers the whole chunk: a lane whose offset is out of range deposits four zeros
// (tests/test_hip_parity.py::test_costvol_edge_boxes pins both on the hardware).  Costs the texture path 6.9 ns
// per CU against 6.9 ns for ONE dword instruction of the channel-interleaved map (tools/ubench_dma.hip).
template <int OFF>
__device__ __forceinline__ void dma_x4_to_lds_at(const BufRsrc& rs, uint32_t lds_base, uint32_t voffset, int soffset)
{
    asm volatile("s_add_u32 m0, %0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_base), "v"(voffset), "s"(rs.v), "s"(soffset), "n"(OFF) : "memory", "m0", "scc");
}

// Same, destination in an SGPR (the shared-box form: a wave issues the instructions dealt to it, their places in the
// box set are wave-uniform run-time values -- a handful per wave, so they stay in SGPRs)
#ifndef SMVS_DMA_SHARED_POLICY_ID
#define SMVS_DMA_SHARED_POLICY_ID 0        // cache policy of the shared-box staging loads (A/B switch of profiling builds): 0 default, 1 nt, 2 sc1, 3 sc0 sc1
#endif
#if SMVS_DMA_SHARED_POLICY_ID == 1
#define SMVS_DMA_SHARED_POLICY " nt"
#elif SMVS_DMA_SHARED_POLICY_ID == 2
#define SMVS_DMA_SHARED_POLICY " sc1"
#elif SMVS_DMA_SHARED_POLICY_ID == 3
#define SMVS_DMA_SHARED_POLICY " sc0 sc1"
#else
#define SMVS_DMA_SHARED_POLICY ""
#endif
__device__ __forceinline__ void dma_x4_to_lds(const BufRsrc& rs, uint32_t lds_byte_addr, uint32_t voffset, int soffset)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen" SMVS_DMA_SHARED_POLICY " lds"
                 :: "s"(lds_byte_addr), "v"(voffset), "s"(rs.v), "s"(soffset) : "memory", "m0");
}

#pragma clang diagnostic pop

// The four corners of one tap for a channel pair out of the PLANAR staging layout [row][channel of the pair][column]
// (row pitch 2*BW dwords): one ds_read2_b32 per corner fetches the corner's dword of both channels into a register
// pair, ready for v_pk_*_f32.  The two 8-bit dword offsets are all the immediate there is, so buffer parity and source
// live in the address register.
template <int BW>
__device__ __forceinline__ void lds_read_tap_planar(uint32_t addr, f32x2& nw, f32x2& ne, f32x2& sw, f32x2& se)
{
    static_assert(3 * BW + 1 <= 255, "ds_read2_b32 offsets are 8-bit dword counts");
    asm volatile("ds_read2_b32 %0, %4 offset0:%5 offset1:%6\n\t"
                 "ds_read2_b32 %1, %4 offset0:%7 offset1:%8\n\t"
                 "ds_read2_b32 %2, %4 offset0:%9 offset1:%10\n\t"
                 "ds_read2_b32 %3, %4 offset0:%11 offset1:%12"
                 : "=&v"(nw), "=&v"(ne), "=&v"(sw), "=&v"(se)
                 : "v"(addr), "n"(0), "n"(BW), "n"(1), "n"(BW + 1), "n"(2 * BW), "n"(3 * BW), "n"(2 * BW + 1), "n"(3 * BW + 1)
                 : "memory");
}

// The four corners of one tap for a channel pair, as four ds_read_b64 (256 B/clk) rather than the
// two ds_read2_b64 (128 B/clk) hipcc merges them into.  Issued from inline asm, so completion is
// tracked by hand: LDS reads of a wave return in order, lds_wait<N>() = s_waitcnt lgkmcnt(N) and
// carries the registers it protects as in/out operands so that no use can be scheduled above it.
// OFF = byte offset of the staging buffer, ROW = byte distance to the south row.
template <int OFF, int ROW>
__device__ __forceinline__ void lds_read_tap(uint32_t addr, f32x2& nw, f32x2& ne, f32x2& sw, f32x2& se)
{
    static_assert(OFF + ROW + 8 < 65536, "ds_read immediate offset");
    asm volatile("ds_read_b64 %0, %4 offset:%5\n\t"
                 "ds_read_b64 %1, %4 offset:%6\n\t"
                 "ds_read_b64 %2, %4 offset:%7\n\t"
                 "ds_read_b64 %3, %4 offset:%8"
                 : "=&v"(nw), "=&v"(ne), "=&v"(sw), "=&v"(se)
                 : "v"(addr), "n"(OFF), "n"(OFF + 8), "n"(OFF + ROW), "n"(OFF + ROW + 8)
                 : "memory");
}

template <int N>
__device__ __forceinline__ void lds_wait(f32x2& a0, f32x2& a1, f32x2& a2, f32x2& a3,
                                         f32x2& b0, f32x2& b1, f32x2& b2, f32x2& b3)
{
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
                 : "n"(N) : "memory");
}

__device__ __forceinline__ uint32_t lds_addr(const void* p)
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

#define SMVS_OOB 0x80000000u   // voffset (+ any channel offset < 2^31) >= num_records: reads back 0

// ---- sampler -------------------------------------------------------------------------------------
struct Tap {
    uint32_t o_nw, o_ne, o_sw, o_se;   // byte offsets inside one H*W plane, or SMVS_OOB (tap dropped)
    float nw, ne, sw, se;
    int x0, y0;                        // north-west cell, -1 .. W-1 / -1 .. H-1 (0 where that axis is out of reach)
    float fw, fn;                      // fractions the weights are products of: nw = (1-fn)(1-fw), ne = (1-fn) fw, sw = fn (1-fw), se = fn fw
};

// Normalised grid coordinate -> tap.  fx = W/2, fy = H/2 (exact in float32).
__device__ __forceinline__ Tap tap_from_grid(float gx, float gy, int H, int W)
{
    Tap t;
    const float x = fmaf(gx + 1.0f, (float)W * 0.5f, -0.5f);   // ATen unnormalise, align_corners=False
    const float y = fmaf(gy + 1.0f, (float)H * 0.5f, -0.5f);
    const float xw = floorf(x), yn = floorf(y);
    const float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
    t.nw = s * e; t.ne = s * w; t.sw = n * e; t.se = n * w;
    t.fw = w; t.fn = n;
    // bounds in float: NaN / huge coordinates compare false everywhere -> all four taps dropped
    const bool xin0 = (xw >= 0.0f) && (xw <= (float)(W - 1));
    const bool xin1 = (xw >= -1.0f) && (xw <= (float)(W - 2));
    const bool yin0 = (yn >= 0.0f) && (yn <= (float)(H - 1));
    const bool yin1 = (yn >= -1.0f) && (yn <= (float)(H - 2));
    const int x0 = (xin0 || xin1) ? (int)xw : 0;
    const int y0 = (yin0 || yin1) ? (int)yn : 0;
    const int base = (y0 * W + x0) * 4;
    t.x0 = x0; t.y0 = y0;
    t.o_nw = (xin0 && yin0) ? (uint32_t)base : SMVS_OOB;
    t.o_ne = (xin1 && yin0) ? (uint32_t)(base + 4) : SMVS_OOB;
    t.o_sw = (xin0 && yin1) ? (uint32_t)(base + 4 * W) : SMVS_OOB;
    t.o_se = (xin1 && yin1) ? (uint32_t)(base + 4 * W + 4) : SMVS_OOB;
    return t;
}

// float32 pixel coordinates (samp.float(), line.float()) -> tap.  half_wm1 = (W-1)/2 as float32
// (the python scalar the reference divides by), warping.py:350-351.
__device__ __forceinline__ Tap tap_from_pixel(float px, float py, int H, int W, float half_wm1, float half_hm1)
{
    const float gx = px / half_wm1 - 1.0f;
    const float gy = py / half_hm1 - 1.0f;
    return tap_from_grid(gx, gy, H, W);
}

// One channel of one source view.  `choff` = c*H*W*4, wave-uniform (scalar offset).
__device__ __forceinline__ float tap_fetch(const BufRsrc& rs, const Tap& t, int choff)
{
    const float a = llvm_raw_buffer_load_f32(rs.v, (int)t.o_nw, choff, 0);
    const float b = llvm_raw_buffer_load_f32(rs.v, (int)t.o_ne, choff, 0);
    const float c = llvm_raw_buffer_load_f32(rs.v, (int)t.o_sw, choff, 0);
    const float d = llvm_raw_buffer_load_f32(rs.v, (int)t.o_se, choff, 0);
    float r = a * t.nw;
    r = fmaf(b, t.ne, r);
    r = fmaf(c, t.sw, r);
    r = fmaf(d, t.se, r);
    return r;
}

// Two channels (c, c+1) of one source view at once: the arithmetic is the same per lane element,
// written on float2 so it maps onto v_pk_mul_f32 / v_pk_fma_f32.

__device__ __forceinline__ f32x2 tap_fetch2(const BufRsrc& rs, const Tap& t, int choff, int chstride)
{
    f32x2 a, b, c, d;
    a.x = llvm_raw_buffer_load_f32(rs.v, (int)t.o_nw, choff, 0);
    b.x = llvm_raw_buffer_load_f32(rs.v, (int)t.o_ne, choff, 0);
    c.x = llvm_raw_buffer_load_f32(rs.v, (int)t.o_sw, choff, 0);
    d.x = llvm_raw_buffer_load_f32(rs.v, (int)t.o_se, choff, 0);
    a.y = llvm_raw_buffer_load_f32(rs.v, (int)t.o_nw, choff + chstride, 0);
    b.y = llvm_raw_buffer_load_f32(rs.v, (int)t.o_ne, choff + chstride, 0);
    c.y = llvm_raw_buffer_load_f32(rs.v, (int)t.o_sw, choff + chstride, 0);
    d.y = llvm_raw_buffer_load_f32(rs.v, (int)t.o_se, choff + chstride, 0);
    f32x2 r = a * t.nw;
    r = __builtin_elementwise_fma(b, (f32x2)(t.ne), r);
    r = __builtin_elementwise_fma(c, (f32x2)(t.sw), r);
    r = __builtin_elementwise_fma(d, (f32x2)(t.se), r);
    return r;
}

// ---- hand-ordered packed arithmetic of the staged cost-volume kernel -----------------------------------------
// gfx950 needs one wait state between a v_pk_*_f32 and a VALU instruction that reads its result; hipcc's
// scheduler does not model that and its hazard pass then fills the gaps with s_nop (66 per channel pair in the
// round-2 kernel = one issue slot in five).  The blocks below fix an order in which every result is consumed at
// least two instructions after it was produced: the taps of TWO sources run interleaved, and the sum and
// sum-of-squares strands of the variance alternate.  The four bilinear weights of a tap stay in two register
// pairs {nw, ne}, {sw, se}; op_sel broadcasts the half a multiply needs (no {w, w} copies: -64 VGPRs at 8 planes).
// Arithmetic and rounding sequence are exactly tap_fetch2 / div_by_views2 below (ATen's sampler, true division
// by the view count), so the bits do not change.

// a = bilinear(ca[0..3]; wa), b = bilinear(cb[0..3]; wb):  r = c0*nw; r = fma(c1, ne, r); r = fma(c2, sw, r); r = fma(c3, se, r)
__device__ __forceinline__ void pk_bilinear2(f32x2& a, f32x2& b,
                                             const f32x2& a0, const f32x2& a1, const f32x2& a2, const f32x2& a3, const f32x2& wan, const f32x2& was,
                                             const f32x2& b0, const f32x2& b1, const f32x2& b2, const f32x2& b3, const f32x2& wbn, const f32x2& wbs)
{
    asm volatile("v_pk_mul_f32 %0, %2, %6 op_sel_hi:[1,0]\n\t"
                 "v_pk_mul_f32 %1, %8, %12 op_sel_hi:[1,0]\n\t"
                 "v_pk_fma_f32 %0, %3, %6, %0 op_sel:[0,1,0]\n\t"
                 "v_pk_fma_f32 %1, %9, %12, %1 op_sel:[0,1,0]\n\t"
                 "v_pk_fma_f32 %0, %4, %7, %0 op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 %1, %10, %13, %1 op_sel_hi:[1,0,1]\n\t"
                 "v_pk_fma_f32 %0, %5, %7, %0 op_sel:[0,1,0]\n\t"
                 "v_pk_fma_f32 %1, %11, %13, %1 op_sel:[0,1,0]"
                 : "=&v"(a), "=&v"(b)
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(wan), "v"(was), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(wbn), "v"(wbs));
}

// One source only (odd source counts): the chain is serial, every step waits one state
__device__ __forceinline__ void pk_bilinear1(f32x2& a, const f32x2& a0, const f32x2& a1, const f32x2& a2, const f32x2& a3,
                                             const f32x2& wan, const f32x2& was)
{
    asm volatile("v_pk_mul_f32 %0, %1, %5 op_sel_hi:[1,0]\n\ts_nop 0\n\t"
                 "v_pk_fma_f32 %0, %2, %5, %0 op_sel:[0,1,0]\n\ts_nop 0\n\t"
                 "v_pk_fma_f32 %0, %3, %6, %0 op_sel_hi:[1,0,1]\n\ts_nop 0\n\t"
                 "v_pk_fma_f32 %0, %4, %6, %0 op_sel:[0,1,0]\n\ts_nop 0"
                 : "=&v"(a) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(wan), "v"(was));
}

// sum += a (+ b), sq += a*a (+ b*b) for a plane whose sources are not finished yet.  Consumes a and b.
__device__ __forceinline__ void pk_accumulate2(f32x2& s, f32x2& t, const f32x2& sin, const f32x2& tin, f32x2& a, f32x2& b)
{
    asm volatile("v_pk_add_f32 %0, %4, %2\n\t"          // s = sin + a
                 "v_pk_mul_f32 %2, %2, %2\n\t"          // a = a*a
                 "v_pk_add_f32 %0, %0, %3\n\t"          // s += b
                 "v_pk_mul_f32 %3, %3, %3\n\t"          // b = b*b
                 "v_pk_add_f32 %1, %5, %2\n\t"          // t = tin + a*a
                 "s_nop 0\n\t"
                 "v_pk_add_f32 %1, %1, %3\n\t"          // t += b*b
                 "s_nop 0"
                 : "=&v"(s), "=&v"(t), "+v"(a), "+v"(b) : "v"(sin), "v"(tin));
}

__device__ __forceinline__ void pk_accumulate1(f32x2& s, f32x2& t, const f32x2& sin, const f32x2& tin, f32x2& a)
{
    asm volatile("v_pk_add_f32 %0, %3, %2\n\t"
                 "v_pk_mul_f32 %2, %2, %2\n\t"
                 "s_nop 0\n\t"
                 "v_pk_add_f32 %1, %4, %2\n\t"
                 "s_nop 0"
                 : "=&v"(s), "=&v"(t), "+v"(a) : "v"(sin), "v"(tin));
}

// Last sources of a plane: accumulate a (+ b), then mean = sum / V, meansq = sq / V (div_by_views2: q0 = x*rv,
// r = fma(-q0, V, x), q = fma(r, rv, q0)) and mean*mean.  Leaves s = mean*mean and t = sq / V; the caller finishes with
// pk_variance(): var = t - s, TWO or more instructions later (the block ends on the write of t).  rvv = {1/V, V}.
__device__ __forceinline__ void pk_finish2(f32x2& s, f32x2& t, const f32x2& sin, const f32x2& tin, f32x2& a, f32x2& b, const f32x2& rvv)
{
    f32x2 m0, q0;
    asm volatile("v_pk_add_f32 %0, %6, %4\n\t"                                   //  1 s = sin + a
                 "v_pk_mul_f32 %4, %4, %4\n\t"                                   //  2 a = a*a
                 "v_pk_add_f32 %0, %0, %5\n\t"                                   //  3 s += b
                 "v_pk_mul_f32 %5, %5, %5\n\t"                                   //  4 b = b*b
                 "v_pk_add_f32 %1, %7, %4\n\t"                                   //  5 t = tin + a*a
                 "v_pk_mul_f32 %2, %0, %8 op_sel_hi:[1,0]\n\t"                   //  6 m0 = s * rv
                 "v_pk_add_f32 %1, %1, %5\n\t"                                   //  7 t += b*b
                 "v_pk_fma_f32 %4, %2, %8, %0 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   //  8 a = fma(-m0, V, s)
                 "v_pk_mul_f32 %3, %1, %8 op_sel_hi:[1,0]\n\t"                   //  9 q0 = t * rv
                 "v_pk_fma_f32 %0, %4, %8, %2 op_sel_hi:[1,0,1]\n\t"             // 10 s = fma(a, rv, m0)      mean
                 "v_pk_fma_f32 %5, %3, %8, %1 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   // 11 b = fma(-q0, V, t)
                 "v_pk_mul_f32 %0, %0, %0\n\t"                                   // 12 s = mean*mean
                 "v_pk_fma_f32 %1, %5, %8, %3 op_sel_hi:[1,0,1]"                 // 13 t = fma(b, rv, q0)      meansq
                 : "=&v"(s), "=&v"(t), "=&v"(m0), "=&v"(q0), "+v"(a), "+v"(b) : "v"(sin), "v"(tin), "v"(rvv));
}

__device__ __forceinline__ void pk_finish1(f32x2& s, f32x2& t, const f32x2& sin, const f32x2& tin, f32x2& a, const f32x2& rvv)
{
    f32x2 m0, q0, r;
    asm volatile("v_pk_add_f32 %0, %6, %5\n\t"                                   //  1 s = sin + a
                 "v_pk_mul_f32 %5, %5, %5\n\t"                                   //  2 a = a*a
                 "v_pk_mul_f32 %2, %0, %8 op_sel_hi:[1,0]\n\t"                   //  3 m0 = s * rv
                 "v_pk_add_f32 %1, %7, %5\n\t"                                   //  4 t = tin + a*a
                 "v_pk_fma_f32 %4, %2, %8, %0 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   //  5 r = fma(-m0, V, s)
                 "v_pk_mul_f32 %3, %1, %8 op_sel_hi:[1,0]\n\t"                   //  6 q0 = t * rv
                 "v_pk_fma_f32 %0, %4, %8, %2 op_sel_hi:[1,0,1]\n\t"             //  7 s = fma(r, rv, m0)
                 "v_pk_fma_f32 %5, %3, %8, %1 op_sel:[0,1,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"   //  8 a = fma(-q0, V, t)
                 "v_pk_mul_f32 %0, %0, %0\n\t"                                   //  9 s = mean*mean
                 "v_pk_fma_f32 %1, %5, %8, %3 op_sel_hi:[1,0,1]"                 // 10 t = fma(a, rv, q0)
                 : "=&v"(s), "=&v"(t), "=&v"(m0), "=&v"(q0), "=&v"(r), "+v"(a) : "v"(sin), "v"(tin), "v"(rvv));
}

// var = meansq - mean*mean.  NOP = 1 when it directly follows pk_finish (nothing in between to cover the wait state).
template <int NOP>
__device__ __forceinline__ f32x2 pk_variance(const f32x2& t, const f32x2& s)
{
    f32x2 v;
    if (NOP) asm volatile("s_nop 0\n\tv_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v) : "v"(t), "v"(s));
    else     asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(v) : "v"(t), "v"(s));
    return v;
}

// ---- contract-tolerance ("fused") arithmetic of the cost-volume build: smvs_set_arith(SMVS_ARITH_FUSED) --------------
// The variance of V values is shift invariant, so it is taken of the DIFFERENCES to the ref feature, d_s = warped_s - ref
// (d_ref = 0), which come out of the bilinear chain for free when the chain starts as fma(c_nw, w_nw, -ref):
//   var = (sum d_s^2) / V - ((sum d_s) / V)^2.
// The constant factors ride on the tap weights (scaled once per tap in the geometry phase, i.e. once per 16 channel pairs)
// and on the ref feature (once per channel pair and 8 planes):
//   2 sources (3 views), weights * sqrt(2)/3:  var = da^2 - da db + db^2          = fma(da, da - db, db*db)     3 operations
//   1 source  (2 views), weights * 1/2:        var = d*d                                                         1 operation
//   S >= 3 sources,      weights * 1/sqrt(V):  var = T - S*S/V, S = sum d, T = sum d^2  = fma(-(S*rV), S, T)    2 S + 1 operations
// against 14 (2 sources) for the reference's sequence  sum, sum of squares, two true divisions, mean^2, subtract.  The
// float64 geometry, the float32 tap coordinates and the unscaled weights are those of the exact instance bit for bit; the
// result differs from the reference's by rounding only, and is the more accurate of the two: the reference's
// meansq - mean^2 cancels, the difference form does not (tests/test_fused_arith.py measures both against a float64
// evaluation).  The reference sequence stays available (SMVS_ARITH_EXACT) and is what every bit-level test runs.
// scalar form (direct-gather kernel and the staged kernel's oversized-box path); same operations as the packed blocks
template <int NSRC>
__device__ __forceinline__ float fused_variance(const float (&d)[NSRC], float rV)
{
    if constexpr (NSRC == 1) return d[0] * d[0];
    else if constexpr (NSRC == 2) return fmaf(d[0], d[0] - d[1], d[1] * d[1]);
    else {
        float S = d[0] + d[1];
        float T = d[0] * d[0];
        T = fmaf(d[1], d[1], T);
#pragma unroll
        for (int s = 2; s < NSRC; ++s) { S = S + d[s]; T = fmaf(d[s], d[s], T); }
        const float p = S * rV;
        return fmaf(-p, S, T);
    }
}

// d = c_nw*w_nw - rk, + c_ne*w_ne, + c_sw*w_sw, + c_se*w_se with the scaled weights ((1-fn)*k)*(1-fw), ...
__device__ __forceinline__ float fused_tap_diff(const BufRsrc& rs, const Tap& t, int choff, float kw, float rk)
{
    const float sk = (1.0f - t.fn) * kw, nk = t.fn * kw, e = 1.0f - t.fw;
    const float a = llvm_raw_buffer_load_f32(rs.v, (int)t.o_nw, choff, 0);
    const float b = llvm_raw_buffer_load_f32(rs.v, (int)t.o_ne, choff, 0);
    const float c = llvm_raw_buffer_load_f32(rs.v, (int)t.o_sw, choff, 0);
    const float d = llvm_raw_buffer_load_f32(rs.v, (int)t.o_se, choff, 0);
    float r = fmaf(a, sk * e, -rk);
    r = fmaf(b, sk * t.fw, r);
    r = fmaf(c, nk * e, r);
    r = fmaf(d, nk * t.fw, r);
    return r;
}

// rk = ref * k for a channel pair (kv = {k, .}); trailing wait state: the next reader may be anything
__device__ __forceinline__ f32x2 pk_scale_lo(const f32x2& r, const f32x2& kv)
{
    f32x2 o;
    asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\ts_nop 0" : "=v"(o) : "v"(r), "v"(kv));
    return o;
}

// Bilinear chains of two sources on differences, a = sum c_a w_a - rk, b = sum c_b w_b - rk, interleaved so that no packed
// result is read in the slot behind its producer; the TAIL of the previous unit (pa, pb: its two differences) rides in the
// gaps.  Tails:
//   SMVS_TAIL_VAR3  (2 sources)   e = pa - pb; f = pb*pb; var = fma(pa, e, f)
//   SMVS_TAIL_ACC0  (first unit of a plane, more to come)   S = pa + pb; T = pa*pa; T = fma(pb, pb, T)
//   SMVS_TAIL_ACC   (middle unit)  S += pa; T = fma(pa, pa, T); S += pb; T = fma(pb, pb, T)
//   SMVS_TAIL_FIN   (last unit of a plane)  the same, then p = S*rV; var = fma(-p, S, T)
#define SMVS_FB_A0 "v_pk_fma_f32 %[a], %[a0], %[wan], %[rk] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
#define SMVS_FB_B0 "v_pk_fma_f32 %[b], %[b0], %[wbn], %[rk] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"
#define SMVS_FB_A1 "v_pk_fma_f32 %[a], %[a1], %[wan], %[a] op_sel:[0,1,0]\n\t"
#define SMVS_FB_B1 "v_pk_fma_f32 %[b], %[b1], %[wbn], %[b] op_sel:[0,1,0]\n\t"
#define SMVS_FB_A2 "v_pk_fma_f32 %[a], %[a2], %[was], %[a] op_sel_hi:[1,0,1]\n\t"
#define SMVS_FB_B2 "v_pk_fma_f32 %[b], %[b2], %[wbs], %[b] op_sel_hi:[1,0,1]\n\t"
#define SMVS_FB_A3 "v_pk_fma_f32 %[a], %[a3], %[was], %[a] op_sel:[0,1,0]\n\t"
#define SMVS_FB_B3 "v_pk_fma_f32 %[b], %[b3], %[wbs], %[b] op_sel:[0,1,0]\n\t"
#define SMVS_FB_INPUTS [a0] "v"(a0), [a1] "v"(a1), [a2] "v"(a2), [a3] "v"(a3), [wan] "v"(wan), [was] "v"(was), \
                       [b0] "v"(b0), [b1] "v"(b1), [b2] "v"(b2), [b3] "v"(b3), [wbn] "v"(wbn), [wbs] "v"(wbs), [rk] "v"(rk)
#define SMVS_FB_ARGS const f32x2& a0, const f32x2& a1, const f32x2& a2, const f32x2& a3, const f32x2& wan, const f32x2& was, \
                     const f32x2& b0, const f32x2& b1, const f32x2& b2, const f32x2& b3, const f32x2& wbn, const f32x2& wbs, const f32x2& rk

__device__ __forceinline__ void pk_fbil2(f32x2& a, f32x2& b, SMVS_FB_ARGS)
{
    asm volatile(SMVS_FB_A0 SMVS_FB_B0 SMVS_FB_A1 SMVS_FB_B1 SMVS_FB_A2 SMVS_FB_B2 SMVS_FB_A3 SMVS_FB_B3
                 : [a] "=&v"(a), [b] "=&v"(b) : SMVS_FB_INPUTS);
}

__device__ __forceinline__ void pk_fbil2_var3(f32x2& a, f32x2& b, f32x2& var, const f32x2& pa, const f32x2& pb, SMVS_FB_ARGS)
{
    f32x2 e, f;
    asm volatile(SMVS_FB_A0 SMVS_FB_B0
                 "v_pk_add_f32 %[e], %[pa], %[pb] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 SMVS_FB_A1 SMVS_FB_B1
                 "v_pk_mul_f32 %[f], %[pb], %[pb]\n\t"
                 SMVS_FB_A2 SMVS_FB_B2
                 "v_pk_fma_f32 %[var], %[pa], %[e], %[f]\n\t"
                 SMVS_FB_A3 SMVS_FB_B3
                 : [a] "=&v"(a), [b] "=&v"(b), [var] "=&v"(var), [e] "=&v"(e), [f] "=&v"(f)
                 : SMVS_FB_INPUTS, [pa] "v"(pa), [pb] "v"(pb));
}

__device__ __forceinline__ f32x2 pk_var3_tail(const f32x2& pa, const f32x2& pb)
{
    f32x2 e, f, var;
    asm volatile("s_nop 0\n\t"
                 "v_pk_add_f32 %[e], %[pa], %[pb] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                 "v_pk_mul_f32 %[f], %[pb], %[pb]\n\t"
                 "s_nop 0\n\t"
                 "v_pk_fma_f32 %[var], %[pa], %[e], %[f]"
                 : [var] "=&v"(var), [e] "=&v"(e), [f] "=&v"(f) : [pa] "v"(pa), [pb] "v"(pb));
    return var;
}

__device__ __forceinline__ void pk_fbil2_acc0(f32x2& a, f32x2& b, f32x2& S, f32x2& T, const f32x2& pa, const f32x2& pb, SMVS_FB_ARGS)
{
    asm volatile(SMVS_FB_A0 SMVS_FB_B0
                 "v_pk_add_f32 %[S], %[pa], %[pb]\n\t"
                 SMVS_FB_A1 SMVS_FB_B1
                 "v_pk_mul_f32 %[T], %[pa], %[pa]\n\t"
                 SMVS_FB_A2 SMVS_FB_B2
                 "v_pk_fma_f32 %[T], %[pb], %[pb], %[T]\n\t"
                 SMVS_FB_A3 SMVS_FB_B3
                 : [a] "=&v"(a), [b] "=&v"(b), [S] "=&v"(S), [T] "=&v"(T)
                 : SMVS_FB_INPUTS, [pa] "v"(pa), [pb] "v"(pb));
}

__device__ __forceinline__ void pk_fbil2_acc(f32x2& a, f32x2& b, f32x2& S, f32x2& T, const f32x2& pa, const f32x2& pb, SMVS_FB_ARGS)
{
    asm volatile(SMVS_FB_A0 SMVS_FB_B0
                 "v_pk_add_f32 %[S], %[S], %[pa]\n\t"
                 SMVS_FB_A1 SMVS_FB_B1
                 "v_pk_fma_f32 %[T], %[pa], %[pa], %[T]\n\t"
                 SMVS_FB_A2 SMVS_FB_B2
                 "v_pk_add_f32 %[S], %[S], %[pb]\n\t"
                 SMVS_FB_A3 SMVS_FB_B3
                 "v_pk_fma_f32 %[T], %[pb], %[pb], %[T]"
                 : [a] "=&v"(a), [b] "=&v"(b), [S] "+v"(S), [T] "+v"(T)
                 : SMVS_FB_INPUTS, [pa] "v"(pa), [pb] "v"(pb));
}

// rvv = {1/V, V}
__device__ __forceinline__ void pk_fbil2_fin(f32x2& a, f32x2& b, f32x2& var, f32x2& S, f32x2& T, const f32x2& pa, const f32x2& pb,
                                             const f32x2& rvv, SMVS_FB_ARGS)
{
    f32x2 p;
    asm volatile(SMVS_FB_A0 SMVS_FB_B0
                 "v_pk_add_f32 %[S], %[S], %[pa]\n\t"
                 SMVS_FB_A1 SMVS_FB_B1
                 "v_pk_fma_f32 %[T], %[pa], %[pa], %[T]\n\t"
                 SMVS_FB_A2 SMVS_FB_B2
                 "v_pk_add_f32 %[S], %[S], %[pb]\n\t"
                 SMVS_FB_A3 SMVS_FB_B3
                 "v_pk_fma_f32 %[T], %[pb], %[pb], %[T]\n\t"
                 "v_pk_mul_f32 %[p], %[S], %[rvv] op_sel_hi:[1,0]\n\t"
                 "s_nop 0\n\t"
                 "v_pk_fma_f32 %[var], %[p], %[S], %[T] neg_lo:[1,0,0] neg_hi:[1,0,0]"
                 : [a] "=&v"(a), [b] "=&v"(b), [var] "=&v"(var), [p] "=&v"(p), [S] "+v"(S), [T] "+v"(T)
                 : SMVS_FB_INPUTS, [pa] "v"(pa), [pb] "v"(pb), [rvv] "v"(rvv));
}

__device__ __forceinline__ f32x2 pk_fin_tail(f32x2& S, f32x2& T, const f32x2& pa, const f32x2& pb, const f32x2& rvv)
{
    f32x2 p, var;
    asm volatile("s_nop 0\n\t"
                 "v_pk_add_f32 %[S], %[S], %[pa]\n\t"
                 "v_pk_fma_f32 %[T], %[pa], %[pa], %[T]\n\t"
                 "v_pk_add_f32 %[S], %[S], %[pb]\n\t"
                 "v_pk_fma_f32 %[T], %[pb], %[pb], %[T]\n\t"
                 "v_pk_mul_f32 %[p], %[S], %[rvv] op_sel_hi:[1,0]\n\t"
                 "s_nop 0\n\t"
                 "v_pk_fma_f32 %[var], %[p], %[S], %[T] neg_lo:[1,0,0] neg_hi:[1,0,0]"
                 : [var] "=&v"(var), [p] "=&v"(p), [S] "+v"(S), [T] "+v"(T) : [pa] "v"(pa), [pb] "v"(pb), [rvv] "v"(rvv));
    return var;
}

__device__ __forceinline__ f32x2 div_by_views2(f32x2 x, float v, float rv)
{
    const f32x2 q0 = x * rv;
    const f32x2 r = __builtin_elementwise_fma(-q0, (f32x2)(v), x);
    return __builtin_elementwise_fma(r, (f32x2)(rv), q0);
}

// x / v for a small integer-valued float v (the view count), correctly rounded in 3 ops:
// q0 = x*rv, r = fma(-q0, v, x) (exact), q = fma(r, rv, q0); rv = RN(1/v).  For v <= 8 the quotient
// x/v is never within 2^-27 relative of a rounding boundary while q0 + r*rv differs from it by
// < 2^-46 relative, so the final rounding is the IEEE one (checked bit-for-bit against the oracle's
// true division in tests/test_hip_parity.py).
__device__ __forceinline__ float div_by_views(float x, float v, float rv)
{
    const float q0 = x * rv;
    const float r = fmaf(-q0, v, x);
    return fmaf(r, rv, q0);
}

// XCD-aware remap of a linear workgroup id (hardware places workgroup i on XCD i % 8): give each
// XCD one contiguous run of the logical id space so neighbouring tiles share that XCD's L2.
// Bijective for any n.  Performance only -- nothing depends on the placement.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t i, uint32_t n)
{
    const uint32_t nx = 8, q = n / nx, r = n % nx, x = i % nx, j = i / nx;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

}  // namespace smvs
