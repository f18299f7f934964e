/*
 * oracle.c -- CPU restatement of the SatMVS RPC plane-sweep hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (satmvs_amd/) may link, import or
 * call this file.  Allowed users: tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg -- as the checker / reported CPU baseline, never as the thing shipped.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
 * against golden vectors produced by importing the Python reference in the build container
 * (tests/golden/gen_golden.py; the reference has no tests of its own -- SURVEY.md section 4 -- and its
 * two known-answer comments, tools/iccv_solver.py:42-64, are reproduced by
 * satmvs_amd/rpc_synth.py, which is host code, not part of this file).
 *
 * Arithmetic conventions (what "the reference computes" means at rounding level):
 *   - RPC math is float64, operation order as written in the reference, one rounding per
 *     torch op, no FMA contraction (compile with -ffp-contract=off).  The only freedom is
 *     the order inside torch.sum(coef*rpc, -1) (20 terms); we add left to right.
 *   - float32 sampling follows ATen's CPU grid_sampler_2d (vectorised AVX path of
 *     torch 2.10): unnormalise = fma(g+1, size/2, -0.5); weights from floor(); value =
 *     fma(se_v,se, fma(sw_v,sw, fma(ne_v,ne, nw_v*nw))).  Verified bit-exact against torch
 *     on random grids (tests/test_oracle_golden.py::test_grid_sample_bit_exact).
 *   - variance: sq/V - (sum/V)*(sum/V) with true float32 divisions, sums accumulated
 *     ref, src1, src2, ... (networks/casred.py:26-53).
 *
 * All reference citations are to /root/reference (read-only; not present on the GPU box).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ---- 170-vector layout: tools/RPCCore.py:8-28, dataset/data_io.py:78-92 ---------------- */
enum { I_LINE_OFF = 0, I_SAMP_OFF = 1, I_LAT_OFF = 2, I_LON_OFF = 3, I_H_OFF = 4,
       I_LINE_SCALE = 5, I_SAMP_SCALE = 6, I_LAT_SCALE = 7, I_LON_SCALE = 8, I_H_SCALE = 9,
       I_LNUM = 10, I_LDEN = 30, I_SNUM = 50, I_SDEN = 70,
       I_LATNUM = 90, I_LATDEN = 110, I_LONNUM = 130, I_LONDEN = 150, RPC_LEN = 170 };

/* modules/warping.py:183-207 -- the 20 monomials, built by the same product chain. */
static inline void plh_coef(double P, double L, double H, double c[20])
{
    c[0] = 1.0;      /* column 0 stays 1 from torch.ones (networks/casred.py:34) */
    c[1] = L;
    c[2] = P;
    c[3] = H;
    c[4] = L * P;
    c[5] = L * H;
    c[6] = P * H;
    c[7] = L * L;
    c[8] = P * P;
    c[9] = H * H;
    c[10] = P * c[5];
    c[11] = L * c[7];
    c[12] = L * c[8];
    c[13] = L * c[9];
    c[14] = L * c[4];
    c[15] = P * c[8];
    c[16] = P * c[9];
    c[17] = L * c[5];
    c[18] = P * c[6];
    c[19] = H * c[9];
}

static inline double dot20(const double c[20], const double *k)
{
    double s = c[0] * k[0];
    for (int i = 1; i < 20; ++i) s = s + c[i] * k[i];
    return s;
}

/* modules/warping.py:255-307 -- image (samp, line) + height -> ground (lat, lon). */
static inline void photo2obj_1(const double *rpc, double samp, double line, double hei,
                               double *lat, double *lon)
{
    double c[20];
    double s = (samp - rpc[I_SAMP_OFF]) / rpc[I_SAMP_SCALE];
    double l = (line - rpc[I_LINE_OFF]) / rpc[I_LINE_SCALE];
    double h = (hei - rpc[I_H_OFF]) / rpc[I_H_SCALE];
    plh_coef(s, l, h, c);                                  /* RPC_PLH_COEF(samp, line, hei) */
    double la = dot20(c, rpc + I_LATNUM) / dot20(c, rpc + I_LATDEN);
    double lo = dot20(c, rpc + I_LONNUM) / dot20(c, rpc + I_LONDEN);
    la = la * rpc[I_LAT_SCALE];
    la = la + rpc[I_LAT_OFF];
    lo = lo * rpc[I_LON_SCALE];
    lo = lo + rpc[I_LON_OFF];
    *lat = la;
    *lon = lo;
}

/* modules/warping.py:218-252 -- ground (lat, lon, h) -> image (samp, line). */
static inline void obj2photo_1(const double *rpc, double lat, double lon, double hei,
                               double *samp, double *line)
{
    double c[20];
    double la = (lat - rpc[I_LAT_OFF]) / rpc[I_LAT_SCALE];
    double lo = (lon - rpc[I_LON_OFF]) / rpc[I_LON_SCALE];
    double h = (hei - rpc[I_H_OFF]) / rpc[I_H_SCALE];
    plh_coef(la, lo, h, c);                                /* RPC_PLH_COEF(lat, lon, hei) */
    double s = dot20(c, rpc + I_SNUM) / dot20(c, rpc + I_SDEN);
    double l = dot20(c, rpc + I_LNUM) / dot20(c, rpc + I_LDEN);
    s = s * rpc[I_SAMP_SCALE];
    s = s + rpc[I_SAMP_OFF];
    l = l * rpc[I_LINE_SCALE];
    l = l + rpc[I_LINE_OFF];
    *samp = s;
    *line = l;
}

/* Batch projectors on flat arrays: the RPC_Photo2Obj / RPC_Obj2Photo helpers
 * (modules/warping.py:255,218) and the numpy twins tools/RPCCore.py:424-489.
 * dir 0: (samp,line,h) -> (lat,lon);  dir 1: (lat,lon,h) -> (samp,line). */
ORC_API void orc_rpc_project(const double *rpc170, const double *a, const double *b,
                             const double *h, double *o0, double *o1, size_t n, int dir)
{
    for (size_t i = 0; i < n; ++i) {
        if (dir == 0) photo2obj_1(rpc170, a[i], b[i], h[i], &o0[i], &o1[i]);
        else          obj2photo_1(rpc170, a[i], b[i], h[i], &o0[i], &o1[i]);
    }
}

/* ---- sampler: F.grid_sample(bilinear, zeros, align_corners=False) fed with a grid that was
 *      normalised with the align_corners=True formula (modules/warping.py:350-359).
 *      px/py are the float32 pixel coordinates (samp.float(), line.float()). ------------------ */
typedef struct { int x0, y0; float nw, ne, sw, se; int m_nw, m_ne, m_sw, m_se; float fw, fn; } tap_t;

static inline tap_t make_tap(float px, float py, int H, int W)
{
    tap_t t;
    /* modules/warping.py:350-351: float32 tensor / python scalar, then - 1 */
    float gx = px / (float)((W - 1) / 2.0) - 1.0f;
    float gy = py / (float)((H - 1) / 2.0) - 1.0f;
    /* ATen unnormalise, align_corners=False */
    float x = fmaf(gx + 1.0f, (float)W * 0.5f, -0.5f);
    float y = fmaf(gy + 1.0f, (float)H * 0.5f, -0.5f);
    float xw = floorf(x), yn = floorf(y);
    float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
    t.nw = s * e; t.ne = s * w; t.sw = n * e; t.se = n * w;
    t.fw = w; t.fn = n;
    /* bounds tested in float so NaN / huge coordinates fall out as "outside" */
    int xin0 = (xw >= 0.0f) && (xw <= (float)(W - 1));
    int xin1 = (xw >= -1.0f) && (xw <= (float)(W - 2));
    int yin0 = (yn >= 0.0f) && (yn <= (float)(H - 1));
    int yin1 = (yn >= -1.0f) && (yn <= (float)(H - 2));
    t.x0 = (xin0 || xin1) ? (int)xw : 0;
    t.y0 = (yin0 || yin1) ? (int)yn : 0;
    t.m_nw = xin0 && yin0; t.m_ne = xin1 && yin0; t.m_sw = xin0 && yin1; t.m_se = xin1 && yin1;
    return t;
}

static inline float tap_fetch(const float *plane, const tap_t *t, int W)
{
    const float *p = plane + (ptrdiff_t)t->y0 * W + t->x0;
    float a = t->m_nw ? p[0] : 0.0f;
    float b = t->m_ne ? p[1] : 0.0f;
    float c = t->m_sw ? p[W] : 0.0f;
    float d = t->m_se ? p[W + 1] : 0.0f;
    float r = a * t->nw;
    r = fmaf(b, t->ne, r);
    r = fmaf(c, t->sw, r);
    r = fmaf(d, t->se, r);
    return r;
}

/* Plain grid_sample on an explicit normalised grid (B,Ho,Wo,2): used only to pin the sampler
 * against torch bit-for-bit. */
ORC_API void orc_grid_sample(const float *inp, const float *grid, float *out,
                             int B, int C, int H, int W, int Ho, int Wo)
{
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < Ho * Wo; ++i) {
            float gx = grid[((size_t)b * Ho * Wo + i) * 2 + 0];
            float gy = grid[((size_t)b * Ho * Wo + i) * 2 + 1];
            tap_t t;
            float x = fmaf(gx + 1.0f, (float)W * 0.5f, -0.5f);
            float y = fmaf(gy + 1.0f, (float)H * 0.5f, -0.5f);
            float xw = floorf(x), yn = floorf(y);
            float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
            t.nw = s * e; t.ne = s * w; t.sw = n * e; t.se = n * w;
            int xin0 = (xw >= 0.0f) && (xw <= (float)(W - 1));
            int xin1 = (xw >= -1.0f) && (xw <= (float)(W - 2));
            int yin0 = (yn >= 0.0f) && (yn <= (float)(H - 1));
            int yin1 = (yn >= -1.0f) && (yn <= (float)(H - 2));
            t.x0 = (xin0 || xin1) ? (int)xw : 0;
            t.y0 = (yin0 || yin1) ? (int)yn : 0;
            t.m_nw = xin0 && yin0; t.m_ne = xin1 && yin0; t.m_sw = xin0 && yin1; t.m_se = xin1 && yin1;
            for (int c = 0; c < C; ++c)
                out[((size_t)b * C + c) * Ho * Wo + i] =
                    tap_fetch(inp + ((size_t)b * C + c) * H * W, &t, W);
        }
}

static inline double height_at(const float *depth, int depth_is_4d, int b, int d, int y, int x,
                               int D, int H, int W)
{
    /* modules/warping.py:329-337: (B,D) is broadcast over the plane; both are cast to double */
    if (depth_is_4d) return (double)depth[(((size_t)b * D + d) * H + y) * W + x];
    return (double)depth[(size_t)b * D + d];
}

/* Source-image pixel position of ref pixel (x,y) on height plane h -- warping.py:340-348. */
static inline void rpc_chain(const double *ref_rpc, const double *src_rpc, int x, int y, double h,
                             float *px, float *py)
{
    double lat, lon, samp, line;
    photo2obj_1(ref_rpc, (double)x, (double)y, h, &lat, &lon);
    obj2photo_1(src_rpc, lat, lon, h, &samp, &line);
    *px = (float)samp;
    *py = (float)line;
}

/* modules/warping.py:310-365 rpc_warping: src_fea (B,C,H,W) -> warped (B,C,D,H,W).
 * src_rpc/ref_rpc are (B,170). */
ORC_API void orc_rpc_warping(const float *src_fea, const double *src_rpc, const double *ref_rpc,
                             const float *depth, int depth_is_4d, float *out,
                             int B, int C, int D, int H, int W)
{
    size_t HW = (size_t)H * W;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float px, py;
                    double h = height_at(depth, depth_is_4d, b, d, y, x, D, H, W);
                    rpc_chain(ref_rpc + (size_t)b * RPC_LEN, src_rpc + (size_t)b * RPC_LEN, x, y, h, &px, &py);
                    tap_t t = make_tap(px, py, H, W);
                    for (int c = 0; c < C; ++c)
                        out[((((size_t)b * C + c) * D + d) * H + y) * W + x] =
                            tap_fetch(src_fea + ((size_t)b * C + c) * HW, &t, W);
                }
    }
}

/* Projection-only variant: the float32 pixel coordinates fed to the sampler and the float64
 * (lat,lon)/(samp,line) intermediates, for pinning against the reference's tensors. */
ORC_API void orc_rpc_warp_coords(const double *src_rpc, const double *ref_rpc, const float *depth,
                                 int depth_is_4d, double *lat, double *lon, double *samp, double *line,
                                 int B, int D, int H, int W)
{
    for (int b = 0; b < B; ++b)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    size_t i = (((size_t)b * D + d) * H + y) * W + x;
                    double h = height_at(depth, depth_is_4d, b, d, y, x, D, H, W);
                    photo2obj_1(ref_rpc + (size_t)b * RPC_LEN, (double)x, (double)y, h, &lat[i], &lon[i]);
                    obj2photo_1(src_rpc + (size_t)b * RPC_LEN, lat[i], lon[i], h, &samp[i], &line[i]);
                }
}

/* ---- homography path: modules/warping.py:6-44 ------------------------------------------- */
/* proj = src_proj @ inverse(ref_proj) (warping.py:19): 4x4 float64, Gauss-Jordan with partial
 * pivoting (torch.inverse is LAPACK getrf/getri; equal up to round-off). */
ORC_API int orc_homo_compose(const double *src_proj, const double *ref_proj, double *out, int B)
{
    for (int b = 0; b < B; ++b) {
        double a[4][8];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                a[i][j] = ref_proj[b * 16 + i * 4 + j];
                a[i][4 + j] = (i == j) ? 1.0 : 0.0;
            }
        for (int col = 0; col < 4; ++col) {
            int piv = col;
            for (int r = col + 1; r < 4; ++r)
                if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
            if (a[piv][col] == 0.0) return 1;
            if (piv != col)
                for (int j = 0; j < 8; ++j) { double t = a[col][j]; a[col][j] = a[piv][j]; a[piv][j] = t; }
            double inv = 1.0 / a[col][col];
            for (int j = 0; j < 8; ++j) a[col][j] = a[col][j] * inv;
            for (int r = 0; r < 4; ++r) {
                if (r == col) continue;
                double f = a[r][col];
                for (int j = 0; j < 8; ++j) a[r][j] = a[r][j] - f * a[col][j];
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double s = 0.0;
                for (int k = 0; k < 4; ++k) s = s + src_proj[b * 16 + i * 4 + k] * a[k][4 + j];
                out[b * 16 + i * 4 + j] = s;
            }
    }
    return 0;
}

static inline void homo_chain(const double *P /* composed 4x4 */, int x, int y, double depth,
                              int H, int W, float *gx, float *gy)
{
    /* warping.py:28-38: rot @ (x,y,1) * depth + trans, perspective divide and normalisation all in
     * float64, cast to float32 at the end. */
    double fx = (double)(float)x, fy = (double)(float)y;
    double rx = P[0] * fx + P[1] * fy + P[2];
    double ry = P[4] * fx + P[5] * fy + P[6];
    double rz = P[8] * fx + P[9] * fy + P[10];
    double X = rx * depth + P[3];
    double Y = ry * depth + P[7];
    double Z = rz * depth + P[11];
    double u = X / Z, v = Y / Z;
    *gx = (float)(u / ((W - 1) / 2.0) - 1.0);
    *gy = (float)(v / ((H - 1) / 2.0) - 1.0);
}

static inline tap_t make_tap_norm(float gx, float gy, int H, int W)
{
    tap_t t;
    float x = fmaf(gx + 1.0f, (float)W * 0.5f, -0.5f);
    float y = fmaf(gy + 1.0f, (float)H * 0.5f, -0.5f);
    float xw = floorf(x), yn = floorf(y);
    float w = x - xw, e = 1.0f - w, n = y - yn, s = 1.0f - n;
    t.nw = s * e; t.ne = s * w; t.sw = n * e; t.se = n * w;
    t.fw = w; t.fn = n;
    int xin0 = (xw >= 0.0f) && (xw <= (float)(W - 1));
    int xin1 = (xw >= -1.0f) && (xw <= (float)(W - 2));
    int yin0 = (yn >= 0.0f) && (yn <= (float)(H - 1));
    int yin1 = (yn >= -1.0f) && (yn <= (float)(H - 2));
    t.x0 = (xin0 || xin1) ? (int)xw : 0;
    t.y0 = (yin0 || yin1) ? (int)yn : 0;
    t.m_nw = xin0 && yin0; t.m_ne = xin1 && yin0; t.m_sw = xin0 && yin1; t.m_se = xin1 && yin1;
    return t;
}

/* homo_warping with the composed matrix `proj` (B,4,4) = src_proj @ inv(ref_proj). */
ORC_API void orc_homo_warping(const float *src_fea, const double *proj, const float *depth,
                              int depth_is_4d, float *out, int B, int C, int D, int H, int W)
{
    size_t HW = (size_t)H * W;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float gx, gy;
                    double h = height_at(depth, depth_is_4d, b, d, y, x, D, H, W);
                    homo_chain(proj + (size_t)b * 16, x, y, h, H, W, &gx, &gy);
                    tap_t t = make_tap_norm(gx, gy, H, W);
                    for (int c = 0; c < C; ++c)
                        out[((((size_t)b * C + c) * D + d) * H + y) * W + x] =
                            tap_fetch(src_fea + ((size_t)b * C + c) * HW, &t, W);
                }
    }
}

/* ---- variance cost volume: networks/casred.py:22-53 (train), :191-212 (pred, one plane) ---
 * geo 0: rpc (B,V,170) float64, view 0 = reference (torch.unbind(proj_matrices,1), casred.py:13)
 * geo 1: proj (B,V-1,16) float64 already composed per source view.
 * feats: V pointers to (B,C,H,W); planes [d_begin,d_end) are written into out (B,C,D,H,W). */
ORC_API void orc_costvol_variance(const float *const *feats, const double *geo_params, int geo,
                                  const float *depth, int depth_is_4d, float *out,
                                  int B, int V, int C, int D, int H, int W, int d_begin, int d_end)
{
    size_t HW = (size_t)H * W;
    float fV = (float)V;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int d = d_begin; d < d_end; ++d)
            for (int y = 0; y < H; ++y) {
                tap_t taps[16];
                for (int x = 0; x < W; ++x) {
                    double h = height_at(depth, depth_is_4d, b, d, y, x, D, H, W);
                    for (int v = 1; v < V; ++v) {
                        if (geo == 0) {
                            float px, py;
                            const double *r = geo_params + (size_t)b * V * RPC_LEN;
                            rpc_chain(r, r + (size_t)v * RPC_LEN, x, y, h, &px, &py);
                            taps[v] = make_tap(px, py, H, W);
                        } else {
                            float gx, gy;
                            homo_chain(geo_params + ((size_t)b * (V - 1) + (v - 1)) * 16, x, y, h, H, W, &gx, &gy);
                            taps[v] = make_tap_norm(gx, gy, H, W);
                        }
                    }
                    for (int c = 0; c < C; ++c) {
                        float r = feats[0][((size_t)b * C + c) * HW + (size_t)y * W + x];
                        float sum = r;
                        float sq = r * r;                       /* ref_volume ** 2 */
                        for (int v = 1; v < V; ++v) {
                            float wv = tap_fetch(feats[v] + ((size_t)b * C + c) * HW, &taps[v], W);
                            sum = sum + wv;                     /* volume_sum + warped */
                            sq = sq + wv * wv;                  /* volume_sq_sum + warped ** 2 */
                        }
                        float m = sum / fV;                     /* div_(num_views) */
                        float q = sq / fV;
                        out[((((size_t)b * C + c) * D + d) * H + y) * W + x] = q - m * m;
                    }
                }
            }
    }
}

/* The same volume evaluated in float64 from the reference's float32 tap positions: the bilinear weights are the exact
 * products of the float32 fractions, the warped values, their sum / sum of squares and the variance are float64.  This is
 * the real-number function that BOTH the reference's float32 sequence and the library's fused arithmetic
 * (SMVS_ARITH_FUSED) approximate; tests/test_fused_arith.py measures each against it.  `scale` receives sum(X^2)/V with
 * X = |ref| resp. sum |corner| * weight of a source: the magnitude the float32 rounding errors of either sequence are
 * proportional to (a bilinear sample of large corners can be small).  Not a reference function. */
ORC_API void orc_costvol_variance_f64(const float *const *feats, const double *geo_params, int geo,
                                      const float *depth, int depth_is_4d, double *out, double *scale,
                                      int B, int V, int C, int D, int H, int W)
{
    size_t HW = (size_t)H * W;
    for (int b = 0; b < B; ++b) {
#pragma omp parallel for collapse(2) schedule(static)
        for (int d = 0; d < D; ++d)
            for (int y = 0; y < H; ++y) {
                tap_t taps[16];
                for (int x = 0; x < W; ++x) {
                    double h = height_at(depth, depth_is_4d, b, d, y, x, D, H, W);
                    for (int v = 1; v < V; ++v) {
                        if (geo == 0) {
                            float px, py;
                            const double *r = geo_params + (size_t)b * V * RPC_LEN;
                            rpc_chain(r, r + (size_t)v * RPC_LEN, x, y, h, &px, &py);
                            taps[v] = make_tap(px, py, H, W);
                        } else {
                            float gx, gy;
                            homo_chain(geo_params + ((size_t)b * (V - 1) + (v - 1)) * 16, x, y, h, H, W, &gx, &gy);
                            taps[v] = make_tap_norm(gx, gy, H, W);
                        }
                    }
                    for (int c = 0; c < C; ++c) {
                        double r = (double)feats[0][((size_t)b * C + c) * HW + (size_t)y * W + x];
                        double sum = r, sq = r * r, mag = r * r;
                        for (int v = 1; v < V; ++v) {
                            const tap_t *t = &taps[v];
                            const float *p = feats[v] + ((size_t)b * C + c) * HW + (ptrdiff_t)t->y0 * W + t->x0;
                            double fw = (double)t->fw, fn = (double)t->fn;
                            double wv = (t->m_nw ? (double)p[0] : 0.0) * ((1.0 - fn) * (1.0 - fw))
                                      + (t->m_ne ? (double)p[1] : 0.0) * ((1.0 - fn) * fw)
                                      + (t->m_sw ? (double)p[W] : 0.0) * (fn * (1.0 - fw))
                                      + (t->m_se ? (double)p[W + 1] : 0.0) * (fn * fw);
                            double av = (t->m_nw ? fabs((double)p[0]) : 0.0) * ((1.0 - fn) * (1.0 - fw))
                                      + (t->m_ne ? fabs((double)p[1]) : 0.0) * ((1.0 - fn) * fw)
                                      + (t->m_sw ? fabs((double)p[W]) : 0.0) * (fn * (1.0 - fw))
                                      + (t->m_se ? fabs((double)p[W + 1]) : 0.0) * (fn * fw);
                            sum += wv;
                            sq += wv * wv;
                            mag += av * av;
                        }
                        size_t o = ((((size_t)b * C + c) * D + d) * H + y) * W + x;
                        double m = sum / V;
                        out[o] = sq / V - m * m;
                        scale[o] = mag / V;
                    }
                }
            }
    }
}

/* ---- regression ---------------------------------------------------------------------------
 * Train path (networks/casred.py:58-62, modules/module.py:433-439): softmax over D, expected
 * height, max probability.  torch's CPU softmax: x - max, exp, sum, divide (float32). */
ORC_API void orc_softmax_regress(const float *reg, const float *depth, int depth_is_4d,
                                 float *out_depth, float *out_conf, int B, int D, int H, int W)
{
    size_t HW = (size_t)H * W;
    for (int b = 0; b < B; ++b)
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < HW; ++i) {
            const float *r = reg + (size_t)b * D * HW + i;
            float mx = r[0];
            for (int d = 1; d < D; ++d) mx = fmaxf(mx, r[d * HW]);
            float den = 0.0f;
            for (int d = 0; d < D; ++d) den = den + expf(r[d * HW] - mx);
            float acc = 0.0f, best = 0.0f;
            for (int d = 0; d < D; ++d) {
                float p = expf(r[d * HW] - mx) / den;
                float hv = depth_is_4d ? depth[((size_t)b * D + d) * HW + i] : depth[(size_t)b * D + d];
                acc = acc + p * hv;
                best = (d == 0 || p > best) ? p : best;
            }
            out_depth[(size_t)b * HW + i] = acc;
            out_conf[(size_t)b * HW + i] = best;
        }
}

/* Height hypotheses of cascade stages 2 and 3 (networks/casred.py:134-145 + modules/depth_range.py:4-20):
 *   cur = bilinear resize (align_corners=False) of the previous height map (B,hp,wp) to the image size (ih,iw);
 *   cur_min = cur - c, cur_max = cur + c with c = (float)(ndepth / 2 * interval); step = (cur_max - cur_min) / (ndepth - 1);
 *   samples[d] = cur_min + d * step at image resolution; trilinear resize (align_corners=False) to (D, ih/scale, iw/scale).
 * ATen's CPU kernels evaluate every two-term interpolation as fma(w0, v0, w1 * v1) (checked bit for bit against the
 * reference, tests/test_oracle_golden.py::test_height_hypotheses); the plane axis keeps its size (weights 1 and 0). */
static float orc_lerp2(float w0, float v0, float w1, float v1) { return fmaf(w0, v0, w1 * v1); }

static void orc_hyp_pixel(const float *prev, int hp, int wp, int ih, int iw, float c, float ndm1, int Y, int X,
                          float *cmin, float *step)
{
    float sh = (float)hp / (float)ih, sw = (float)wp / (float)iw;
    float sy = fmaxf(sh * ((float)Y + 0.5f) - 0.5f, 0.0f), sx = fmaxf(sw * ((float)X + 0.5f) - 0.5f, 0.0f);
    int y0 = (int)sy, x0 = (int)sx;
    int y1 = y0 + 1 < hp ? y0 + 1 : hp - 1, x1 = x0 + 1 < wp ? x0 + 1 : wp - 1;
    float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    float top = orc_lerp2(lx0, prev[y0 * wp + x0], lx1, prev[y0 * wp + x1]);
    float bot = orc_lerp2(lx0, prev[y1 * wp + x0], lx1, prev[y1 * wp + x1]);
    float cur = orc_lerp2(ly0, top, ly1, bot);
    *cmin = cur - c;
    *step = ((cur + c) - *cmin) / ndm1;
}

ORC_API int orc_height_hypotheses(const float *prev, int B, int hp, int wp, int ih, int iw, int ndepth, double interval,
                                  int H, int W, float *out)
{
    if (ih % H || iw % W || ih / H != iw / W) return 1;
    int scale = ih / H;
    if (scale != 1 && scale != 2) return 1;
    float c = (float)(ndepth / 2.0 * interval), ndm1 = (float)(ndepth - 1);
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float cm[4], st[4];
                int n = scale == 1 ? 1 : 4;
                for (int k = 0; k < n; ++k)
                    orc_hyp_pixel(prev + (size_t)b * hp * wp, hp, wp, ih, iw, c, ndm1,
                                  scale == 1 ? y : 2 * y + (k >> 1), scale == 1 ? x : 2 * x + (k & 1), &cm[k], &st[k]);
                for (int d = 0; d < ndepth; ++d) {
                    float fd = (float)d, v;
                    if (scale == 1) v = cm[0] + fd * st[0];
                    else {
                        float a = cm[0] + fd * st[0], bb = cm[1] + fd * st[1], cc = cm[2] + fd * st[2], dd = cm[3] + fd * st[3];
                        v = orc_lerp2(0.5f, orc_lerp2(0.5f, a, 0.5f, bb), 0.5f, orc_lerp2(0.5f, cc, 0.5f, dd));
                    }
                    out[(((size_t)b * ndepth + d) * H + y) * W + x] = v;
                }
            }
    return 0;
}

/* UCS-Net hypotheses of stages 2 and 3 (networks/ucs.py:49-58 + modules/depth_range.py:45-86, uncertainty_aware_samples):
 *   cur = bilinear resize (align_corners=False) of the previous stage's height map (B,hp,wp) to this stage's (H,W), ev = the same
 *   of its standard-deviation map; low = cur - ev, high = cur + ev, each replaced by the batch item's range bound where it
 *   crosses it (masked assignment: (low - min) < 0, (high - max) > 0); step = (high - low) / (float)(ndepth - 1);
 *   samples[d] = low + step * d + 1e-12, all float32, at the stage resolution (no further resize). */
static float orc_bilinear(const float *q, int hp, int wp, int H, int W, int y, int x)
{
    float sh = (float)hp / (float)H, sw = (float)wp / (float)W;
    float sy = fmaxf(sh * ((float)y + 0.5f) - 0.5f, 0.0f), sx = fmaxf(sw * ((float)x + 0.5f) - 0.5f, 0.0f);
    int y0 = (int)sy, x0 = (int)sx;
    int y1 = y0 + 1 < hp ? y0 + 1 : hp - 1, x1 = x0 + 1 < wp ? x0 + 1 : wp - 1;
    float ly1 = sy - (float)y0, lx1 = sx - (float)x0, ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    float top = orc_lerp2(lx0, q[y0 * wp + x0], lx1, q[y0 * wp + x1]);
    float bot = orc_lerp2(lx0, q[y1 * wp + x0], lx1, q[y1 * wp + x1]);
    return orc_lerp2(ly0, top, ly1, bot);
}

ORC_API int orc_ucs_hypotheses(const float *prev, const float *prev_var, const float *rmin, const float *rmax,
                               int B, int hp, int wp, int ndepth, int H, int W, float *out)
{
    if (ndepth < 2) return 1;
    float ndm1 = (float)ndepth - 1.0f;
    for (int b = 0; b < B; ++b)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float cur = orc_bilinear(prev + (size_t)b * hp * wp, hp, wp, H, W, y, x);
                float ev = orc_bilinear(prev_var + (size_t)b * hp * wp, hp, wp, H, W, y, x);
                float low = cur - ev, high = cur + ev;
                if (low - rmin[b] < 0.0f) low = rmin[b];
                if (high - rmax[b] > 0.0f) high = rmax[b];
                float step = (high - low) / ndm1;
                for (int d = 0; d < ndepth; ++d) {
                    float v = low + step * (float)d;
                    out[(((size_t)b * ndepth + d) * H + y) * W + x] = v + 1e-12f;
                }
            }
    return 0;
}

/* CascadeMVSNet / UCSNet regression (networks/casmvs.py:66-74, networks/ucs.py:60-74): softmax over D, expected
 * height, photometric confidence = probability mass of the four hypotheses [idx-1, idx+2] around
 * idx = clamp(trunc(E[index]), 0, D-1) -- F.pad(prob,(0,0,0,0,1,2)) + 4*avg_pool3d((4,1,1)) + gather -- and UCSNet's
 * variance = lamb * sqrt(sum prob * (h - depth)^2).  float32, term by term like the torch composite. */
ORC_API void orc_window_regress(const float *reg, const float *depth, int depth_is_4d,
                                float *out_depth, float *out_conf, float *out_var, float lamb,
                                int B, int D, int H, int W)
{
    size_t HW = (size_t)H * W;
    for (int b = 0; b < B; ++b)
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < HW; ++i) {
            const float *r = reg + (size_t)b * D * HW + i;
            float mx = r[0];
            for (int d = 1; d < D; ++d) mx = fmaxf(mx, r[d * HW]);
            float den = 0.0f;
            for (int d = 0; d < D; ++d) den = den + expf(r[d * HW] - mx);
            float acc = 0.0f, fidx = 0.0f;
            for (int d = 0; d < D; ++d) {
                float p = expf(r[d * HW] - mx) / den;
                float hv = depth_is_4d ? depth[((size_t)b * D + d) * HW + i] : depth[(size_t)b * D + d];
                acc = acc + p * hv;
                fidx = fidx + p * (float)d;
            }
            int idx = (int)fidx;
            idx = idx < 0 ? 0 : (idx > D - 1 ? D - 1 : idx);
            float conf = 0.0f;
            for (int k = -1; k <= 2; ++k) {
                int d = idx + k;
                conf = conf + ((d >= 0 && d < D) ? expf(r[d * HW] - mx) / den : 0.0f);
            }
            out_depth[(size_t)b * HW + i] = acc;
            out_conf[(size_t)b * HW + i] = conf;
            if (out_var) {
                float v = 0.0f;
                for (int d = 0; d < D; ++d) {
                    float p = expf(r[d * HW] - mx) / den;
                    float hv = depth_is_4d ? depth[((size_t)b * D + d) * HW + i] : depth[(size_t)b * D + d];
                    float dh = hv - acc;
                    v = v + (dh * dh) * p;
                }
                out_var[(size_t)b * HW + i] = lamb * sqrtf(v);
            }
        }
}

/* Pred path, one plane (networks/casred.py:218-231): prob = exp(double(reg)) with no
 * max-subtraction; running max, sum of h*prob, sum of prob -- all float64. */
ORC_API void orc_stream_regress_step(const float *reg_plane, const float *depth_plane, int depth_is_plane,
                                     double *exp_sum, double *depth_img, double *max_prob,
                                     int B, int H, int W, int D, int d)
{
    size_t HW = (size_t)H * W;
    for (int b = 0; b < B; ++b)
        for (size_t i = 0; i < HW; ++i) {
            double p = exp((double)reg_plane[(size_t)b * HW + i]);
            double hv = depth_is_plane ? (double)depth_plane[((size_t)b * D + d) * HW + i]
                                       : (double)depth_plane[(size_t)b * D + d];
            size_t o = (size_t)b * HW + i;
            double flag = (max_prob[o] < p) ? 1.0 : 0.0;         /* update_flag_image */
            max_prob[o] = flag * p + (1.0 - flag) * max_prob[o];
            depth_img[o] = hv * p + depth_img[o];
            exp_sum[o] = exp_sum[o] + p;
        }
}

/* networks/casred.py:234-236 */
ORC_API void orc_stream_regress_final(const double *exp_sum, const double *depth_img, const double *max_prob,
                                      float *out_depth, float *out_conf, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        double den = exp_sum[i] + 1e-10;
        out_depth[i] = (float)(depth_img[i] / den);
        out_conf[i] = (float)(max_prob[i] / den);
    }
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ===========================================================================================
 * RED regulariser primitives (modules/module.py:6-58 ConvGRUCell2, :595-693 RED / slice_RED):
 * 3x3 convolutions (stride 1/2, pad 1), 3x3 transposed convolutions (stride 1 or 2, pad 1,
 * output_padding 0/1), GroupNorm(1, C).  torch's float32 accumulation order inside its conv
 * kernels is unspecified; the oracle accumulates in double and rounds once, so it sits within
 * float32 round-off of every correct implementation (tolerance stated in the tests: 1e-5).
 * =========================================================================================== */
ORC_API void orc_conv2d3x3(const float *in, const float *w, const float *bias, float *out,
                           int B, int Cin, int Cout, int H, int W, int stride)
{
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double acc = bias ? (double)bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < 3; ++ky) {
                            const int iy = oy * stride - 1 + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < 3; ++kx) {
                                const int ix = ox * stride - 1 + kx;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)in[(((size_t)b * Cin + ci) * H + iy) * W + ix] *
                                       (double)w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
                            }
                        }
                    out[(((size_t)b * Cout + co) * Ho + oy) * Wo + ox] = (float)acc;
                }
}

/* nn.Conv2d(k=K, stride, padding=K/2) for the feature extractor (modules/module.py:442-543 uses K = 1, 3, 5). */
ORC_API void orc_conv2d_k(const float *in, const float *w, const float *bias, float *out,
                          int B, int Cin, int Cout, int H, int W, int K, int stride)
{
    const int pad = K / 2;
    const int Ho = (H + 2 * pad - K) / stride + 1, Wo = (W + 2 * pad - K) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double acc = bias ? (double)bias[co] : 0.0;
                    for (int ci = 0; ci < Cin; ++ci)
                        for (int ky = 0; ky < K; ++ky) {
                            const int iy = oy * stride - pad + ky;
                            if (iy < 0 || iy >= H) continue;
                            for (int kx = 0; kx < K; ++kx) {
                                const int ix = ox * stride - pad + kx;
                                if (ix < 0 || ix >= W) continue;
                                acc += (double)in[(((size_t)b * Cin + ci) * H + iy) * W + ix] *
                                       (double)w[(((size_t)co * Cin + ci) * K + ky) * K + kx];
                            }
                        }
                    out[(((size_t)b * Cout + co) * Ho + oy) * Wo + ox] = (float)acc;
                }
}

/* nn.ConvTranspose2d(k=3, pad=1): weight (Cin, Cout, 3, 3); out size (H-1)*stride - 2 + 3 + out_pad */
ORC_API void orc_convT2d3x3(const float *in, const float *w, const float *bias, float *out,
                            int B, int Cin, int Cout, int H, int W, int stride, int out_pad)
{
    const int Ho = (H - 1) * stride - 2 + 3 + out_pad, Wo = (W - 1) * stride - 2 + 3 + out_pad;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int oy = 0; oy < Ho; ++oy)
                for (int ox = 0; ox < Wo; ++ox) {
                    double acc = bias ? (double)bias[co] : 0.0;
                    for (int ky = 0; ky < 3; ++ky) {
                        const int ty = oy + 1 - ky;                 /* = iy * stride */
                        if (ty < 0 || ty % stride) continue;
                        const int iy = ty / stride;
                        if (iy >= H) continue;
                        for (int kx = 0; kx < 3; ++kx) {
                            const int tx = ox + 1 - kx;
                            if (tx < 0 || tx % stride) continue;
                            const int ix = tx / stride;
                            if (ix >= W) continue;
                            for (int ci = 0; ci < Cin; ++ci)
                                acc += (double)in[(((size_t)b * Cin + ci) * H + iy) * W + ix] *
                                       (double)w[(((size_t)ci * Cout + co) * 3 + ky) * 3 + kx];
                        }
                    }
                    out[(((size_t)b * Cout + co) * Ho + oy) * Wo + ox] = (float)acc;
                }
}

/* nn.GroupNorm(1, C, eps, affine): statistics over (C, H*W) per sample, biased variance. */
ORC_API void orc_groupnorm1(float *x, const float *gamma, const float *beta, float eps, int B, int C, int HW)
{
    for (int b = 0; b < B; ++b) {
        float *p = x + (size_t)b * C * HW;
        double s = 0.0, q = 0.0;
        const size_t n = (size_t)C * HW;
        for (size_t i = 0; i < n; ++i) { s += p[i]; q += (double)p[i] * p[i]; }
        const double mean = s / (double)n;
        double var = q / (double)n - mean * mean;
        if (var < 0.0) var = 0.0;
        const double rstd = 1.0 / sqrt(var + (double)eps);
        for (int c = 0; c < C; ++c)
            for (int i = 0; i < HW; ++i) {
                const size_t k = (size_t)c * HW + i;
                p[k] = (float)(((double)p[k] - mean) * rstd * (double)gamma[c] + (double)beta[c]);
            }
    }
}

/* ===========================================================================================
 * CostRegNet primitives (modules/module.py:324-410 Conv3d/Deconv3d, :546-577 CostRegNet):
 * 3x3x3 convolutions (stride 1/2, pad 1) and transposed convolutions (stride 2, pad 1,
 * output_padding 1), accumulated in double (see the RED note above).
 * =========================================================================================== */
ORC_API void orc_conv3d3(const float *in, const float *w, float *out,
                         int B, int Cin, int Cout, int D, int H, int W, int stride)
{
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int od = 0; od < Do; ++od)
                for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Wo; ++ox) {
                        double acc = 0.0;
                        for (int ci = 0; ci < Cin; ++ci)
                            for (int kd = 0; kd < 3; ++kd) {
                                const int id = od * stride - 1 + kd;
                                if (id < 0 || id >= D) continue;
                                for (int ky = 0; ky < 3; ++ky) {
                                    const int iy = oy * stride - 1 + ky;
                                    if (iy < 0 || iy >= H) continue;
                                    for (int kx = 0; kx < 3; ++kx) {
                                        const int ix = ox * stride - 1 + kx;
                                        if (ix < 0 || ix >= W) continue;
                                        acc += (double)in[((((size_t)b * Cin + ci) * D + id) * H + iy) * W + ix] *
                                               (double)w[((((size_t)co * Cin + ci) * 3 + kd) * 3 + ky) * 3 + kx];
                                    }
                                }
                            }
                        out[((((size_t)b * Cout + co) * Do + od) * Ho + oy) * Wo + ox] = (float)acc;
                    }
}

/* nn.ConvTranspose3d(k=3, stride=2, pad=1, output_padding=1): weight (Cin,Cout,3,3,3), out = 2x */
ORC_API void orc_convT3d3s2(const float *in, const float *w, float *out,
                            int B, int Cin, int Cout, int D, int H, int W)
{
    const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
#pragma omp parallel for collapse(3) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int od = 0; od < Do; ++od)
                for (int oy = 0; oy < Ho; ++oy)
                    for (int ox = 0; ox < Wo; ++ox) {
                        double acc = 0.0;
                        for (int kd = 0; kd < 3; ++kd) {
                            const int td = od + 1 - kd;
                            if (td < 0 || (td & 1) || td / 2 >= D) continue;
                            for (int ky = 0; ky < 3; ++ky) {
                                const int ty = oy + 1 - ky;
                                if (ty < 0 || (ty & 1) || ty / 2 >= H) continue;
                                for (int kx = 0; kx < 3; ++kx) {
                                    const int tx = ox + 1 - kx;
                                    if (tx < 0 || (tx & 1) || tx / 2 >= W) continue;
                                    for (int ci = 0; ci < Cin; ++ci)
                                        acc += (double)in[((((size_t)b * Cin + ci) * D + td / 2) * H + ty / 2) * W + tx / 2] *
                                               (double)w[((((size_t)ci * Cout + co) * 3 + kd) * 3 + ky) * 3 + kx];
                                }
                            }
                        }
                        out[((((size_t)b * Cout + co) * Do + od) * Ho + oy) * Wo + ox] = (float)acc;
                    }
}
