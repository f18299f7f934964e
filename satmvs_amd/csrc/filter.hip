// filter.hip -- geometric-consistency check between a reference and a source height map.
//
// One launch replaces reproject_with_depth + the per-pair part of check_geometric_consistency
// (/root/reference/tools/rpc_filter.py:11-70): per reference pixel (x, y, h)
//   1. image -> ground through the reference view's inverse RPC, ground -> source image (float64);
//   2. the source height map sampled there: cv2.remap(depth_src, float32 coordinates, INTER_LINEAR, BORDER_CONSTANT -999);
//   3. that source pixel + sampled height -> ground through the source view's inverse RPC -> back into the reference image;
//   4. mask = |reprojected - pixel| < p_ratio  and  |sampled height - reference height| < d_ratio; heights outside the mask -> 0.
// Pinned by tests/golden/filter.npz = the reference's own rpc_filter.py + rpc_tensor.py run as they are -- except
// cv2.remap, a third-party step (opencv-python 4.5.5.62, environment.yml:156; cv2 is absent from this image, so this
// step is restated from OpenCV's published algorithm and NOT pinned by a reference run): coordinates are rounded to 1/32
// pixel (cvRound(x * 32): round half to even), the four bilinear weights come from the 5-bit fractions, taps outside the
// image take the border value.
#include "smvs_device.h"
#include "smvs_host.h"

namespace smvs {

__device__ __forceinline__ float remap_linear_const(const float* __restrict__ img, int H, int W, float fx, float fy, float border)
{
    // cv::remap, INTER_LINEAR on CV_32F: fixed-point coordinates with INTER_BITS = 5
    const int sx = __float2int_rn(fx * 32.0f), sy = __float2int_rn(fy * 32.0f);
    const int ix = sx >> 5, iy = sy >> 5;
    const float ax = (float)(sx & 31) * (1.0f / 32.0f), ay = (float)(sy & 31) * (1.0f / 32.0f);
    auto at = [&](int y, int x) { return (x >= 0 && x < W && y >= 0 && y < H) ? img[(size_t)y * W + x] : border; };
    const float w00 = (1.0f - ax) * (1.0f - ay), w01 = ax * (1.0f - ay), w10 = (1.0f - ax) * ay, w11 = ax * ay;
    return at(iy, ix) * w00 + at(iy, ix + 1) * w01 + at(iy + 1, ix) * w10 + at(iy + 1, ix + 1) * w11;
}

__global__ __launch_bounds__(256)
void geo_consistency_kernel(const float* __restrict__ depth_ref, const double* __restrict__ rpc_ref,
                            const float* __restrict__ depth_src, const double* __restrict__ rpc_src,
                            int H, int W, int Hs, int Ws, double p_ratio, double d_ratio,
                            unsigned char* __restrict__ mask, float* __restrict__ depth_reproj,
                            double* __restrict__ x_src, double* __restrict__ y_src,
                            double* __restrict__ x_back, double* __restrict__ y_back)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const cgeo_t rr = as_cgeo(rpc_ref), rs = as_cgeo(rpc_src);
    const double h = (double)depth_ref[i];
    double lat, lon, xs, ys;
    rpc_photo2obj(rr, rpc_inv_image(rr), (double)x, (double)y, h, lat, lon);
    rpc_obj2photo(rs, rpc_inv_ground(rs), lat, lon, h, xs, ys);
    const float sampled = remap_linear_const(depth_src, Hs, Ws, (float)xs, (float)ys, -999.0f);
    double lat2, lon2, xb, yb;
    rpc_photo2obj(rs, rpc_inv_image(rs), xs, ys, (double)sampled, lat2, lon2);
    rpc_obj2photo(rr, rpc_inv_ground(rr), lat2, lon2, (double)sampled, xb, yb);
    const double dx = xb - (double)x, dy = yb - (double)y;
    const double dist = sqrt(dx * dx + dy * dy);
    const float ddiff = fabsf(sampled - depth_ref[i]);
    const bool ok = (dist < p_ratio) && ((double)ddiff < d_ratio);
    mask[i] = ok ? 1 : 0;
    // check_geometric_consistency zeroes the heights outside the mask (rpc_filter.py:66); reproject_with_depth (the caller
    // that asks for the back-projected coordinates) returns the raw remap value, NaN / border value included (:30-47)
    depth_reproj[i] = (ok || x_back) ? sampled : 0.0f;
    x_src[i] = xs; y_src[i] = ys;
    if (x_back) { x_back[i] = xb; y_back[i] = yb; }
}

// The pinhole twin (/root/reference/tools/pinhole_filter.py:7-67).  mats = P_ref, inverse(P_ref), P_src, inverse(P_src), row-major
// 4 x 4 float64 (P = [K @ E[:3]; 0 0 0 1], formed and inverted on the host in numpy like the reference does).  Per reference pixel:
//   tmp = (d x, d y, d, 1);  xy = P_src (inverse(P_ref) tmp);  (x_src, y_src) = float32(xy[:2] / xy[2])
//   sampled = cv2.remap(depth_src, x_src, y_src, INTER_LINEAR)             default border: constant 0
//   tmp = (sampled xy0, sampled xy1, sampled, 1) with the UNROUNDED float64 xy;  back = P_ref (inverse(P_src) tmp), float32(back[:2] / back[2])
//   mask = sqrt((x_back - x)^2 + (y_back - y)^2) < p_thre  and  |sampled - d| / d < float32(relative_d_thre)      (float64 / float32 as numpy promotes)
__device__ __forceinline__ void mat4_apply(const double* __restrict__ M, double x, double y, double z, double w, double* o)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = ((M[4 * i] * x + M[4 * i + 1] * y) + M[4 * i + 2] * z) + M[4 * i + 3] * w;
}

__global__ __launch_bounds__(256)
void pinhole_geo_consistency_kernel(const float* __restrict__ depth_ref, const float* __restrict__ depth_src, const double* __restrict__ mats,
                                    int H, int W, int Hs, int Ws, double p_thre, float rel_thre,
                                    unsigned char* __restrict__ mask, float* __restrict__ depth_reproj,
                                    float* __restrict__ x_src, float* __restrict__ y_src, float* __restrict__ x_back, float* __restrict__ y_back)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i % W);
    const double* P_ref = mats, *inv_ref = mats + 16, *P_src = mats + 32, *inv_src = mats + 48;
    const float df = depth_ref[i];
    const double d = (double)df;
    double v[4], w[4];
    mat4_apply(inv_ref, d * (double)x, d * (double)y, d, 1.0, v);
    mat4_apply(P_src, v[0], v[1], v[2], v[3], w);
    const double sx = w[0] / w[2], sy = w[1] / w[2];
    const float xs = (float)sx, ys = (float)sy;
    const float sampled = remap_linear_const(depth_src, Hs, Ws, xs, ys, 0.0f);
    const double sd = (double)sampled;
    mat4_apply(inv_src, sd * sx, sd * sy, sd, 1.0, v);
    mat4_apply(P_ref, v[0], v[1], v[2], v[3], w);
    const float xb = (float)(w[0] / w[2]), yb = (float)(w[1] / w[2]);
    const double dx = (double)xb - (double)x, dy = (double)yb - (double)y;
    const double dist = sqrt(dx * dx + dy * dy);
    const float rel = __fdiv_rn(fabsf(sampled - df), df);
    const bool ok = (dist < p_thre) && (rel < rel_thre);
    mask[i] = ok ? 1 : 0;
    depth_reproj[i] = (ok || x_back) ? sampled : 0.0f;      // reproject_with_depth (x_back given) returns the raw remap value
    x_src[i] = xs; y_src[i] = ys;
    if (x_back) { x_back[i] = xb; y_back[i] = yb; }
}

}  // namespace smvs

extern "C" {

SMVS_EXPORT int smvs_pinhole_geo_consistency(const float* depth_ref, const float* depth_src, const double* mats,
                                             int H, int W, int Hs, int Ws, double p_thre, double relative_d_thre,
                                             unsigned char* mask, float* depth_reproj, float* x_src, float* y_src,
                                             float* x_back, float* y_back, void* stream)
{
    using namespace smvs;
    if (!depth_ref || !depth_src || !mats || !mask || !depth_reproj || !x_src || !y_src) return fail(SMVS_ERR_ARG, "null pointer argument");
    if ((x_back == nullptr) != (y_back == nullptr)) return fail(SMVS_ERR_ARG, "x_back and y_back go together");
    if (H < 1 || W < 1 || Hs < 1 || Ws < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(pinhole_geo_consistency_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       depth_ref, depth_src, mats, H, W, Hs, Ws, p_thre, (float)relative_d_thre, mask, depth_reproj, x_src, y_src, x_back, y_back);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "pinhole_geo_consistency launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

SMVS_EXPORT int smvs_rpc_geo_consistency(const float* depth_ref, const double* rpc_ref, const float* depth_src,
                                         const double* rpc_src, int H, int W, int Hs, int Ws, double p_ratio, double d_ratio,
                                         unsigned char* mask, float* depth_reproj, double* x_src, double* y_src,
                                         double* x_back, double* y_back, void* stream)
{
    using namespace smvs;
    if (!depth_ref || !rpc_ref || !depth_src || !rpc_src || !mask || !depth_reproj || !x_src || !y_src)
        return fail(SMVS_ERR_ARG, "null pointer argument");
    if ((x_back == nullptr) != (y_back == nullptr)) return fail(SMVS_ERR_ARG, "x_back and y_back go together");
    if (H < 1 || W < 1 || Hs < 1 || Ws < 1) return fail(SMVS_ERR_ARG, "non-positive dimension");
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(geo_consistency_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       depth_ref, rpc_ref, depth_src, rpc_src, H, W, Hs, Ws, p_ratio, d_ratio, mask, depth_reproj,
                       x_src, y_src, x_back, y_back);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SMVS_ERR_LAUNCH, "geo_consistency launch: %s", hipGetErrorString(e));
    return SMVS_OK;
}

}  // extern "C"
