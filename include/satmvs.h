/*
 * satmvs.h -- C ABI of the MI355X-native RPC plane-sweep cost-volume engine (libsatmvs_hip.so).
 *
 * The reference (WHU-GPCV/SatMVS) has no FFI layer: its boundary is the Python operator surface
 * of modules/warping.py and networks/casred.py.  Each entry point below names the reference
 * function(s) it replaces (file:line under /root/reference); satmvs_amd/modules/warping.py and
 * satmvs_amd/networks/casred.py bind them with ctypes under the reference's own names, and
 * INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - The caller owns every buffer.  All pointers are DEVICE pointers (HBM) unless stated;
 *     tensors are contiguous, float32 features/volumes in NCHW / NCDHW order, float64 camera
 *     parameters.  Nothing is allocated or freed inside the library.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Work is enqueued
 *     on it and the call returns without synchronising.
 *   - Return value: SMVS_OK (0) or an SMVS_ERR_* code; smvs_last_error() then returns a
 *     thread-local message.  Nothing is written on an argument error.
 *   - Re-entrant.  The caller selects the device (hipSetDevice / torch.cuda.device) before
 *     calling; nothing is cached per thread.  The only process-wide state is the DEFAULT arithmetic (smvs_set_arith; a call may
 *     carry its own) and a mutex-guarded,
 *     per-device pool of helper streams/events that smvs_red_pred_planes / smvs_red_volume_planes
 *     borrow for the duration of a call (bounded by the peak number of concurrent calls on a device).
 *   - The shipped library never reads the environment: tuning / A/B switches exist only in builds
 *     made with -DSMVS_TUNING (tools/ab_build.sh).
 *   - Deliberate differences from SURVEY.md section 8b's sketch: the Python binding is ctypes over
 *     this header (satmvs_amd/_lib.py), not a torch.utils.cpp_extension shim -- no torch types or
 *     headers are needed to build or call the library; and there is no smvs_shard_allreduce(ncclComm_t):
 *     the one exchange of the path (a (3,B,H,W) float64 slab, 7 MB at 768x384) goes through
 *     torch.distributed (backend "nccl" = RCCL), satmvs_amd/shard.py, DESIGN.md section 5.
 *   - depth_is_4d: 1 = per-voxel heights (B,D,H,W); 0 = per-plane heights (B,D) -- both forms
 *     of `depth_values` accepted by modules/warping.py:329-332; bits 8-9 may carry SMVS_CALL_ARITH_* (see smvs_set_arith).
 */
#ifndef SATMVS_H
#define SATMVS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SMVS_OK = 0, SMVS_ERR_ARG = 1, SMVS_ERR_LAUNCH = 2, SMVS_ERR_UNSUPPORTED = 3 };

/* Library identification: "satmvs-hip <version> gfx950". */
const char* smvs_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char* smvs_last_error(void);
/* The plane pipelines (smvs_red_pred_planes / smvs_red_volume_planes) normally fan out over two library-owned
 * non-blocking helper streams joined to the caller's stream by events.  smvs_red_set_streams(0) keeps every launch
 * of later calls on the caller's stream (slower, but legal inside hipStreamBeginCapture / hipGraph capture);
 * any other value restores the default.  Process-wide, thread-safe; returns the previous setting (0 or 2). */
int smvs_red_set_streams(int n);
/* Arithmetic of the variance build (smvs_*_costvol_fwd[_gen], and the plane pipelines that call them).
 *   SMVS_ARITH_EXACT  the reference's float32 rounding sequence operation for operation (sum, sum of squares, two true
 *                     divisions by the view count, mean^2, subtract: networks/casred.py:26-53): bit-identical to the
 *                     CPU oracle; what every bit-level test runs.
 *   SMVS_ARITH_FUSED  the same float64 geometry, float32 tap coordinates and bilinear weights, but the variance
 *                     is taken of the differences to the ref feature with its constant factors folded into the weights
 *                     (11 instead of 22 packed operations per plane and channel pair at 3 views).  Differs from the
 *                     reference by float32 rounding only -- |delta| <= 1e-5 * max(1, |v|) on the volume (SURVEY.md
 *                     section 8c; measured <= 3.1e-6 on unit-variance features).  Against a float64 evaluation of the same
 *                     taps it is 6x closer than the reference's own sequence on photo-consistent features (no
 *                     meansq - mean^2 cancellation), equal on independent random features at 2-3 views, 2-4x the
 *                     reference's rounding error at 4-8 views (tests/test_fused_arith.py).
 * Which one a call runs:
 *   - a call may carry its own: OR SMVS_CALL_ARITH_EXACT or SMVS_CALL_ARITH_FUSED into the `depth_is_4d` argument of
 *     smvs_*_costvol_fwd / smvs_red_pred_planes / smvs_red_volume_planes, or set smvs_height_gen.arith for the *_gen forms
 *     (two models, or nn.DataParallel replicas on their threads, can run different arithmetics without shared state);
 *   - the STAND-ALONE builds (smvs_*_costvol_fwd[_gen][_pc]) without a bit take the process default, which starts as
 *     SMVS_ARITH_FUSED and is moved by smvs_set_arith (thread-safe; returns the previous default, -1 for an unknown mode);
 *   - the PLANE PIPELINES (smvs_red_pred_planes / smvs_red_volume_planes[_gen]) without a bit run SMVS_ARITH_EXACT whatever
 *     the process default is (round 6): behind a peaky softmax the fused volume's 1e-5 can move a regressed height by more
 *     than north_star's 1e-3 m (2.1e-3 m at 3 of 294 912 pixels of the well-conditioned 768 x 384 cascade,
 *     profiles/r05_cascade_float64.txt), and the build is a few per cent of a pipeline's time.  The Python networks and
 *     compute_depth_* functions default the same way (satmvs_amd/_lib.py, pipeline_arith_scope).
 * Every other entry point that takes `depth_is_4d` ignores the two bits. */
enum { SMVS_ARITH_EXACT = 0, SMVS_ARITH_FUSED = 1 };
enum { SMVS_CALL_ARITH_EXACT = 0x100, SMVS_CALL_ARITH_FUSED = 0x200, SMVS_CALL_ARITH_MASK = 0x300 };
int smvs_set_arith(int mode);
int smvs_get_arith(void);
/* Releases what the library keeps between calls (the pooled helper streams and events of the plane pipelines).
 * Call with no library work in flight, e.g. before unloading; later calls re-create what they need.  Returns SMVS_OK. */
int smvs_shutdown(void);

/* ---- height hypotheses generated inside the kernels ----------------------------
 * Stages 2 and 3 of the cascades derive their hypotheses from the previous stage's height map:
 * networks/casred.py:134-145 (bilinear resize to the image size, trilinear resize of the samples to the stage
 * size) + modules/depth_range.py:4-20 (cur -/+ ndepth/2*interval, ndepth samples).  The *_gen entry points below
 * take this description instead of a (B,D,H,W) tensor and evaluate it per pixel, with ATen's rounding.
 * `interval` is a double because the reference multiplies python floats before the single cast to float32.
 * Stage 1 needs no generator: its hypotheses are (B,D) planes (depth_is_4d = 0), exactly. */
typedef struct smvs_height_gen {
    const float* prev_height;   /* device, (B, prev_h, prev_w) float32: the previous stage's "depth" output */
    int prev_h, prev_w;
    int img_h, img_w;           /* image size; img / stage size must be 1 or 2 */
    int ndepth;                 /* D of this stage */
    double interval;            /* depth_inteval_pixel = depth_interals_ratio[stage] * min_interval */
    /* UCS-Net sampler (modules/depth_range.py:45-86 behind the two bilinear resizes of networks/ucs.py:49-58) when prev_var is
     * not null: hypotheses span prev_height -+ prev_var (both resized to this stage's grid, so img_h, img_w = the stage size),
     * clamped to [range_min[b], range_max[b]]; `interval` is ignored.  All three null: the interval sampler above. */
    const float* prev_var;      /* device, (B, prev_h, prev_w): the previous stage's "variance" output */
    const float* range_min;     /* device, (B): depth_values[:, 0] */
    const float* range_max;     /* device, (B): depth_values[:, -1] */
    int arith;                  /* 0, SMVS_CALL_ARITH_EXACT or SMVS_CALL_ARITH_FUSED: arithmetic of the variance build of THIS call */
} smvs_height_gen;
/* The hypotheses as a tensor (what the reference materialises), out (B,ndepth,H,W): training path and tests. */
int smvs_height_hypotheses(const smvs_height_gen* gen, float* out, int B, int H, int W, void* stream);

/* ---- fused warp + variance cost volume ----------------------------------------------------
 * Replaces the per-source loop  rpc_warping() + volume_sum/volume_sq_sum + variance
 *   modules/warping.py:310-365 (rpc_warping), networks/casred.py:22-53 (train, whole volume),
 *   networks/casred.py:191-212 (pred, one plane), networks/casmvs.py:26-59, networks/ucs.py:27-58.
 * ref_fea (B,C,H,W); src_fea: HOST array of n_src device pointers, each (B,C,H,W), in view order;
 * rpc (B,V,170) float64 with V = n_src+1 and view 0 = reference -- the layout of
 * `proj_matrices` before torch.unbind(.,1) (casred.py:13); depth per depth_is_4d.
 * Builds planes [d_begin,d_end) of the D hypotheses; plane d is written to plane index
 * d - d_begin + d_out_off of out_var (B,C,D_out,H,W).  (Whole volume: 0,D,D,0.  One plane of the
 * pred loop: d,d+1,1,0.  Depth shard g of G: g*D/G,(g+1)*D/G,D/G,0.)
 * variance = sq/V - (sum/V)^2 in float32, accumulated ref, src0, src1, ... like the reference. */
int smvs_rpc_costvol_fwd(const float* ref_fea, const float* const* src_fea, int n_src,
                         const double* rpc, const float* depth, int depth_is_4d, float* out_var,
                         int B, int C, int D, int H, int W,
                         int d_begin, int d_end, int D_out, int d_out_off, void* stream);

/* Same with the pinhole homography of modules/warping.py:6-44 (geo_model="pinhole").
 * proj (B,n_src,4,4) float64 = src_proj @ inverse(ref_proj) per source view, as produced by
 * smvs_homo_compose. */
int smvs_homo_costvol_fwd(const float* ref_fea, const float* const* src_fea, int n_src,
                          const double* proj, const float* depth, int depth_is_4d, float* out_var,
                          int B, int C, int D, int H, int W,
                          int d_begin, int d_end, int D_out, int d_out_off, void* stream);
/* ---- plane-constant heights: collapsed source cubics (round 6) -------------------------------------
 * Stage 1 of every reference cascade sweeps PLANE-CONSTANT heights (networks/casred.py:138-149,
 * modules/depth_range.py:23-42; modules/warping.py:329-332 accepts the (B,D) form, and the broadcast (B,D,H,W)
 * form carries the same numbers).  With a plane's height fixed, RPC_Obj2Photo's four trivariate cubics per source
 * view (modules/warping.py:218-252, RPC_PLH_COEF :183-207) are bivariate in (lat, lon): 10 coefficients instead of 20.
 *   smvs_rpc_plane_coef        folds them once per (batch item, plane, source) into `plane_coef`, a caller-owned device
 *                              buffer of smvs_rpc_plane_coef_bytes(B, n_src, D) bytes, 64-byte aligned (doubles: the planes'
 *                              heights -- depth[b,d], or depth[b,d,0,0] of a 4-D tensor -- at b D + d, padded to a multiple
 *                              of 8; 24 doubles per batch item: the views' reciprocal scales, divided once; then
 *                              [b][source][cubic][d][6]: the H-dependent 6 of the 10 bivariate coefficients of each
 *                              cubic); only planes [d_begin, d_end) are written (a plane-at-a-time caller folds what it builds).
 *   smvs_rpc_costvol_fwd_pc    = smvs_rpc_costvol_fwd with that buffer (NULL: identical to smvs_rpc_costvol_fwd).  A wave
 *                              runs its source-view geometry from the folded records (heights, scales and cubics all
 *                              come from the workspace, so nothing waits on the depth tensor) and THEN compares its own
 *                              heights with the folded planes'; if any differs (per-voxel hypotheses, a jittered pixel,
 *                              NaN) it discards that work and evaluates the trivariate cubics on its own heights -- the
 *                              result never depends on trusting the caller, but heights that are not plane-constant pay
 *                              for both (cfg2 tile: 0.73 vs 0.58 ms), so send those to smvs_rpc_costvol_fwd.
 * Same polynomials re-associated: source coordinates move by float64 rounding (~1e-13 px), volumes stay inside every
 * tolerance of smvs_rpc_costvol_fwd.  `plane_coef` must have been prepared from the same rpc / depth / D; it can be
 * reused for any number of launches (plane windows, shards) of that geometry.  rpc (B,V,170), V = n_src + 1. */
size_t smvs_rpc_plane_coef_bytes(int B, int n_src, int D);
int smvs_rpc_plane_coef(const double* rpc, const float* depth, int depth_is_4d, double* plane_coef,
                        int B, int n_src, int D, int H, int W, int d_begin, int d_end, void* stream);
int smvs_rpc_costvol_fwd_pc(const float* ref_fea, const float* const* src_fea, int n_src,
                            const double* rpc, const float* depth, int depth_is_4d, const double* plane_coef,
                            float* out_var, int B, int C, int D, int H, int W,
                            int d_begin, int d_end, int D_out, int d_out_off, void* stream);

/* The same two launches with the heights generated in the kernel (see smvs_height_gen above). */
int smvs_rpc_costvol_fwd_gen(const float* ref_fea, const float* const* src_fea, int n_src,
                             const double* rpc, const smvs_height_gen* gen, float* out_var,
                             int B, int C, int D, int H, int W,
                             int d_begin, int d_end, int D_out, int d_out_off, void* stream);
int smvs_homo_costvol_fwd_gen(const float* ref_fea, const float* const* src_fea, int n_src,
                              const double* proj, const smvs_height_gen* gen, float* out_var,
                              int B, int C, int D, int H, int W,
                              int d_begin, int d_end, int D_out, int d_out_off, void* stream);

/* ---- stand-alone warps (the operator surface itself) ------------------------------------------
 * smvs_rpc_warp_fwd  = rpc_warping(src_fea, src_rpc, ref_rpc, depth_values, coef)
 *                      modules/warping.py:310-365 (coef is not needed); also serves
 *                      rpc_warping_enisum (:139-178) after the QC tensors are mapped back to the
 *                      20 coefficients on the host.  src_rpc, ref_rpc (B,170); out (B,C,D,H,W).
 * smvs_rpc_warp_bwd  = its autograd w.r.t. src_fea (the grid is built under no_grad,
 *                      warping.py:322): grad_src (B,C,H,W) must be zero-filled by the caller,
 *                      contributions are accumulated with float32 atomics. */
int smvs_rpc_warp_fwd(const float* src_fea, const double* src_rpc, const double* ref_rpc,
                      const float* depth, int depth_is_4d, float* out,
                      int B, int C, int D, int H, int W, void* stream);
int smvs_rpc_warp_bwd(const float* grad_out, const double* src_rpc, const double* ref_rpc,
                      const float* depth, int depth_is_4d, float* grad_src,
                      int B, int C, int D, int H, int W, void* stream);

/* homo_warping(src_fea, src_proj, ref_proj, depth_values), modules/warping.py:6-44.
 * proj (B,4,4) = src_proj @ inverse(ref_proj) from smvs_homo_compose. */
int smvs_homo_warp_fwd(const float* src_fea, const double* proj, const float* depth, int depth_is_4d,
                       float* out, int B, int C, int D, int H, int W, void* stream);
int smvs_homo_warp_bwd(const float* grad_out, const double* proj, const float* depth, int depth_is_4d,
                       float* grad_src, int B, int C, int D, int H, int W, void* stream);
/* out[b] = src_proj[b] @ inverse(ref_proj[b]) (warping.py:19), 4x4 float64, n matrices. */
int smvs_homo_compose(const double* src_proj, const double* ref_proj, double* out, int n, void* stream);

/* ---- backward of the fused volume (train.py:284 loss.backward through casred.py:22-53) -------
 * grad_var (B,C,D,H,W) -> grad_ref (B,C,H,W) and grad_src[s] (B,C,H,W), all ACCUMULATED with
 * float32 atomics (zero-filled by the caller; grad_src is a HOST array of n_src device pointers).  geo_kind 0 = rpc (B,V,170), 1 = homography
 * (B,n_src,4,4).  Recomputes the taps instead of saving a warped volume.  Where the taps of a wave's 32 x 2 pixels x 8
 * planes fall into a 64 x 8-cell box of every source view (up to four source views) the contributions are summed in
 * float64 in LDS first and reach memory as one float32 atomic per touched cell; summation order is not fixed either way
 * (atomics), values differ from a sequential float32 sum by rounding only.  Limits: C*H*W*4 < 2 GiB, H, W <= 32766. */
int smvs_costvol_bwd(int geo_kind, const float* grad_var, const float* ref_fea, const float* const* src_fea,
                     int n_src, const double* geo, const float* depth, int depth_is_4d,
                     float* grad_ref, float* const* grad_src,
                     int B, int C, int D, int H, int W, void* stream);

/* ---- batch projectors --------------------------------------------------------------------------
 * RPC_Photo2Obj / RPC_Obj2Photo (modules/warping.py:255-307, :218-252) and the offline-tool twins
 * tools/RPCCore.py:424-489, tools/rpc_tensor.py:109-165 on flat float64 arrays.
 * dir 0: (samp, line, h) -> (lat, lon);  dir 1: (lat, lon, h) -> (samp, line).
 * rpc170: one 170-vector (device).  a, b, h, o0, o1: n doubles each (device). */
int smvs_rpc_project(const double* rpc170, const double* a, const double* b, const double* h,
                     double* o0, double* o1, size_t n, int dir, void* stream);

/* ---- geometric-consistency check (post-processing) ---------------------------------
 * One (reference, source) pair of tools/rpc_filter.py:11-70 (reproject_with_depth + check_geometric_consistency):
 * reference pixel + height -> ground -> source image (float64); the source height map sampled there like
 * cv2.remap(INTER_LINEAR, BORDER_CONSTANT, -999) on float32 coordinates; back to the ground with the sampled
 * height and into the reference image; mask = (|reprojected - pixel| < p_ratio) & (|sampled - height| < d_ratio).
 * depth_ref (H,W), depth_src (Hs,Ws) float32; rpc_* 170 float64; mask (H,W) uint8; depth_reproj (H,W) float32: 0 outside
 * the mask when x_back/y_back are NULL (check_geometric_consistency), the raw sampled height everywhere when they are
 * given (reproject_with_depth); x_src, y_src (H,W) float64 source-image coordinates; x_back, y_back (H,W) float64 or
 * both NULL. */
int smvs_rpc_geo_consistency(const float* depth_ref, const double* rpc_ref, const float* depth_src,
                             const double* rpc_src, int H, int W, int Hs, int Ws, double p_ratio, double d_ratio,
                             unsigned char* mask, float* depth_reproj, double* x_src, double* y_src,
                             double* x_back, double* y_back, void* stream);

/* The pinhole twin: one (reference, source) pair of tools/pinhole_filter.py:7-67 (reproject_with_depth +
 * check_geometric_consistency).  mats (device): P_ref, inverse(P_ref), P_src, inverse(P_src), row-major 4 x 4 float64 with
 * P = [K @ E[:3]; 0 0 0 1] (:17-24; formed and inverted by the caller, on the host like the reference).  Reference pixel * depth
 * -> world -> source pixel in float64, float32 coordinates into cv2.remap(INTER_LINEAR, default border: constant 0), the
 * sampled depth back into the reference view; mask = (|reprojected - pixel| < p_thre) & (|sampled - depth| / depth <
 * float32(relative_d_thre)).  depth_ref (H,W), depth_src (Hs,Ws) float32; mask uint8; depth_reproj float32 (0 outside the
 * mask when x_back / y_back are NULL, the raw sampled depth when they are given); x_src, y_src, x_back, y_back (H,W) float32. */
int smvs_pinhole_geo_consistency(const float* depth_ref, const float* depth_src, const double* mats,
                                 int H, int W, int Hs, int Ws, double p_thre, double relative_d_thre,
                                 unsigned char* mask, float* depth_reproj, float* x_src, float* y_src,
                                 float* x_back, float* y_back, void* stream);

/* ---- GroupNorm(1, C) of the recurrent regulariser, training path -------------------------------------
 * reference: modules/module.py:15-20 (three nn.GroupNorm(1, C, 1e-5) per ConvGRU cell) and :38-52 (sigmoid / tanh of
 * the normalised gates); differentiated by train.py:284.  x (B,C,HW) float32 with batch stride x_batch_stride elements
 * (>= C*HW: the two gate halves of a (B,2C,H,W) tensor are normalised where they lie); act 0 none, 1 sigmoid, 2 tanh:
 *   y = act((x - mean_b) * rstd_b * gamma_c + beta_c),   mean / variance over the C*HW values of sample b.
 * mean_rstd (B,2) float32 out (kept for the backward); workspace: 2*B*ceil(C*HW/4096) doubles (forward), 2*B*C*ceil(HW/4096) doubles
 * (backward), caller-owned, contents irrelevant on entry.  Backward: dx (batch stride dx_batch_stride), dgamma (C),
 * dbeta (C) are overwritten; y = the forward's output (needed when act != 0).  Statistics and the reductions of the
 * backward are accumulated in float64, without atomics (fixed order: deterministic). */
int smvs_groupnorm1_fwd(const float* x, long long x_batch_stride, const float* gamma, const float* beta, float eps,
                        int act, float* y, float* mean_rstd, double* workspace, int B, int C, int HW, void* stream);
int smvs_groupnorm1_bwd(const float* dy, const float* x, long long x_batch_stride, const float* y, const float* gamma,
                        const float* mean_rstd, int act, float* dx, long long dx_batch_stride, float* dgamma,
                        float* dbeta, double* workspace, int B, int C, int HW, void* stream);

/* Weight and bias gradient of a 3x3, stride-1, pad-1 convolution (the ConvGRU cells' gate_conv / output_conv, modules/module.py:13-14
 * under loss.backward(), train.py:284):  dw[co][ci][ky][kx] += sum_{b,y,x} dy[b][co][y][x] * x[b][ci][y+ky-1][x+kx-1],
 * db[co] += sum dy[b][co][y][x].  x (B,Cin,H,W), dy (B,Cout,H,W), dw (Cout,Cin,3,3), db (Cout) or NULL; dw and db are ACCUMULATED
 * with float atomics -- the caller zero-fills them for a plain gradient. */
int smvs_conv3x3_wgrad(const float* x, const float* dy, float* dw, float* db, int B, int Cin, int Cout, int H, int W, void* stream);
/* The same correlation with the window tensor read at `stride` 1 or 2 -- the weight gradient of every 3x3 layer of the RED
 * regulariser (modules/module.py:595-693 under loss.backward()):
 *   dw[g][c][ky][kx] += sum_{b,y,x} grid[b][g][y][x] * window[b][c][stride*y + ky - 1][stride*x + kx - 1]
 * grid (B,Cgrid,H,W), window (B,Cwin,stride*H,stride*W), dw (Cgrid,Cwin,3,3); dgrid_sum (Cgrid) += sum of grid, or NULL.
 *   nn.Conv2d(stride s, pad 1): window = input, grid = output gradient -> dw = weight gradient (Cout,Cin,3,3), dgrid_sum = bias gradient;
 *   nn.ConvTranspose2d(stride s, pad 1, output_padding s-1): window = output gradient, grid = input -> dw = weight gradient
 *   (Cin_layer,Cout_layer,3,3) (its bias gradient is the plain sum of the output gradient: not computed here). */
int smvs_conv3x3_wgrad_strided(const float* window, const float* grid, float* dw, float* dgrid_sum,
                               int B, int Cwin, int Cgrid, int H, int W, int stride, void* stream);
/* smvs_conv3x3_wgrad for a convolution over cat(xA, xB) without the concatenated tensor: xA (B,CA,H,W), xB (B,CB,H,W) (CB = 0: xA only;
 * CA even otherwise), dw (Cout, CA+CB, 3, 3). */
int smvs_conv3x3_wgrad_cat(const float* xA, int CA, const float* xB, int CB, const float* dy, float* dw, float* db,
                           int B, int Cout, int H, int W, void* stream);
/* The same sums over a LIST of n tensors (host arrays of n device pointers each; win2 NULL when CB = 0) of Bper samples each -- the planes
 * of a training step's plane loop, whose activations and gradients are separate allocations: one launch per layer and step instead of one
 * per layer and plane.  window tensors (Bper, CA [+ CB], stride*H, stride*W), grid tensors (Bper, Cgrid, H, W); dw (Cgrid, CA+CB, 3, 3)
 * and dgrid_sum (Cgrid, or NULL) are accumulated into (the caller zeroes them). */
int smvs_conv3x3_wgrad_list(const float* const* win, const float* const* win2, const float* const* grid, int n,
                            float* dw, float* dgrid_sum, int Bper, int CA, int CB, int Cgrid, int H, int W, int stride, void* stream);

/* Weight gradient of the 3x3x3 / pad 1 layers of the 3-D regulariser CostRegNet (modules/module.py:324-410, 546-577 under
 * loss.backward(), train.py:284 with --model casmvs / ucs) -- the 2-D correlation above with a depth axis:
 *   dw[g][c][kd][ky][kx] += sum_{b,d,y,x} grid[b][g][d][y][x] * window[b][c][s*d + kd - 1][s*y + ky - 1][s*x + kx - 1]
 * grid (B,Cgrid,D,H,W), window (B,Cwin,s*D,s*H,s*W), dw (Cgrid,Cwin,3,3,3) ACCUMULATED into (the caller zero-fills it), s = stride 1 or 2.
 *   nn.Conv3d(stride s, pad 1): window = input, grid = output gradient -> dw = weight gradient (Cout,Cin,3,3,3);
 *   nn.ConvTranspose3d(stride 2, pad 1, output_padding 1): window = output gradient, grid = input -> dw = weight gradient
 *   (Cin_layer,Cout_layer,3,3,3).  Volumes are read in place through their (B,C,D,H,W) strides; a channel of either tensor must stay
 *   below 2^31 bytes (SMVS_ERR_ARG otherwise: callers keep torch's operator).
 *   workspace: NULL -- the waves add their partial sums to dw with float atomics (order-dependent rounding; the ~100-400 waves that
 *   share 144 weights serialise on five cache lines) -- or smvs_conv3d_wgrad_workspace_floats(...) floats of scratch: the waves store their
 *   partial sums there and a second kernel adds them to dw in a fixed order (deterministic, and what the shipped training path uses). */
size_t smvs_conv3d_wgrad_workspace_floats(int B, int Cwin, int Cgrid, int D, int H, int W);
int smvs_conv3d_wgrad(const float* window, const float* grid, float* dw, float* workspace, size_t workspace_floats, int B, int Cwin,
                      int Cgrid, int D, int H, int W, int stride, void* stream);

/* A single 3x3x3 / pad 1 layer of CostRegNet as a stand-alone call on the kernels of smvs_costreg_fwd (direct or MFMA by channel
 * count) WITHOUT the folded BatchNorm -- the TRAINING forward of its convolutions (modules/module.py:324-410 under autograd, batch
 * statistics follow as a separate operator) and their input gradients, which are the adjoint layers on the same kernels:
 *   smvs_conv3d_packed_floats(cin, cout)  floats of the packed weights of a layer from cin to cout channels
 *   smvs_conv3d_pack(w, packed, cin, cout, layout): weights read from w as
 *       layout 0: w[co][ci][27]       an nn.Conv3d weight (kinds 0 / 1); the input gradient of an nn.ConvTranspose3d of weight (cout,cin,27)
 *       layout 1: w[ci][co][27]       scatter taps for kind 2: an nn.ConvTranspose3d(stride 2) weight; the input gradient of a stride-2
 *                                     nn.Conv3d of weight (cin, cout, 27)
 *       layout 2: w[ci][co][26 - k]   the input gradient of a stride-1 nn.Conv3d of weight (cin, cout, 27) as a correlation (kind 0)
 *   smvs_conv3d_fwd(kind, ...)  out = [relu](layer(in (B,Cin,Di,Hi,Wi))) [+ skip]
 *       kind 0: correlation, stride 1, out (B,Cout,Di,Hi,Wi);   kind 1: correlation, stride 2 (even dims), out (B,Cout,Di/2,Hi/2,Wi/2);
 *       kind 2: transposed convolution, stride 2, output_padding 1, out (B,Cout,2Di,2Hi,2Wi).   skip: NULL or a tensor of out's shape. */
size_t smvs_conv3d_packed_floats(int cin, int cout);
int smvs_conv3d_pack(const float* w, float* packed, int cin, int cout, int layout, void* stream);
int smvs_conv3d_fwd(int kind, const float* in, const float* packed, const float* skip, float* out, int B, int Cin, int Cout,
                    int Di, int Hi, int Wi, int relu, void* stream);

/* nn.BatchNorm3d in TRAINING form (batch statistics over (B, N = D*H*W) per channel) with the block's ReLU -- the normalisation of every
 * Conv3d / Deconv3d block of CostRegNet under autograd (modules/module.py:324-410):
 *   fwd: y = [relu]((x - mean) * rstd * gamma + beta);  saved_mean_rstd (C,2) for the backward (an opaque pair: the mean is kept
 *        relative to the channel's first element, which both directions re-read from x, so that |mean| >> std loses nothing);  running_mean / running_var (or NULL, NULL)
 *        updated like torch.nn.functional.batch_norm(training=True): (1 - momentum) * running + momentum * batch (unbiased variance);
 *        num_batches_tracked (int64 scalar on the device, or NULL) += 1 like nn.BatchNorm's forward
 *   bwd: dx, dgamma (C), dbeta (C) from dy, the layer's INPUT x and saved_mean_rstd; with relu != 0 the gradient passes where the forward's
 *        output was positive (recomputed from x: neither a mask nor the output is kept)
 * x, y, dy, dx: (B,C,N) contiguous float32; workspace: 2*C doubles of scratch, cleared by the call unless `relu` carries
 * SMVS_BN_WORKSPACE_ZERO (the caller hands over zeroed memory: one fill for all the layers of a forward).  relu: bit 0 = apply ReLU. */
#define SMVS_BN_WORKSPACE_ZERO 2
int smvs_batchnorm_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                             long long* num_batches_tracked, float momentum, float eps, int relu, float* y, float* saved_mean_rstd,
                             double* workspace, int B, int C, long long N, void* stream);
int smvs_batchnorm_train_bwd(const float* dy, const float* x, const float* gamma, const float* beta, const float* saved_mean_rstd, int relu,
                             float* dx, float* dgamma, float* dbeta, double* workspace, int B, int C, long long N, void* stream);

/* A single 3x3 / pad 1 layer of the RED regulariser as a stand-alone call on the kernels of the plane loop -- the TRAINING forward of
 * its convolutions (modules/module.py:34-57, :625-644 under autograd) and their input gradients:
 *   smvs_conv3x3_packed_floats(cin, cout)  floats of the packed weights
 *   smvs_conv3x3_pack(w, packed, cin, cout, layout): the correlation from cin to cout channels whose weights are read from w as
 *       layout 0: w[co][ci][ky][kx]      an nn.Conv2d weight (stride 1 / 2); the input gradient of an nn.ConvTranspose2d of weight (cout,cin,3,3)
 *       layout 1: w[ci][co][ky][kx] kept as scatter taps for kind 2: an nn.ConvTranspose2d(stride 2) weight; the input gradient of a
 *                                        stride-2 nn.Conv2d of weight (cin, cout, 3, 3)
 *       layout 2: w[ci][co][2-ky][2-kx]  a stride-1 nn.ConvTranspose2d as a correlation; the input gradient of a stride-1 nn.Conv2d of
 *                                        weight (cin, cout, 3, 3)
 *   smvs_conv3x3_fwd(kind, ...)  out = [relu](layer(cat(xA (B,CA,H,W), xB (B,CB,H,W) or NULL)) + bias (Cout) or NULL)
 *       kind 0: correlation, stride 1, out (B,Cout,H,W);   kind 1: correlation, stride 2 (H, W even), out (B,Cout,H/2,W/2);
 *       kind 2: transposed convolution, stride 2, pad 1, output_padding 1 (layout-1 weights; one operand, no bias), out (B,Cout,2H,2W)
 * float32; accumulation in input-channel order (direct kernels) or as a k-ordered fmaf chain on v_mfma_f32_32x32x2_f32 (correlations
 * with 32 / 64 / 128 output channels): same class of rounding as torch's direct convolution, 2e-5 relative in the tests. */
size_t smvs_conv3x3_packed_floats(int cin, int cout);
int smvs_conv3x3_pack(const float* w, float* packed, int cin, int cout, int layout, void* stream);
int smvs_conv3x3_fwd(int kind, const float* xA, int CA, const float* xB, int CB, const float* packed, const float* bias, const float* init,
                     float* out, int B, int Cout, int H, int W, int relu, void* stream);
/* init (kinds 0 / 1; same shape as out, or NULL; may be out itself): added to the sums before bias / ReLU -- an input gradient that continues
 * the contributions already collected for that tensor (whole-cell ConvGRU backward). */

/* Both gate norms of a ConvGRU cell in one call (modules/module.py:15-16, :37-40): x (B, 2C, HW) contiguous = the gate
 * convolution's output; channels [0, C) are normalised with (gamma, beta), channels [C, 2C) with (gamma2, beta2), each half
 * over its own C*HW values, then the activation.  y and dx (B, 2C, HW); mean_rstd (2B, 2) (sample 2b + half); workspace
 * as for smvs_groupnorm1_* with 2B samples (4*B*ceil(C*HW/4096) / 4*B*C*ceil(HW/4096) doubles). */
int smvs_groupnorm1_pair_fwd(const float* x, const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                             float eps, int act, float* y, float* mean_rstd, double* workspace, int B, int C, int HW,
                             void* stream);
/* The same two forwards with the cell's next step folded into the apply pass (one launch per step and cell less, each):
 *   _fwd_blend:      additionally out = u * h + (1 - u) * y (modules/module.py:57); u (B,C,HW) at batch stride u_batch_stride elements
 *                    (the u half of the (B,2C,H,W) gate tensor), h and out (B,C,HW) contiguous; y is written as well (the backward needs it);
 *   _pair_fwd_mul:   additionally rh = y[:, :C] * h (modules/module.py:43: the second operand of the candidate convolution, which
 *                    smvs_conv3x3_fwd takes as (x, rh): no concatenation); h, rh (B,C,HW) contiguous. */
int smvs_groupnorm1_fwd_blend(const float* x, long long x_batch_stride, const float* gamma, const float* beta, float eps,
                              int act, float* y, float* mean_rstd, double* workspace, const float* u, long long u_batch_stride,
                              const float* h, float* out, int B, int C, int HW, void* stream);
int smvs_groupnorm1_pair_fwd_mul(const float* x, const float* gamma, const float* beta, const float* gamma2, const float* beta2,
                                 float eps, int act, float* y, float* mean_rstd, double* workspace, const float* h, float* rh,
                                 int B, int C, int HW, void* stream);
int smvs_groupnorm1_pair_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* gamma2,
                             const float* mean_rstd, int act, float* dx, float* dgamma, float* dbeta, float* dgamma2,
                             float* dbeta2, double* workspace, int B, int C, int HW, void* stream);

/* The ConvGRU cell's element-wise steps (modules/module.py:43-44 and :57), training path, one launch each way:
 *   smvs_gru_mul_cat:  out (B, Cx+Ch, HW) = cat(x (B,Cx,HW), r * h (B,Ch,HW));   backward: dr = dcat[:, Cx:] * h, dh = dcat[:, Cx:] * r
 *                      (dx is the first Cx channels of dcat as they are)
 *   smvs_gru_blend:    out = u * h + (1 - u) * y over n contiguous floats (pointers 16-byte aligned);
 *                      backward: du = dy (h - y), dh = dy u, dcand = dy (1 - u). */
int smvs_gru_mul_cat_fwd(const float* x, const float* r, const float* h, float* out, int B, int Cx, int Ch, int HW, void* stream);
int smvs_gru_mul_cat_bwd(const float* dcat, const float* r, const float* h, float* dr, float* dh, int B, int Cx, int Ch, int HW, void* stream);
/* the same with the state gradient accumulated and stored in place: dcat[:, Cx:] <- dcat[:, Cx:] * r + dh_acc; dr = dcat[:, Cx:] * h */
int smvs_gru_mul_cat_bwd_acc(float* dcat, const float* r, const float* h, const float* dh_acc, float* dr, int B, int Cx, int Ch, int HW, void* stream);
int smvs_gru_blend_fwd(const float* u, const float* h, const float* y, float* out, long long n, void* stream);
int smvs_gru_blend_bwd(const float* dy, const float* u, const float* h, const float* y, float* du, float* dh, float* dcand, long long n, void* stream);

/* ---- regression ----------------------------------------------------------------------------------
 * Train path: softmax over D + expected height + max probability,
 *   networks/casred.py:58-62 and modules/module.py:433-439 (depth_regression).
 * reg (B,D,H,W) float32 regulariser output; out_depth, out_conf (B,H,W). */
int smvs_softmax_regress_fwd(const float* reg, const float* depth, int depth_is_4d,
                             float* out_depth, float* out_conf, int B, int D, int H, int W, void* stream);
/* CascadeMVSNet / UCSNet flavour: softmax over D + expected height + the probability mass of the four hypotheses
 * around the expected index (networks/casmvs.py:66-74: F.pad(.,(1,2)) + 4*avg_pool3d((4,1,1)) gathered at
 * clamp(trunc(E[index]))), and, when out_var is not NULL, UCSNet's lamb * sqrt(sum p*(h - depth)^2)
 * (networks/ucs.py:73-74).  reg (B,D,H,W); out_depth, out_conf, out_var (B,H,W). */
int smvs_window_regress_fwd(const float* reg, const float* depth, int depth_is_4d,
                            float* out_depth, float* out_conf, float* out_var, float lamb,
                            int B, int D, int H, int W, void* stream);
/* softmax / window regression with the heights generated in the kernel (smvs_height_gen, gen->ndepth == D) */
int smvs_softmax_regress_fwd_gen(const float* reg, const smvs_height_gen* gen,
                                 float* out_depth, float* out_conf, int B, int D, int H, int W, void* stream);
int smvs_window_regress_fwd_gen(const float* reg, const smvs_height_gen* gen,
                                float* out_depth, float* out_conf, float* out_var, float lamb,
                                int B, int D, int H, int W, void* stream);
/* Pred path, one plane d: prob = exp(double(reg)); max_prob = max(.,prob); depth_img += h*prob;
 * exp_sum += prob (networks/casred.py:218-231).  Accumulators (B,H,W) float64, zeroed by the
 * caller before plane 0.  reg_plane (B,H,W). */
int smvs_stream_regress_step(const float* reg_plane, const float* depth, int depth_is_4d,
                             double* exp_sum, double* depth_img, double* max_prob,
                             int B, int D, int H, int W, int d, void* stream);
/* depth = depth_img/(exp_sum+1e-10), conf = max_prob/(exp_sum+1e-10) -> float32
 * (networks/casred.py:234-236).  n = B*H*W. */
int smvs_stream_regress_final(const double* exp_sum, const double* depth_img, const double* max_prob,
                              float* out_depth, float* out_conf, size_t n, void* stream);
/* Reduce step of the plane-sharded regression's exchange (satmvs_amd/shard.py; north_star's "RCCL all-reduce of the per-plane
 * cost slab", networks/casred.py:176-236 sharded over height planes): recv holds `world` copies [rank][chunk] (one per rank, in
 * rank order) of `chunk` consecutive elements, starting at element `first`, of the flattened (3,B,H,W) float64 accumulators
 * [exp_sum | depth_img | max_prob] (row_len = B*H*W); out[i] = sum over ranks for elements of the first two rows, max for the
 * third, folded in rank order.  out may be the slab position the following all-gather sends from. */
int smvs_regress_fold(const double* recv, double* out, int world, size_t chunk, size_t first, size_t row_len, void* stream);

/* ---- recurrent encoder-decoder regulariser (RED), one height plane per call ----------------------
 * Replaces slice_RED_Regularization.forward (modules/module.py:672-693) and the loop body of
 * RED_Regularization.forward (:625-644) incl. ConvGRUCell2 (:6-58).  Hidden sizes 8/16/32/64 as in
 * the reference (:617-620); C = input (feature) channels.
 *
 * smvs_red_pack_weights: params = HOST array of 48 device pointers, the module's parameters in this
 * order -- for conv_gru1..4: gate_conv.weight, gate_conv.bias, reset_gate_norm.weight, .bias,
 * update_gate_norm.weight, .bias, output_conv.weight, .bias, output_norm.weight, .bias; then
 * conv1.conv.weight, conv2.conv.weight, conv3.conv.weight, upconv1.conv.weight, upconv2.conv.weight,
 * upconv3.conv.weight, upconv2d.weight, upconv2d.bias.  packed: smvs_red_packed_floats(C) floats,
 * owned by the caller; repack whenever the parameters change.
 * smvs_red_step_fwd: cost (B,C,H,W) = the variance plane (the network consumes -cost); state1..4
 * (B,8,H,W) (B,16,H/2,W/2) (B,32,H/4,W/4) (B,64,H/8,W/8) updated in place; reg_out (B,1,H,W);
 * workspace of smvs_red_workspace_bytes(B,C,H,W) bytes; H, W multiples of 8. */
size_t smvs_red_packed_floats(int C);
size_t smvs_red_workspace_bytes(int B, int C, int H, int W);
int smvs_red_pack_weights(const float* const* params, int C, float* packed, void* stream);
int smvs_red_step_fwd(const float* packed, const float* cost, float* state1, float* state2, float* state3,
                      float* state4, float* reg_out, void* workspace, size_t workspace_bytes,
                      int B, int C, int H, int W, void* stream);

/* The plane loop of compute_depth_when_pred (networks/casred.py:191-231) for planes [d_begin,d_end) in
 * one call: per plane  fused warp+variance of that plane -> RED step -> float64 streaming regression,
 * enqueued back to back.  acc (3,B,H,W) float64 = [exp_sum, depth_img, max_prob] (zeroed by the caller
 * before plane 0; finish with smvs_stream_regress_final, or all-reduce it first when planes are sharded
 * over GPUs).  geo: rpc (B,V,170) for geo_kind 0, composed homographies (B,n_src,4,4) for geo_kind 1.
 * workspace: smvs_red_pred_workspace_bytes(B,C,H,W) bytes. */
size_t smvs_red_pred_workspace_bytes(int B, int C, int H, int W);
int smvs_red_pred_planes(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                         const double* geo, const float* depth, int depth_is_4d, const float* packed,
                         float* state1, float* state2, float* state3, float* state4, double* acc,
                         void* workspace, size_t workspace_bytes,
                         int B, int C, int D, int H, int W, int d_begin, int d_end, void* stream);

/* Same plane pipeline with the regularised planes written to reg_volume (B,D,H,W) instead of the regression
 * accumulators: the whole-volume network's path (compute_depth_when_train under no_grad: networks/casred.py:22-62
 * with RED_Regularization.forward, modules/module.py:625-647) without materialising the (B,C,D,H,W) variance
 * volume; follow with smvs_softmax_regress_fwd. */
int smvs_red_volume_planes(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                           const double* geo, const float* depth, int depth_is_4d, const float* packed,
                           float* state1, float* state2, float* state3, float* state4, float* reg_volume,
                           void* workspace, size_t workspace_bytes,
                           int B, int C, int D, int H, int W, int d_begin, int d_end, void* stream);
/* Both plane pipelines with the heights generated in the kernels (smvs_height_gen, gen->ndepth == D). */
int smvs_red_pred_planes_gen(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                             const double* geo, const smvs_height_gen* gen, const float* packed,
                             float* state1, float* state2, float* state3, float* state4, double* acc,
                             void* workspace, size_t workspace_bytes,
                             int B, int C, int D, int H, int W, int d_begin, int d_end, void* stream);
int smvs_red_volume_planes_gen(int geo_kind, const float* ref_fea, const float* const* src_fea, int n_src,
                               const double* geo, const smvs_height_gen* gen, const float* packed,
                               float* state1, float* state2, float* state3, float* state4, float* reg_volume,
                               void* workspace, size_t workspace_bytes,
                               int B, int C, int D, int H, int W, int d_begin, int d_end, void* stream);

/* ---- 3-D convolutional cost regulariser (CostRegNet), inference form ---------------------------------
 * Replaces CostRegNet.forward (modules/module.py:546-577; Conv3d :324, Deconv3d :369) for
 * CascadeMVSNet (networks/casmvs.py) and UCSNet (networks/ucs.py); base_channels = 8.  BatchNorm3d uses
 * its running statistics (eval mode), folded into a per-channel scale/shift.
 * smvs_costreg_pack_weights: params = HOST array of 51 device pointers -- for conv0, conv1, conv2, conv3,
 * conv4, conv5, conv6, conv7, conv9, conv11: conv.weight, bn.weight, bn.bias, bn.running_mean,
 * bn.running_var; then prob.weight.  packed: smvs_costreg_packed_floats(C) floats owned by the caller.
 * smvs_costreg_fwd: vol (B,C,D,H,W) variance volume -> out (B,1,D,H,W); D, H, W multiples of 8;
 * workspace of smvs_costreg_workspace_bytes bytes. */
size_t smvs_costreg_packed_floats(int C);
size_t smvs_costreg_workspace_bytes(int B, int C, int D, int H, int W);
int smvs_costreg_pack_weights(const float* const* params, int C, float* packed, void* stream);
int smvs_costreg_fwd(const float* packed, const float* vol, float* out, void* workspace, size_t workspace_bytes,
                     int B, int C, int D, int H, int W, void* stream);

/* ---- feature extractor (FeatureNet), inference form ---------------------------------------------------
 * Replaces FeatureNet.forward (modules/module.py:442-543; num_stage 3; Conv2d :19-60, Deconv2d :62-114,
 * DeConv2dFuse :117-140) applied to every view (networks/casred.py:116-121): all views of all samples in one
 * call (N = B*V images).  arch 0 = arch_mode "unet" (casred, ucs), arch 1 = arch_mode "fpn" (casmvs: 1x1
 * laterals added to the nearest-upsampled coarser level, module.py:527-536).  BatchNorm2d uses its running
 * statistics, folded into a per-channel scale/shift.
 * smvs_featnet_pack_weights: params = HOST array of device pointers.  Both variants: for conv0.0, conv0.1,
 * conv1.0, conv1.1, conv1.2, conv2.0, conv2.1, conv2.2: conv.weight, bn.weight, bn.bias, bn.running_mean,
 * bn.running_var (40).  arch 0 continues with the same five for deconv1.deconv, deconv1.conv, deconv2.deconv,
 * deconv2.conv, then out1.weight, out2.weight, out3.weight (63 in all); arch 1 with out1.weight,
 * inner1.weight, inner1.bias, out2.weight, inner2.weight, inner2.bias, out3.weight (47 in all).
 * packed: smvs_featnet_packed_floats(base_channels, arch) floats owned by the caller.
 * smvs_featnet_fwd: imgs (N,3,H,W) -> stage1 (N,4c,H/4,W/4), stage2 (N,2c,H/2,W/2), stage3 (N,c,H,W);
 * H, W multiples of 4; workspace of smvs_featnet_workspace_bytes bytes. */
size_t smvs_featnet_packed_floats(int base_channels, int arch);
size_t smvs_featnet_workspace_bytes(int N, int H, int W, int base_channels, int arch);
int smvs_featnet_pack_weights(const float* const* params, int base_channels, int arch, float* packed, void* stream);
int smvs_featnet_fwd(const float* packed, const float* imgs, float* stage1, float* stage2, float* stage3,
                     void* workspace, size_t workspace_bytes, int N, int H, int W, int base_channels, int arch,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SATMVS_H */
