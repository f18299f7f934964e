// smvs_host.h -- host-side plumbing shared by the C-ABI entry points: error codes, the
// thread-local last-error message, export macro.  No global mutable state besides that
// thread-local buffer: entry points are re-entrant (nn.DataParallel replicas call in from one
// Python thread per device, /root/reference/train.py:129).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/satmvs.h"

#define SMVS_EXPORT __attribute__((visibility("default")))

namespace smvs {

char* last_error_buf();          // thread-local, 512 bytes

__attribute__((format(printf, 2, 3)))
inline int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// smvs_height_gen (C ABI) -> device-side HeightGen for a stage of size H x W; returns a message on bad arguments
struct HeightGenHost { const float* prev; int hp, wp, ih, iw, scale; float c, ndm1; const float *var, *rmin, *rmax; };
inline const char* height_gen_check(const smvs_height_gen* g, int D, int H, int W, HeightGenHost& o)
{
    if (!g || !g->prev_height) return "null height generator";
    if (g->prev_h < 1 || g->prev_w < 1 || g->img_h < 1 || g->img_w < 1 || g->ndepth < 2) return "bad height generator sizes";
    if (g->ndepth != D) return "height generator ndepth differs from D";
    if (g->img_h % H || g->img_w % W || g->img_h / H != g->img_w / W) return "image size is not an integer multiple of the stage size";
    const int scale = g->img_h / H;
    if (scale != 1 && scale != 2) return "generated heights support image/stage scale 1 or 2 (stage 1 passes (B,D) planes)";
    o.prev = g->prev_height; o.hp = g->prev_h; o.wp = g->prev_w; o.ih = g->img_h; o.iw = g->img_w; o.scale = scale;
    o.c = (float)(g->ndepth / 2.0 * (double)g->interval);
    o.ndm1 = (float)(g->ndepth - 1);
    o.var = g->prev_var; o.rmin = g->range_min; o.rmax = g->range_max;
    if (o.var) {
        if (!o.rmin || !o.rmax) return "UCS height generator needs range_min and range_max";
        if (scale != 1) return "UCS height generator resizes straight to the stage grid (img size = stage size)";
    }
    return nullptr;
}

// A/B and tuning switches exist only in tuning builds (tools/ab_build.sh x -DSMVS_TUNING); the shipped library
// never reads the environment: every switch folds to its default at compile time.
#ifdef SMVS_TUNING
#include <stdlib.h>
inline int tune_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
constexpr int tune_int(const char*, int dflt) { return dflt; }
#endif

}  // namespace smvs
