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

// A/B and tuning switches exist only in tuning builds (tools/ab_build.sh x -DSMVS_TUNING); the shipped library
// never reads the environment: every switch folds to its default at compile time.
#ifdef SMVS_TUNING
#include <stdlib.h>
inline int tune_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
constexpr int tune_int(const char*, int dflt) { return dflt; }
#endif

}  // namespace smvs
