// costvol_fused.hip -- the contract-tolerance ("fused arithmetic") instances of the plane-sweep warp + variance build.
//
// Same kernels as costvol.hip (costvol_kernels.h) with AR = AR_FUSED: float64 geometry, float32 tap coordinates and the
// LDS staging are unchanged; the variance is taken of the differences to the ref feature with its constant factors folded
// into the tap weights (smvs_device.h, "fused arithmetic": 11 instead of 22 packed operations per plane and channel pair
// at 3 views).  Replaces the same reference lines as costvol.hip -- /root/reference/networks/casred.py:22-53, :191-212,
// /root/reference/modules/warping.py:310-365 -- within the stated tolerance instead of bit for bit.  A translation unit of
// its own so that the two families of instances compile in parallel.
#include "costvol_kernels.h"

namespace smvs {

hipError_t launch_costvol_fused(int geo_kind, const CostVolParams& p, hipStream_t st)
{
    return geo_kind == 0 ? launch_nsrc<0, AR_FUSED>(p, st) : launch_nsrc<1, AR_FUSED>(p, st);
}

}  // namespace smvs
