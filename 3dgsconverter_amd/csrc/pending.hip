// pending.hip -- entry points declared in include/gsx_hip.h whose kernels are not written yet.
#include "gsx_common.h"

extern "C" {

int gsx_density_voxels(const float *, const float *, const float *, int64_t, int64_t, double, int64_t, int64_t,
                       int64_t *, int64_t *, int64_t *, int64_t *)
{
    GSX_FAIL("gsx_density_voxels: not implemented yet");
}

int gsx_density_mask(const float *, const float *, const float *, int64_t, int64_t, double, const int64_t *, int64_t,
                     uint8_t *)
{
    GSX_FAIL("gsx_density_mask: not implemented yet");
}

int gsx_kmeans_lloyd(const float *, int64_t, int, int, int, const float *, float *, int32_t *)
{
    GSX_FAIL("gsx_kmeans_lloyd: not implemented yet");
}

int gsx_quantize_sorted_codebook(const float *, int64_t, const float *, int, uint8_t *)
{
    GSX_FAIL("gsx_quantize_sorted_codebook: not implemented yet");
}
}
