// ct_raygen.h -- primary-ray generation shared by the megakernel (ct_kernels.hip) and the
// wavefront pipeline (ct_pipeline.hip).
#ifndef CT_RAYGEN_H
#define CT_RAYGEN_H

#include "ct_params.h"

namespace ctd {

__device__ __forceinline__ d3 primary_direction(const FrameParams &P, float px, float py) {
    // CanvasToViewport (raythread.cpp:186-194) then v * rotation (mymath.h:68-75)
    const double vx = (double)px * (double)P.inv_height, vy = (double)py * (double)P.inv_height, vz = (double)1.0f;
    return make_d3(vx * P.rot[0] + vy * P.rot[3] + vz * P.rot[6], vx * P.rot[1] + vy * P.rot[4] + vz * P.rot[7],
                   vx * P.rot[2] + vy * P.rot[5] + vz * P.rot[8]);
}

__device__ __forceinline__ int32_t jitter_int(const FrameParams &P, uint32_t xi, uint32_t yi, uint32_t sample,
                                              uint32_t axis) {
    if (P.rng_mode == CTGPU_RNG_LIBC) {
        const uint64_t row = (uint64_t)P.row_ordinal[yi];
        const uint64_t k = ((uint64_t)xi * P.traced_rows + row) * 32u + sample * 2u + axis;
        return P.rand_table[k];
    }
    const uint64_t k = ((uint64_t)xi * (uint64_t)P.height + yi) * 32u + sample * 2u + axis;
    return hash_rand(P.rng_seed, k);
}


}  // namespace ctd
#endif
