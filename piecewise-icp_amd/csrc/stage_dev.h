// Scalar pieces of the reference's stage logic that both the host loop and the ICP tail evaluate - one source for both, so the
// device's decision and the host's record cannot drift apart:
//   pw_octree_bbox        pcl::octree::OctreePointCloud::defineBoundingBox + getKeyBitSize from the tight float
//                         min / max of the cloud; resolution = double(Res2 * 2)  (R.cpp:881-886)
//   pw_bb_corner_change   calBoundingBoxCornerChange (C.cpp:410-419)
// and the guard of a speculative update: while the schedule is still in Stage 1 the host cannot know, before it has seen an
// iteration's transformation, whether the iteration switches to Stage 2 (R.cpp:891-894: maxBBchange < LoD_min) - in which case
// the update and the next front can start at once - or needs the dense search first.  The ICP tail that converges evaluates
// that comparison itself and leaves (maxBBchange, flag) in two slot words; the update launch that was enqueued behind the batch
// runs only if the flag is set.  The host takes the same two words from the mailbox message.
#pragma once
#include <cfloat>
#include <cmath>

#include <hip/hip_runtime.h>

struct StageGuard {
    const unsigned* bbox6 = nullptr;   // min x y z | max x y z of the source cloud as the previous update left it (order-preserving
                                       // unsigned encoding of floats, xform_dev.h: f2ord_dev)
    unsigned* out = nullptr;           // [0] <- float bits of maxBBchange, [1] <- 1 if maxBBchange < DTmin else 0; nullptr: not armed
    double resolution = 0.0;           // double(Res2 * 2)
    float DTmin = 0.f;
};

__host__ __device__ inline float pw_ord2f(unsigned u) {
    const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &b, 4);
    return f;
}

__host__ __device__ inline void pw_octree_bbox(const float* mn, const float* mx, double resolution, double* bb) {
    const float minValue = FLT_EPSILON * 512.0f;
    const float eps = FLT_EPSILON;
    double lo[3], hi[3];
    for (int d = 0; d < 3; ++d) { lo[d] = mn[d]; hi[d] = (double)(mx[d] + minValue); }
    unsigned mk = 2;
    for (int d = 0; d < 3; ++d) {
        const unsigned k = (unsigned)ceil((hi[d] - lo[d] - eps) / resolution);
        if (k > mk) mk = k;
    }
    // (log only decides an integer here: ceil(log2(mk) - eps); a last-bit difference between two log implementations cannot
    // move it, log2 of an integer is never that close to an integer + eps)
    unsigned depth = (unsigned)ceil(log((double)mk) / log(2.0) - eps);
    if (depth > 32) depth = 32;
    const double side = (double)(1u << depth) * resolution;
    for (int d = 0; d < 3; ++d) {
        const double over = (side - (hi[d] - lo[d])) / 2.0;
        if (over > eps) { lo[d] -= over; hi[d] += over; }
    }
    bb[0] = lo[0]; bb[1] = lo[1]; bb[2] = lo[2]; bb[3] = hi[0]; bb[4] = hi[1]; bb[5] = hi[2];
}

__host__ __device__ inline float pw_bb_corner_change(const double* bb, const float* T) {
    float r = 0.f;
    for (int k = 0; k < 2; ++k) {
        const float c[3] = {(float)bb[3 * k], (float)bb[3 * k + 1], (float)bb[3 * k + 2]};
        float t[3];
        for (int i = 0; i < 3; ++i) t[i] = T[4 * i] * c[0] + T[4 * i + 1] * c[1] + T[4 * i + 2] * c[2] + T[4 * i + 3] * 1.0f;
        const float dx = t[0] - c[0], dy = t[1] - c[1], dz = t[2] - c[2];
        const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
        if (nrm > r) r = nrm;
    }
    return r;
}
