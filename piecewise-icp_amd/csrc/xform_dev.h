// The bulk-transform role of an outer iteration's update (R.cpp:943-954): cloud2 <- T * cloud2 and the tight bounding box of the
// moved cloud folded into the iteration's slot by the block that finishes last.  Shared by k_transform_all (loop.hip) and the
// merged transform + front launch (patch.hip).
#pragma once
#include "common.h"
#include "devmath.h"

struct Mat4 {
    float m[16];
};

namespace pwdev {

constexpr int kBoxParts = 64;
constexpr int kXfBlock = 256;             // block size of k_transform_all / k_xf_front (k_xf_vcm: 1024)

__device__ __forceinline__ unsigned f2ord_dev(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// centroids + boundary points and the patch points: block `bid` of `nb` blocks of BLOCK threads
template <int BLOCK = kXfBlock>
__device__ __forceinline__ void xf_rest_block(const Mat4& T, const float4* ctbp_in, float4* ctbp, int n_ctbp, const float4* pat_in,
                                              float4* pat, int n_pat, int bid, int nb) {
    const int stride = nb * BLOCK, ntot = n_ctbp + n_pat;
    for (int i = bid * BLOCK + (int)threadIdx.x; i < ntot; i += 4 * stride) {
        float4* q[4];
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = i + u * stride;
            q[u] = (j < n_ctbp) ? (ctbp + j) : (pat + (j - n_ctbp));
            if (j < ntot) v[u] = (j < n_ctbp) ? ctbp_in[j] : pat_in[j - n_ctbp];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u * stride < ntot) *q[u] = xform_point(T.m, v[u]);
    }
}

// block `bid` of `nb_cloud` blocks of BLOCK threads; `sh`: BLOCK / 64 x 6 floats of LDS
template <int BLOCK = kXfBlock>
__device__ __forceinline__ void xf_cloud_block(const Mat4& T, const float4* cloud_in, float4* cloud, int n, int bid, int nb_cloud,
                                               unsigned* __restrict__ bbox_part, unsigned* __restrict__ slot, float (*sh)[6]) {
    constexpr int kBlock = BLOCK;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    {
        const int stride = nb_cloud * kBlock;
        for (int i = bid * kBlock + threadIdx.x; i < n; i += 4 * stride) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * stride < n) v[u] = cloud_in[i + u * stride];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * stride < n) {
                    const float4 w = xform_point(T.m, v[u]);
                    cloud[i + u * stride] = w;
                    mn[0] = fminf(mn[0], w.x); mx[0] = fmaxf(mx[0], w.x);
                    mn[1] = fminf(mn[1], w.y); mx[1] = fmaxf(mx[1], w.y);
                    mn[2] = fminf(mn[2], w.z); mx[2] = fmaxf(mx[2], w.z);
                }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            mn[d] = fminf(mn[d], __shfl_xor(mn[d], o));
            mx[d] = fmaxf(mx[d], __shfl_xor(mx[d], o));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int d = 0; d < 3; ++d) { sh[wave][d] = mn[d]; sh[wave][3 + d] = mx[d]; }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    unsigned* part = bbox_part + (bid & (kBoxParts - 1)) * 32;
    if (threadIdx.x < 3) {
        float a = sh[0][threadIdx.x], b = sh[0][3 + threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) { a = fminf(a, sh[w][threadIdx.x]); b = fmaxf(b, sh[w][3 + threadIdx.x]); }
        // 64 partial boxes, one 128-byte line each: same-line atomics from all XCDs serialise (~5 ns apiece)
        const unsigned r0 = atomicMin(&part[threadIdx.x], f2ord_dev(a));
        const unsigned r1 = atomicMax(&part[3 + threadIdx.x], f2ord_dev(b));
        asm volatile("" ::"v"(r0), "v"(r1));      // wait until both have been PERFORMED (see below)
    }
    // Fold in the same launch: the block that completes a partial box counts it, the block that completes the last
    // partial box folds all of them into the slot and re-arms the buffer (two levels, so that no counter line sees
    // more than ~nb_cloud/64 + 64 atomics).  The box atomics have to be PERFORMED before the count: they are
    // device-coherent read-modify-writes, so waiting for their return values (above) is enough.  A device-scope fence
    // would also do, but it writes the L2 back, which is ruinous in a launch that has just written the whole cloud;
    // a mere acknowledgement wait is NOT enough (the update may still be on its way to the coherence point).
    unsigned last = 0;
    if (threadIdx.x == 0) {
        const int pidx = bid & (kBoxParts - 1);
        const unsigned np = (unsigned)((nb_cloud - pidx + kBoxParts - 1) / kBoxParts);
        if (atomicAdd(&part[8], 1u) == np - 1u) {
            const unsigned nparts = (unsigned)min(nb_cloud, kBoxParts);
            if (atomicAdd(&bbox_part[kBoxParts * 32], 1u) == nparts - 1u) last = 1u;
        }
    }
    last = (unsigned)__shfl((int)last, 0);
    if (!last) return;
    const int t = threadIdx.x;
    unsigned umn[3], umx[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        umn[d] = __hip_atomic_load(&bbox_part[t * 32 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        umx[d] = __hip_atomic_load(&bbox_part[t * 32 + 3 + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            umn[d] = min(umn[d], (unsigned)__shfl_xor((int)umn[d], o));
            umx[d] = max(umx[d], (unsigned)__shfl_xor((int)umx[d], o));
        }
#pragma unroll
    for (int d = 0; d < 3; ++d) { bbox_part[t * 32 + d] = 0xffffffffu; bbox_part[t * 32 + 3 + d] = 0u; }
    bbox_part[t * 32 + 8] = 0u;
    if (t == 0) {
        bbox_part[kBoxParts * 32] = 0u;
#pragma unroll
        for (int d = 0; d < 3; ++d) { slot[4 + d] = umn[d]; slot[7 + d] = umx[d]; }
    }
}

}  // namespace pwdev
