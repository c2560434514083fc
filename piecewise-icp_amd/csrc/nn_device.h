// Device-side exact 1-NN search over a GridDesc (shared by the search, ICP and VCM kernels).
// See grid.hip for the exactness argument.
#pragma once

#include <hip/hip_runtime.h>

#include "common.h"

__device__ __forceinline__ int cell_of(float p, float o, float inv_h) {
    float t = floorf((p - o) * inv_h);
    // keep far-away / non-finite queries inside int range
    t = fminf(fmaxf(t, -1.0e6f), 1.0e6f);
    return (t == t) ? (int)t : 0;
}

// ---- exact 1-NN ------------------------------------------------------------------------------------
struct NNBest {
    float d2;
    int idx;
};

__device__ __forceinline__ void scan_points(const float4* __restrict__ pts, int lo, int hi, float qx, float qy,
                                            float qz, NNBest& b) {
    for (int j = lo; j < hi; ++j) {
        float4 p = pts[j];
        float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
        float d2 = dx * dx;     // flann::L2_Simple<float>: result += diff*diff, x then y then z
        d2 = d2 + dy * dy;
        d2 = d2 + dz * dz;
        int id = __float_as_int(p.w);
        if (d2 < b.d2 || (d2 == b.d2 && id < b.idx)) { b.d2 = d2; b.idx = id; }
    }
}

// points of the x-range [x0,x1] (clipped) of row (y,z); returns #points scanned
__device__ __forceinline__ int scan_row(const GridDesc& g, int y, int z, int x0, int x1, float qx, float qy,
                                        float qz, NNBest& b) {
    if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) return 0;
    x0 = max(x0, 0);
    x1 = min(x1, g.nx - 1);
    if (x0 > x1) return 0;
    int row = (z * g.ny + y) * g.nx;
    int lo = g.cell_start[row + x0], hi = g.cell_start[row + x1 + 1];
    scan_points(g.pts, lo, hi, qx, qy, qz, b);
    return hi - lo;
}

__device__ __forceinline__ NNBest nn_query(const GridDesc& g, float qx, float qy, float qz, unsigned& examined) {
    NNBest b;
    b.d2 = INFINITY;
    b.idx = 0x7fffffff;
    if (g.n <= 0) { b.idx = -1; return b; }
    int cx = cell_of(qx, g.ox, g.inv_h), cy = cell_of(qy, g.oy, g.inv_h), cz = cell_of(qz, g.oz, g.inv_h);
    // Chebyshev distance (in cells) from the query cell to the grid box, and the ring that covers it all
    int ex = max(0, max(-cx, cx - (g.nx - 1)));
    int ey = max(0, max(-cy, cy - (g.ny - 1)));
    int ez = max(0, max(-cz, cz - (g.nz - 1)));
    int r = max(max(ex, ey), max(ez, 1));
    int rcover = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));
    unsigned cnt = 0;
    // full block of radius r
    for (int dz = -r; dz <= r; ++dz)
        for (int dy = -r; dy <= r; ++dy) cnt += scan_row(g, cy + dy, cz + dz, cx - r, cx + r, qx, qy, qz, b);
    for (;;) {
        float bound = (float)r * g.h - 2.0f * g.slack;
        if (b.idx != 0x7fffffff && bound > 0.0f && b.d2 < bound * bound * 0.99999f) break;
        if (r >= rcover) break;
        ++r;
        for (int dz = -r; dz <= r; ++dz)
            for (int dy = -r; dy <= r; ++dy) {
                if (dz == -r || dz == r || dy == -r || dy == r) {
                    cnt += scan_row(g, cy + dy, cz + dz, cx - r, cx + r, qx, qy, qz, b);
                } else {
                    cnt += scan_row(g, cy + dy, cz + dz, cx - r, cx - r, qx, qy, qz, b);
                    cnt += scan_row(g, cy + dy, cz + dz, cx + r, cx + r, qx, qy, qz, b);
                }
            }
    }
    if (b.idx == 0x7fffffff) b.idx = -1;
    examined = cnt;
    return b;
}

__device__ __forceinline__ void add_examined(unsigned long long* ctr, unsigned cnt) {
    if (!ctr) return;
    unsigned long long c = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(ctr, c);
}

