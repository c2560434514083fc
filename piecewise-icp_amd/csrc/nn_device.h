// Device-side exact 1-NN search over a two-level uniform grid (shared by the search, ICP and VCM kernels).
//
// Result = argmin over ALL target points of the float expression ((dx*dx) + dy*dy) + dz*dz
// (flann::L2_Simple<float>), ties to the lowest index.  Three stages, each exact on its own terms:
//   1. the 27-cell stencil of the FINE level (9 contiguous row ranges).  Done if the best float d2 is provably
//      below that of every point outside the stencil (conservative bound incl. cell-assignment rounding slack).
//   2. otherwise, with a candidate at distance rho: every closer point lies in the cube [q-rho, q+rho]; scan
//      exactly the COARSE cells that cube touches (a handful of row ranges) — no ring-by-ring growth.
//   3. with no candidate (empty stencil) or an oversized cube: block/shell expansion on the coarse level until
//      the same kind of bound holds or the whole grid has been scanned.
//
// Memory-level parallelism: the per-query work is a chain of dependent loads (cell_start -> points), so rows
// are processed in batches — all begin/end loads of a batch are issued before the first point load, and the
// point loop is unrolled by two with both loads up front.
#pragma once

#include <hip/hip_runtime.h>

#include "common.h"

__device__ __forceinline__ int cell_of(float p, float o, float inv_h) {
    float t = floorf((p - o) * inv_h);
    // keep far-away / non-finite queries inside int range
    t = fminf(fmaxf(t, -1.0e6f), 1.0e6f);
    return (t == t) ? (int)t : 0;
}

// Best candidate as ONE 64-bit key: (float bits of d2) << 32 | index.  d2 >= +0, so the unsigned order of the
// key is the lexicographic order (d2, index): a single unsigned min implements "smaller distance, ties to the
// lowest index" without branches.  NaN distances (bits > +inf) can never win, as with `d2 < best`.
struct NNBest {
    unsigned long long key;
    __device__ __forceinline__ float d2() const { return __uint_as_float((unsigned)(key >> 32)); }
    __device__ __forceinline__ int idx() const { return (int)(unsigned)(key & 0xffffffffull); }
    __device__ __forceinline__ bool found() const { return (unsigned)(key & 0xffffffffull) != 0x7fffffffu; }
};

constexpr int kNoIdx = 0x7fffffff;
constexpr unsigned long long kKeyInit = (0x7f800000ull << 32) | 0x7fffffffull;    // (+inf, no index)

__device__ __forceinline__ void nn_consider(const float4 p, float qx, float qy, float qz, NNBest& b) {
    const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
    float d2 = dx * dx;     // flann::L2_Simple<float>: result += diff*diff, x then y then z
    d2 = d2 + dy * dy;
    d2 = d2 + dz * dz;
    const unsigned long long k = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)__float_as_uint(p.w);
    b.key = (k < b.key) ? k : b.key;
}

__device__ __forceinline__ void scan_points(const float4* __restrict__ pts, int lo, int hi, float qx, float qy,
                                            float qz, NNBest& b) {
    int j = lo;
    for (; j + 1 < hi; j += 2) {
        const float4 p0 = pts[j], p1 = pts[j + 1];
        nn_consider(p0, qx, qy, qz, b);
        nn_consider(p1, qx, qy, qz, b);
    }
    if (j < hi) nn_consider(pts[j], qx, qy, qz, b);
}

// lane `sub` of a group of `stride` lanes takes every stride-th point (cooperative far path)
__device__ __forceinline__ void scan_points_strided(const float4* __restrict__ pts, int lo, int hi, int sub, int stride,
                                                    float qx, float qy, float qz, NNBest& b) {
    for (int j = lo + sub; j < hi; j += stride) nn_consider(pts[j], qx, qy, qz, b);
}

// [begin,end) of the points of cells x0..x1 (clipped) of row (y,z); empty if the row is outside the grid
__device__ __forceinline__ void row_range(const GridLevel& g, int y, int z, int x0, int x1, int& lo, int& hi) {
    lo = 0; hi = 0;
    if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) return;
    x0 = max(x0, 0);
    x1 = min(x1, g.nx - 1);
    if (x0 > x1) return;
    const int row = (z * g.ny + y) * g.nx;
    lo = g.cell_start[row + x0];
    hi = g.cell_start[row + x1 + 1];
}

// all rows (y in [y0,y1], z in [z0,z1]) with the x-range [x0,x1]
__device__ __forceinline__ unsigned scan_box(const GridLevel& g, int x0, int x1, int y0, int y1, int z0, int z1,
                                             float qx, float qy, float qz, NNBest& b) {
    unsigned cnt = 0;
    const int wy = y1 - y0 + 1;
    const int nrows = wy * (z1 - z0 + 1);
    for (int t0 = 0; t0 < nrows; t0 += 4) {
        int lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = t0 + k;
            lo[k] = hi[k] = 0;
            if (t < nrows) row_range(g, y0 + t % wy, z0 + t / wy, x0, x1, lo[k], hi[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            scan_points(g.pts, lo[k], hi[k], qx, qy, qz, b);
            cnt += (unsigned)(hi[k] - lo[k]);
        }
    }
    return cnt;
}

// scan_box shared by a group of `stride` lanes: same rows for every lane, points interleaved
__device__ __forceinline__ unsigned scan_box_coop(const GridLevel& g, int x0, int x1, int y0, int y1, int z0, int z1,
                                                  int sub, int stride, float qx, float qy, float qz, NNBest& b) {
    unsigned cnt = 0;
    const int wy = y1 - y0 + 1;
    const int nrows = wy * (z1 - z0 + 1);
    for (int t0 = 0; t0 < nrows; t0 += 4) {
        int lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = t0 + k;
            lo[k] = hi[k] = 0;
            if (t < nrows) row_range(g, y0 + t % wy, z0 + t / wy, x0, x1, lo[k], hi[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            scan_points_strided(g.pts, lo[k], hi[k], sub, stride, qx, qy, qz, b);
            if (sub == 0) cnt += (unsigned)(hi[k] - lo[k]);
        }
    }
    return cnt;
}

// the shell of Chebyshev radius r around (cx,cy,cz)
__device__ __forceinline__ unsigned scan_shell(const GridLevel& g, int cx, int cy, int cz, int r, float qx, float qy,
                                               float qz, NNBest& b) {
    unsigned cnt = 0;
    const int w = 2 * r + 1;
    const int nrows = w * w;
    for (int t0 = 0; t0 < nrows; t0 += 4) {
        int lo[8], hi[8];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = t0 + k;
            lo[2 * k] = hi[2 * k] = lo[2 * k + 1] = hi[2 * k + 1] = 0;
            if (t < nrows) {
                const int dz = t / w - r, dy = t % w - r;
                if (dz == -r || dz == r || dy == -r || dy == r) {
                    row_range(g, cy + dy, cz + dz, cx - r, cx + r, lo[2 * k], hi[2 * k]);
                } else {
                    row_range(g, cy + dy, cz + dz, cx - r, cx - r, lo[2 * k], hi[2 * k]);
                    row_range(g, cy + dy, cz + dz, cx + r, cx + r, lo[2 * k + 1], hi[2 * k + 1]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            scan_points(g.pts, lo[k], hi[k], qx, qy, qz, b);
            cnt += (unsigned)(hi[k] - lo[k]);
        }
    }
    return cnt;
}

// true when every point outside the scanned block of Chebyshev radius r has a larger float d2 than b
__device__ __forceinline__ bool nn_resolved(const GridLevel& g, int r, const NNBest& b) {
    const float bound = (float)r * g.h - 2.0f * g.slack;
    return b.found() && bound > 0.0f && b.d2() < bound * bound * 0.99999f;
}

// stage 3: block of radius r0 (where the grid starts) then shell by shell, on one level
__device__ __forceinline__ unsigned nn_expand(const GridLevel& g, float qx, float qy, float qz, NNBest& b) {
    const int cx = cell_of(qx, g.ox, g.inv_h), cy = cell_of(qy, g.oy, g.inv_hy), cz = cell_of(qz, g.oz, g.inv_hz);
    const int ex = max(0, max(-cx, cx - (g.nx - 1)));
    const int ey = max(0, max(-cy, cy - (g.ny - 1)));
    const int ez = max(0, max(-cz, cz - (g.nz - 1)));
    int r = max(max(ex, ey), max(ez, 1));
    const int rcover = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));
    unsigned cnt = scan_box(g, cx - r, cx + r, cy - r, cy + r, cz - r, cz + r, qx, qy, qz, b);
    for (;;) {
        if (nn_resolved(g, r, b)) break;
        if (r >= rcover) break;
        ++r;
        cnt += scan_shell(g, cx, cy, cz, r, qx, qy, qz, b);
    }
    return cnt;
}

// U loads in flight per pass.  The searches are bound by memory latency (SQ_WAIT_ANY ~78 % of the wave cycles of the
// dense kernel with two loads in flight), not by issue: a row of a column grid holds ~27 points, so U = 2 means ~14
// dependent round trips per row and U = 8 four.
template <int U>
__device__ __forceinline__ void scan_points_u(const float4* __restrict__ pts, int lo, int hi, float qx, float qy, float qz,
                                              NNBest& b) {
    for (int j = lo; j < hi; j += U) {
        float4 p[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j + u < hi) p[u] = pts[j + u];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j + u < hi) nn_consider(p[u], qx, qy, qz, b);
    }
}

#ifndef PWICP_STAGE1_UNROLL
#define PWICP_STAGE1_UNROLL 2
#endif

// stage 1: fine 27-cell stencil; returns true when the result is final
__device__ __forceinline__ bool nn_stage1(const GridDesc& gd, float qx, float qy, float qz, NNBest& b, unsigned& cnt) {
    const GridLevel& g = gd.fine;
    const int cx = cell_of(qx, g.ox, g.inv_h), cy = cell_of(qy, g.oy, g.inv_hy), cz = cell_of(qz, g.oz, g.inv_hz);
    int lo[9], hi[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) row_range(g, cy + (k % 3) - 1, cz + (k / 3) - 1, cx - 1, cx + 1, lo[k], hi[k]);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        scan_points_u<PWICP_STAGE1_UNROLL>(g.pts, lo[k], hi[k], qx, qy, qz, b);
        cnt += (unsigned)(hi[k] - lo[k]);
    }
    return nn_resolved(g, 1, b);
}

// stages 2 and 3 on the coarse level, seeded with the best candidate so far (if any)
__device__ __forceinline__ void nn_stage23(const GridDesc& gd, float qx, float qy, float qz, NNBest& b, unsigned& cnt) {
    const GridLevel& c = gd.coarse;
    if (b.found()) {
        // candidate known -> scan the coarse cells touching the cube [q - rho, q + rho]
        const float rho = sqrtf(b.d2()) * 1.00001f + 2.0f * c.slack;
        const int x0 = max(cell_of(qx - rho, c.ox, c.inv_h), 0), x1 = min(cell_of(qx + rho, c.ox, c.inv_h), c.nx - 1);
        const int y0 = max(cell_of(qy - rho, c.oy, c.inv_hy), 0), y1 = min(cell_of(qy + rho, c.oy, c.inv_hy), c.ny - 1);
        const int z0 = max(cell_of(qz - rho, c.oz, c.inv_hz), 0), z1 = min(cell_of(qz + rho, c.oz, c.inv_hz), c.nz - 1);
        if ((y1 - y0 + 1) * (z1 - z0 + 1) <= 64) {
            if (x0 <= x1 && y0 <= y1 && z0 <= z1) cnt += scan_box(c, x0, x1, y0, y1, z0, z1, qx, qy, qz, b);
            return;
        }
    }
    cnt += nn_expand(c, qx, qy, qz, b);       // block / shell expansion on the coarse level
}

__device__ __forceinline__ NNBest nn_query(const GridDesc& gd, float qx, float qy, float qz, unsigned& examined) {
    NNBest b;
    b.key = kKeyInit;
    examined = 0;
    if (gd.fine.n <= 0) return b;
    unsigned cnt = 0;
    if (!nn_stage1(gd, qx, qy, qz, b, cnt)) nn_stage23(gd, qx, qy, qz, b, cnt);
    examined = cnt;
    return b;
}

// ---- group-cooperative search: kGroup consecutive lanes share ONE query ------------------------------------------------
// For launches with few queries (centroids, boundary points) the cost of a query is its chain of dependent memory
// round trips (begin/end words -> points, row after row), not throughput.  Spreading the rows of a query over the
// lanes of a group shortens that chain ~5x.  All decisions are taken on group-uniform values (after a min over the
// group), so the lanes of a group never diverge; the result is the same exact (d2, index) minimum.
constexpr int kGroup = 8;

template <int G = kGroup>
__device__ __forceinline__ void group_min(NNBest& b) {
#pragma unroll
    for (int o = 1; o < G; o <<= 1) {
        const unsigned long long other = __shfl_xor(b.key, o);
        b.key = other < b.key ? other : b.key;
    }
}

// up to four loads in flight per pass: a short range costs ONE memory round trip instead of one per two points
__device__ __forceinline__ unsigned scan_points4(const float4* __restrict__ pts, int lo, int hi, int step, float qx, float qy,
                                                 float qz, NNBest& b) {
    unsigned cnt = 0;
    for (int j = lo; j < hi; j += 4 * step) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j + u * step < hi) p[u] = pts[j + u * step];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j + u * step < hi) { nn_consider(p[u], qx, qy, qz, b); ++cnt; }
    }
    return cnt;
}

// rows (y in [y0,y1], z in [z0,z1]) x [x0,x1], shared by the group: the lanes fetch the begin/end words of eight rows
// at once (one row each), hand them round with shuffles, and then walk every row TOGETHER, points interleaved over the
// lanes, four loads in flight each.  The chain is 1 + sum_rows ceil(n_row / 32) round trips, whatever the shape of
// the box (the coarse boxes of stage 2 have few, long rows).  (Every lane walking its OWN row when the rows are short - the
// boxes of stage 2 behind a stencil candidate are ~4 rows of ~5 points, tools/qstat_front.py - shortens the chain on paper and
// lengthens the front launches by 2-3 us: four times the distinct lines per load instruction.)
template <int G = kGroup>
__device__ __forceinline__ unsigned scan_box_group(const GridLevel& g, int x0, int x1, int y0, int y1, int z0, int z1, int sub,
                                                   float qx, float qy, float qz, NNBest& b) {
    unsigned cnt = 0;
    const int wy = y1 - y0 + 1;
    const int nrows = wy * (z1 - z0 + 1);
    const int gbase = (int)(__lane_id() & ~(unsigned)(G - 1));
    for (int t0 = 0; t0 < nrows; t0 += G) {
        int lo_s = 0, hi_s = 0;
        const int t = t0 + sub;
        if (t < nrows) row_range(g, y0 + t % wy, z0 + t / wy, x0, x1, lo_s, hi_s);
#pragma unroll
        for (int k = 0; k < G; ++k) {
            const int lo = __shfl(lo_s, gbase + k), hi = __shfl(hi_s, gbase + k);
            cnt += scan_points4(g.pts, lo + sub, hi, G, qx, qy, qz, b);
        }
    }
    return cnt;
}

// G: lanes per query (a power of two <= 8; sub = lane % G).  8 for the launches with ~10^4 queries (ICP, VCM: the chain of
// round trips is everything), 4 where 10^5 queries share the chip with other work (the front launches: half the waves and
// about 0.6x the instructions per query, the chain as long - a lane's two or three rows are requested together).
#ifdef PWICP_QSTAT
static __device__ unsigned long long pw_qstat[32];
#define QS_ADD(i_, v_) do { if (sub == 0) atomicAdd(&pw_qstat[i_], (unsigned long long)(v_)); } while (0)
#else
#define QS_ADD(i_, v_) do { } while (0)
#endif
template <int G = kGroup>
__device__ __forceinline__ NNBest nn_query_group(const GridDesc& gd, float qx, float qy, float qz, int sub) {
    NNBest b;
    b.key = kKeyInit;
    const GridLevel& g = gd.fine;
    if (g.n <= 0) return b;
    // stage 1: the 9 stencil rows dealt over the G lanes (row r to lane r % G)
    {
        const int cx = cell_of(qx, g.ox, g.inv_h), cy = cell_of(qy, g.oy, g.inv_hy), cz = cell_of(qz, g.oz, g.inv_hz);
        constexpr int R = (9 + G - 1) / G;
        int lo[R], hi[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const int r = sub + i * G;
            lo[i] = hi[i] = 0;
            if (r < 9) row_range(g, cy + (r % 3) - 1, cz + (r / 3) - 1, cx - 1, cx + 1, lo[i], hi[i]);
        }
#pragma unroll
        for (int i = 0; i < R; ++i) scan_points4(g.pts, lo[i], hi[i], 1, qx, qy, qz, b);
        group_min<G>(b);
        if (nn_resolved(g, 1, b)) { QS_ADD(0, 1); return b; }
    }
    const GridLevel& c = gd.coarse;
    // stage 2: candidate known -> coarse cells touching the cube [q - rho, q + rho]
    if (b.found()) {
        const float rho = sqrtf(b.d2()) * 1.00001f + 2.0f * c.slack;
        const int x0 = max(cell_of(qx - rho, c.ox, c.inv_h), 0), x1 = min(cell_of(qx + rho, c.ox, c.inv_h), c.nx - 1);
        const int y0 = max(cell_of(qy - rho, c.oy, c.inv_hy), 0), y1 = min(cell_of(qy + rho, c.oy, c.inv_hy), c.ny - 1);
        const int z0 = max(cell_of(qz - rho, c.oz, c.inv_hz), 0), z1 = min(cell_of(qz + rho, c.oz, c.inv_hz), c.nz - 1);
        if ((y1 - y0 + 1) * (z1 - z0 + 1) <= 64) {
            unsigned qs_n = 0;
            if (x0 <= x1 && y0 <= y1 && z0 <= z1) qs_n = scan_box_group<G>(c, x0, x1, y0, y1, z0, z1, sub, qx, qy, qz, b);
            group_min<G>(b);
            (void)qs_n;
            QS_ADD(1, 1); QS_ADD(8, (y1 - y0 + 1) * (z1 - z0 + 1)); QS_ADD(9, qs_n);
            return b;
        }
    }
    // stage 3: block, then shells, on the coarse level; a group-min after every block/shell keeps the stop test uniform
    {
        const int cx = cell_of(qx, c.ox, c.inv_h), cy = cell_of(qy, c.oy, c.inv_hy), cz = cell_of(qz, c.oz, c.inv_hz);
        const int ex = max(0, max(-cx, cx - (c.nx - 1)));
        const int ey = max(0, max(-cy, cy - (c.ny - 1)));
        const int ez = max(0, max(-cz, cz - (c.nz - 1)));
        int r = max(max(ex, ey), max(ez, 1));
        const int rcover = max(max(max(cx, c.nx - 1 - cx), max(cy, c.ny - 1 - cy)), max(cz, c.nz - 1 - cz));
        scan_box_group<G>(c, cx - r, cx + r, cy - r, cy + r, cz - r, cz + r, sub, qx, qy, qz, b);
        group_min<G>(b);
        while (!nn_resolved(c, r, b) && r < rcover) {
            ++r;
            // shell of radius r: rows dealt round-robin; inner rows contribute their two end cells
            const int w = 2 * r + 1;
            for (int t = sub; t < w * w; t += G) {
                const int dz = t / w - r, dy = t % w - r;
                int lo0, hi0, lo1 = 0, hi1 = 0;
                if (dz == -r || dz == r || dy == -r || dy == r) {
                    row_range(c, cy + dy, cz + dz, cx - r, cx + r, lo0, hi0);
                } else {
                    row_range(c, cy + dy, cz + dz, cx - r, cx - r, lo0, hi0);
                    row_range(c, cy + dy, cz + dz, cx + r, cx + r, lo1, hi1);
                }
                scan_points4(c.pts, lo0, hi0, 1, qx, qy, qz, b);
                scan_points4(c.pts, lo1, hi1, 1, qx, qy, qz, b);
            }
            group_min<G>(b);
        }
    }
    QS_ADD(2, 1);
    return b;
}

// ---- distance-only search with disc pruning (dense cloud-to-cloud queries) ---------------------------------------------
// The dense Stage-1 launch (calPercentileDistBetween2PC, C.cpp:266-281) needs min d2 only, not the index: the running
// best is a float and one v_min_f32 replaces the 64-bit (d2, index) compare.  Instead of a fixed 27-cell stencil the
// search takes a first candidate from the query's own cell-row segment and then scans exactly the cells a disc of that
// radius touches, row by row (rows whose distance to the query already exceeds the candidate are never visited, and
// the x-range of a row shrinks with its distance: cells of the circle, not of its bounding box).
// Exactness: every target point closer than the candidate lies inside the ball [q - rho, q + rho]; the float cell
// assignment is monotone in the coordinate, and `rho` carries the rounding slack of the cell boundaries (2 * slack, as
// in nn_stage23), so every cell that can hold such a point is visited.  The result is the exact minimum of the float
// expression ((dx*dx)+dy*dy)+dz*dz over ALL targets.
// PERM: axis roles of the level (GridLevel::perm).  The point and the query come in the level's order; the squared
// distance is ALWAYS ((dx*dx)+dy*dy)+dz*dz over the original axes: level order (y, z, x) -> original x is the third
// component, level order (z, x, y) -> original x is the second.
template <int PERM = 0>
__device__ __forceinline__ void nn_consider_d2(const float4 p, float qx, float qy, float qz, float& best) {
    const float d0 = qx - p.x, d1 = qy - p.y, d2_ = qz - p.z;
    const float dx = PERM == 0 ? d0 : (PERM == 1 ? d2_ : d1);
    const float dy = PERM == 0 ? d1 : (PERM == 1 ? d0 : d2_);
    const float dz = PERM == 0 ? d2_ : (PERM == 1 ? d1 : d0);
    float d2 = dx * dx;
    d2 = d2 + dy * dy;
    d2 = d2 + dz * dz;
    best = fminf(best, d2);          // coordinates are finite (checked at upload): d2 is never NaN
}

// ---- lean variants for the dense kernel: same exact result, fewer instructions ---------------------------------------------
// (the dense launch is bound by vector-ALU issue — ~4 cycles per wave instruction — so the search logic around the
// candidates counts as much as the candidates: no integer divisions, no correctly-rounded sqrt in the PRUNING radius
// (v_sqrt_f32 with a relative margin instead: a larger radius is always safe), cell indices without the NaN / range
// guards of cell_of (queries are finite and near the grid), four candidates per pass.)
__device__ __forceinline__ int icell(float p, float o, float inv_h) { return (int)floorf((p - o) * inv_h); }
__device__ __forceinline__ float fast_sqrt_up(float x) { return __builtin_amdgcn_sqrtf(x) * 1.0002f; }   // >= sqrt(x) for x >= 0

template <int PERM = 0>
__device__ __forceinline__ void scan_d2x4(const float4* __restrict__ pts, int lo, int hi, float qx, float qy, float qz, float& best) {
    int j = lo;
    for (; j + 4 <= hi; j += 4) {
        const float4 a = pts[j], b = pts[j + 1], c = pts[j + 2], d = pts[j + 3];
        nn_consider_d2<PERM>(a, qx, qy, qz, best);
        nn_consider_d2<PERM>(b, qx, qy, qz, best);
        nn_consider_d2<PERM>(c, qx, qy, qz, best);
        nn_consider_d2<PERM>(d, qx, qy, qz, best);
    }
    if (j + 2 <= hi) {
        const float4 a = pts[j], b = pts[j + 1];
        nn_consider_d2<PERM>(a, qx, qy, qz, best);
        nn_consider_d2<PERM>(b, qx, qy, qz, best);
        j += 2;
    }
    if (j < hi) nn_consider_d2<PERM>(pts[j], qx, qy, qz, best);
}

// the same on a level of the dense search: from the packed 12-byte copy of its points (global_load_dwordx3: 10.7 instead of 8
// points per 128-byte line - a vector-memory instruction of this search costs by the cache lines it touches: 39.6 -> 38.3 us).
// Four candidates per pass, NO tails (round 5): the minimum over ALL target points is what the search returns, so a candidate beyond
// the range's end (a real target point of the next cell) can never make the result wrong - only reading outside the array must not
// happen, and the packed copy ends in kPts3Pad far-away points.  (Measured and not kept, profiles/r05_dense_variants.txt (2): tails
// of two and one, 33.6 against 32.8 us; the points stored in pairs for v_pk_add_f32 / v_pk_mul_f32, 17 % fewer vector instructions,
// 34.8 us - the compiler packs what it can of the plain form by itself.)
constexpr int kPts3Pad = 20;           // far-away points behind the last one of a packed copy
struct PwXyz3 { float x, y, z; };
template <int PERM = 0, bool P3 = true>
__device__ __forceinline__ void scan_d2_level(const GridLevel& g, int lo, int hi, float qx, float qy, float qz, float& best) {
    if (!P3) { scan_d2x4<PERM>(g.pts, lo, hi, qx, qy, qz, best); return; }      // (a level without the packed copy)
    const PwXyz3* __restrict__ p3 = (const PwXyz3*)g.pts3;
    for (int j = lo; j < hi; j += 4) {
        const PwXyz3 a = p3[j], b = p3[j + 1], c = p3[j + 2], d = p3[j + 3];
        nn_consider_d2<PERM>(make_float4(a.x, a.y, a.z, 0.f), qx, qy, qz, best);
        nn_consider_d2<PERM>(make_float4(b.x, b.y, b.z, 0.f), qx, qy, qz, best);
        nn_consider_d2<PERM>(make_float4(c.x, c.y, c.z, 0.f), qx, qy, qz, best);
        nn_consider_d2<PERM>(make_float4(d.x, d.y, d.z, 0.f), qx, qy, qz, best);
    }
}

// scan_disc without divisions / exact square roots; rows of one z-slab are taken four at a time (all begin/end words of
// the batch in flight before the first point load).  Same contract as scan_disc.
// ROWS: also count the rows whose begin / end words are read, in bits 20.. of the result (the cost probe of
// pw_dense_level_for; the search itself uses ROWS = false).
// P3: the level carries the packed copy of its points (the small-cell levels of the dense search)
template <int PERM = 0, bool ROWS = false, bool P3 = false>
__device__ __forceinline__ unsigned scan_disc_lean(const GridLevel& g, float qx, float qy, float qz, float rho, int sy, int sz,
                                                   int sx0, int sx1, int slo, int shi, float& best) {
    unsigned cnt = 0;
    const bool gy = g.inv_hy != 0.0f, gz = g.inv_hz != 0.0f;
    const int y0 = gy ? max(icell(qy - rho, g.oy, g.inv_hy), 0) : 0, y1 = gy ? min(icell(qy + rho, g.oy, g.inv_hy), g.ny - 1) : 0;
    const int z0 = gz ? max(icell(qz - rho, g.oz, g.inv_hz), 0) : 0, z1 = gz ? min(icell(qz + rho, g.oz, g.inv_hz), g.nz - 1) : 0;
    const float rho2 = rho * rho, slack2 = 2.0f * g.slack;
    for (int z = z0; z <= z1; ++z) {
        float remz = rho2;
        if (gz) {
            const float lo = g.oz + (float)z * g.h;
            const float ez = fmaxf(fmaxf(lo - qz, qz - (lo + g.h)) - slack2, 0.0f);
            remz = rho2 - ez * ez;
            if (!(remz > 0.0f)) continue;
        }
        for (int yb = y0; yb <= y1; yb += 4) {
            int lo[4], hi[4], lo2[4], hi2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo[k] = hi[k] = lo2[k] = hi2[k] = 0;
                const int y = yb + k;
                if (y > y1) continue;
                float rem = remz;
                if (gy) {
                    const float l = g.oy + (float)y * g.h;
                    const float ey = fmaxf(fmaxf(l - qy, qy - (l + g.h)) - slack2, 0.0f);
                    rem = remz - ey * ey;
                    if (!(rem > 0.0f)) continue;
                }
                const float rx = fast_sqrt_up(rem) + slack2;
                const int x0 = max(icell(qx - rx, g.ox, g.inv_h), 0), x1 = min(icell(qx + rx, g.ox, g.inv_h), g.nx - 1);
                if (x0 > x1) continue;
                const int row = (z * g.ny + y) * g.nx;
                if (ROWS) cnt += 1u << 20;
                if (y == sy && z == sz) {
                    if (x0 < sx0) { lo[k] = g.cell_start[row + x0]; hi[k] = slo; }
                    if (x1 > sx1) { lo2[k] = shi; hi2[k] = g.cell_start[row + x1 + 1]; }
                } else {
                    lo[k] = g.cell_start[row + x0];
                    hi[k] = g.cell_start[row + x1 + 1];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                scan_d2_level<PERM, P3>(g, lo[k], hi[k], qx, qy, qz, best);
                scan_d2_level<PERM, P3>(g, lo2[k], hi2[k], qx, qy, qz, best);
                cnt += (unsigned)(hi[k] - lo[k]) + (unsigned)(hi2[k] - lo2[k]);
            }
        }
    }
    return cnt;
}

// ---- round 6: the ball on a level of COLUMNS as ONE flat loop per lane ---------------------------------------------------------
// tools/dense_slots.py (profiles/r06_dense_variants.txt): a wave of scan_disc_lean executes, per (z slab, batch of four rows, row,
// piece), a scan loop that runs to its slowest lane's count - 11.1 passes of four candidates for the ball where the lanes' OWN sums
// need 7.0 - and ~600 of its ~1 000 vector instructions are the walk around the candidates (a z loop, a y loop, four row slots with
// two pieces each), not the candidates.  On a level of columns the ball has ONE row axis and at most kDiscRowsMax rows:
//   1. every row's x-range from scan_disc_lean's own formulas (same floats, same cells), begin / end words of ALL touched rows
//      requested together (unrolled: one round trip), the own row reduced to the pieces phase A has not scanned;
//   2. the non-empty ranges written to a per-lane list in LDS (lane-major: conflict-free 8-byte accesses);
//   3. one loop over the list, four candidates per pass, no tails: the wave runs max over lanes of the lane's SUM of passes.
// Same candidates as scan_disc_lean visits, same float expression each, and a minimum does not depend on the order: the same d2, bit
// for bit.  (First tried and measured slower: the two neighbour rows' words preloaded before phase A with the other lanes left to
// scan_disc_lean, inline - 36.0 us - or compacted behind the block - 32.8 us - against 30.4; profiles/r06_dense_variants.txt.)
constexpr int kDiscRowsMax = 7;            // rows a ball of at most kMaxRhoCells (2.75) cells can touch: floor(2 * 2.75) + 2
constexpr int kDiscRangesMax = 8;          // ... plus the second piece of the own row
// x-range of row `r` of the ball (scan_disc_lean's formulas, one row axis): false = the row is not touched
__device__ __forceinline__ bool disc_row_cells(const GridLevel& g, float ux, float v, float ov, int r, float rho2, float slack2, int& x0, int& x1) {
    const float l = ov + (float)r * g.h;
    const float e = fmaxf(fmaxf(l - v, v - (l + g.h)) - slack2, 0.0f);
    const float rem = rho2 - e * e;
    if (!(rem > 0.0f)) return false;
    const float rx = fast_sqrt_up(rem) + slack2;
    x0 = max(icell(ux - rx, g.ox, g.inv_h), 0);
    x1 = min(icell(ux + rx, g.ox, g.inv_h), g.nx - 1);
    return x0 <= x1;
}
// list: kDiscRangesMax rows of `stride` int2 (lane-major), `tid` = this lane's column.  home: the row whose cells sx0 .. sx1 = points
// [slo, shi) phase A has scanned (INT_MIN: none).  Returns the number of ranges, -1 if the ball has more rows than the list holds.
__device__ __forceinline__ int disc_ranges_columns(const GridLevel& g, float ux, float uy, float uz, float rho, int home, int sx0, int sx1,
                                                   int slo, int shi, int2* __restrict__ list, int stride, int tid, unsigned& cnt) {
    const bool gy = g.inv_hy != 0.0f;
    const float v = gy ? uy : uz, ov = gy ? g.oy : g.oz, inv_hv = gy ? g.inv_hy : g.inv_hz;
    const int nv = gy ? g.ny : g.nz;
    const int v0 = max(icell(v - rho, ov, inv_hv), 0), v1 = min(icell(v + rho, ov, inv_hv), nv - 1);
    if (v1 - v0 >= kDiscRowsMax) return -1;
    const float rho2 = rho * rho, slack2 = 2.0f * g.slack;
    int lo[kDiscRangesMax], hi[kDiscRangesMax];
#pragma unroll
    for (int k = 0; k < kDiscRangesMax; ++k) lo[k] = hi[k] = 0;
#pragma unroll
    for (int k = 0; k < kDiscRowsMax; ++k) {
        const int r = v0 + k;
        int x0, x1;
        if (r <= v1 && disc_row_cells(g, ux, v, ov, r, rho2, slack2, x0, x1)) {
            const int* __restrict__ cs = g.cell_start + (size_t)r * g.nx;         // (the other axis has ONE cell: row = r * nx)
            if (r == home) {
                if (x0 < sx0) { lo[k] = cs[x0]; hi[k] = slo; }
                if (x1 > sx1) { lo[kDiscRowsMax] = shi; hi[kDiscRowsMax] = cs[x1 + 1]; }
            } else {
                lo[k] = cs[x0];
                hi[k] = cs[x1 + 1];
            }
        }
    }
    int n = 0;
#pragma unroll
    for (int k = 0; k < kDiscRangesMax; ++k)
        if (hi[k] > lo[k]) {
            list[n * stride + tid] = make_int2(lo[k], hi[k]);
            cnt += (unsigned)(hi[k] - lo[k]);
            ++n;
        }
    return n;
}
// Phase A of the dense search: the first HEAD points of a non-empty range requested at once (one round trip instead of HEAD / 4
// dependent ones - the own row segment holds 12 +- 2 points on the bench pair), the rest four per pass.  Points past the range's
// end are real target points of the next cells: harmless for a minimum over all targets (see scan_d2_level).  Measured INSIDE the
// loop (tools/inloop_dense.sh: rocprofv3 kernel trace of bench.py's steps), where the launch finds the L2 cold: HEAD = 8 / 12:
// 29.3 / 28.8 us against 29.7; replayed back to back it is worth 0.3 us and 16 points are slower.
template <int PERM, int HEAD>
__device__ __forceinline__ void scan_d2_head(const GridLevel& g, int lo, int hi, float qx, float qy, float qz, float& best) {
    if (hi <= lo) return;
    const PwXyz3* __restrict__ p3 = (const PwXyz3*)g.pts3;
    PwXyz3 v[HEAD];
#pragma unroll
    for (int k = 0; k < HEAD; ++k) v[k] = p3[lo + k];
#pragma unroll
    for (int k = 0; k < HEAD; ++k) nn_consider_d2<PERM>(make_float4(v[k].x, v[k].y, v[k].z, 0.f), qx, qy, qz, best);
    for (int j = lo + HEAD; j < hi; j += 4) {
        const PwXyz3 a = p3[j], b = p3[j + 1], c = p3[j + 2], d = p3[j + 3];
        nn_consider_d2<PERM>(make_float4(a.x, a.y, a.z, 0.f), qx, qy, qz, best);
        nn_consider_d2<PERM>(make_float4(b.x, b.y, b.z, 0.f), qx, qy, qz, best);
        nn_consider_d2<PERM>(make_float4(c.x, c.y, c.z, 0.f), qx, qy, qz, best);
        nn_consider_d2<PERM>(make_float4(d.x, d.y, d.z, 0.f), qx, qy, qz, best);
    }
}

template <int PERM>
__device__ __forceinline__ void scan_ranges_flat(const GridLevel& g, const int2* __restrict__ list, int stride, int tid, int n, float ux, float uy,
                                                 float uz, float& best) {
    const PwXyz3* __restrict__ p3 = (const PwXyz3*)g.pts3;
    // (the NEXT range is read while the current one is walked: a range switch is two moves, not an LDS round trip in the lane's chain.
    // Measured and not kept, profiles/r06_dense_variants.txt: the next pass's points requested before the current four are evaluated,
    // 30.2 against 28.2 us replayed, 32.0 against 29.7 inside the loop.)
    int2 r0 = make_int2(0, 0), r1 = make_int2(0, 0);
    if (n > 0) r0 = list[tid];
    if (n > 1) r1 = list[stride + tid];
    int j = r0.x, e = r0.y, k = 2;
    while (j < e) {
        const PwXyz3 a = p3[j], b = p3[j + 1], c = p3[j + 2], d = p3[j + 3];
        nn_consider_d2<PERM>(make_float4(a.x, a.y, a.z, 0.f), ux, uy, uz, best);
        nn_consider_d2<PERM>(make_float4(b.x, b.y, b.z, 0.f), ux, uy, uz, best);
        nn_consider_d2<PERM>(make_float4(c.x, c.y, c.z, 0.f), ux, uy, uz, best);
        nn_consider_d2<PERM>(make_float4(d.x, d.y, d.z, 0.f), ux, uy, uz, best);
        j += 4;
        if (j >= e) {
            j = r1.x; e = r1.y;                          // (empty once the list is used up: the loop ends)
            r1 = make_int2(0, 0);
            if (k < n) { r1 = list[k * stride + tid]; ++k; }
        }
    }
}

// scan_disc_lean shared by a group of G lanes (one query): the rows of the ball are dealt to the lanes round-robin and every lane
// walks ITS rows as the per-lane search does (four rows' begin / end words in flight, then their points four at a time) - a wide
// ball (40 rows, 200 candidates) is 1/G of the rows and candidates per lane, i.e. a chain of ~10 round trips instead of ~90.
// Same pruning, same float expression per candidate: after the group minimum, the same exact d2.
// (measured and not kept, profiles/r05_dense_variants.txt (9): the lane's four rows side by side, two points of each per step)
template <int G>
__device__ __forceinline__ void scan_disc_group(const GridLevel& g, float qx, float qy, float qz, float rho, int sub, float& best) {
    const bool gy = g.inv_hy != 0.0f, gz = g.inv_hz != 0.0f;
    const int y0 = gy ? max(icell(qy - rho, g.oy, g.inv_hy), 0) : 0, y1 = gy ? min(icell(qy + rho, g.oy, g.inv_hy), g.ny - 1) : 0;
    const int z0 = gz ? max(icell(qz - rho, g.oz, g.inv_hz), 0) : 0, z1 = gz ? min(icell(qz + rho, g.oz, g.inv_hz), g.nz - 1) : 0;
    const float rho2 = rho * rho, slack2 = 2.0f * g.slack;
    const int wy = y1 - y0 + 1;
    const int nrows = (y1 >= y0 && z1 >= z0) ? wy * (z1 - z0 + 1) : 0;
    for (int t0 = sub; t0 < nrows; t0 += 4 * G) {
        int lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = hi[k] = 0;
            const int t = t0 + k * G;
            if (t >= nrows) continue;
            const int y = y0 + t % wy, z = z0 + t / wy;
            float rem = rho2;
            if (gz) {
                const float l = g.oz + (float)z * g.h;
                const float ez = fmaxf(fmaxf(l - qz, qz - (l + g.h)) - slack2, 0.0f);
                rem -= ez * ez;
            }
            if (gy) {
                const float l = g.oy + (float)y * g.h;
                const float ey = fmaxf(fmaxf(l - qy, qy - (l + g.h)) - slack2, 0.0f);
                rem -= ey * ey;
            }
            if (!(rem > 0.0f)) continue;
            const float rx = fast_sqrt_up(rem) + slack2;
            const int x0 = max(icell(qx - rx, g.ox, g.inv_h), 0), x1 = min(icell(qx + rx, g.ox, g.inv_h), g.nx - 1);
            if (x0 > x1) continue;
            const int row = (z * g.ny + y) * g.nx;
            lo[k] = g.cell_start[row + x0];
            hi[k] = g.cell_start[row + x1 + 1];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) scan_d2x4<0>(g.pts, lo[k], hi[k], qx, qy, qz, best);
    }
#pragma unroll
    for (int o = 1; o < G; o <<= 1) best = fminf(best, __shfl_xor(best, o));
}

// diagnostic: points examined.  `ctr` is an array of 256 counters, 128 bytes apart (one per cache line), indexed
// by block: same-line atomics from every wave would serialise (~5 ns each) and dominate a fast kernel.
__device__ __forceinline__ void add_examined(unsigned long long* ctr, unsigned cnt) {
    if (!ctr) return;
    unsigned long long c = cnt;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&ctr[(blockIdx.x & 255) * 16], c);
}
