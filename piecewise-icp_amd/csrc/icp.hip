// Inner point-to-plane ICP and the variance-covariance matrix on the device.
//
// Reference: P2PICPwithPatchNormal src/Registration.cpp:1255-1269 (= pcl::IterativeClosestPointWithNormals::
// align: correspondence search, TransformationEstimationPointToPlaneLLS, transformPointCloudWithNormals,
// DefaultConvergenceCriteria) and calTransParaVCM Registration.cpp:1273-1343.
//
// One inner iteration = ONE launch (k_icp_iter), no host round trip:
//   accumulate (grid over the S stable centroids, 8 lanes per centroid): applies the previous incremental transform to
//                the working source (points + normals), exact 1-NN in the target-centroid grid, forms the row
//                [a b c nx ny nz | d] in float exactly as PCL does, widens to double and reduces the 21+6
//                sums (+ sum of d2 for the MSE test) with wave shuffles -> LDS -> one partial per block;
//   solve      (first wave of the block that finishes last): fixed-order sum of the block partials, 6x6 LU inverse,
//                x = inv*ATb, Rz*Ry*Rx matrix in double -> float, final = T*final, convergence tests, done flag,
//                and — in the last launch of a batch — the iteration record to the host mailbox.
// Launches after convergence are no-ops (done flag), so iterations are enqueued in small batches.
// calTransParaVCM is two launches of the same shape (k_vcm_normal, k_vcm_finish).
#include "common.h"
#include "devmath.h"
#include "icp.h"
#include "nn_device.h"

using namespace pwdev;

namespace {

constexpr int kBlock = 256;
constexpr int kNSums = 28;    // 21 ATA (upper triangle) + 6 ATb + 1 sum(d2)

__device__ __forceinline__ void block_reduce_store(double* v, int nv, double* __restrict__ partial_out) {
    __shared__ double sh[kBlock / 64][kNSums];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < nv; ++k) {
        double x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if (lane == 0) sh[wave][k] = x;
    }
    __syncthreads();
    if (threadIdx.x < nv) {
        double s = sh[0][threadIdx.x];
        for (int w = 1; w < kBlock / 64; ++w) s += sh[w][threadIdx.x];
        partial_out[threadIdx.x] = s;
    }
}

// Synchronisation inside ONE wave that communicates through LDS: the lanes run in lockstep and the LDS serves a wave's
// requests in order, so a fence (waits + no compiler reordering) is enough; no s_barrier, hence usable by the first
// wave of a larger block while its other waves have already left.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// Device -> host mailbox message of an ICP batch (see k_mail in loop.hip): word ranges a | b | the ICP state, then the
// sequence number with a system-scope release.  Executed by the first wave of one block.
__device__ __forceinline__ void icp_send_mail(const IcpMail& m, const IcpState* st) {
    const int t = threadIdx.x;
    if (t >= 64) return;
    const unsigned* c = (const unsigned*)st;
    const int nc = (int)(sizeof(IcpState) / 4);
    for (int i = t; i < m.na; i += 64) m.dst[i] = __hip_atomic_load(&m.a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = t; i < m.nb; i += 64) m.dst[m.na + i] = __hip_atomic_load(&m.b[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = t; i < nc; i += 64) m.dst[m.na + m.nb + i] = __hip_atomic_load(&c[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence_system();
    wave_sync();
    if (t == 0) __hip_atomic_store(m.seq_ptr, m.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Group-cooperative accumulate: kGroup (8) consecutive lanes share one stable centroid.  They split the rows of its
// nearest-neighbour search (nn_query_group: the search is a chain of dependent memory round trips at this size),
// then each lane keeps 4 of the 28 sums of the point's LLS row, so the reduction over the wave's 8 points needs
// 3 shuffle steps on 4 values instead of 6 steps on 28.
constexpr int kAccBlock = 1024;                       // 128 points per block: few partials for the solve kernel
constexpr int kAccPts = kAccBlock / kGroup;

__device__ void icp_solve_tail(IcpState* st, const double* partials, int ns, double mse_rel);

// One inner iteration in ONE launch: every block accumulates its 128 points; the block that finishes last (device
// counter) reduces the partials in a fixed order and solves the 6x6 system — the former second kernel, whose launch
// and first dependent loads cost as much as its arithmetic.  `mail` (optional, last launch of a batch) publishes the
// iteration record to the host mailbox from the same launch.
__global__ void __launch_bounds__(kAccBlock) k_icp_iter(GridDesc g, const float4* __restrict__ tgt,
                                                        const float4* __restrict__ tgt_n, float4* __restrict__ src,
                                                        float4* __restrict__ srcn, int ns_host,
                                                        const unsigned* __restrict__ ns_dev, IcpState* st,
                                                        double* __restrict__ partials, unsigned* __restrict__ counter,
                                                        double mse_rel, IcpMail mail) {
    __shared__ double sh[kAccBlock / 64][32];
    if (st->done) {                 // converged in an earlier launch of the batch: only the message is left to do
        if (mail.dst && blockIdx.x == 0) icp_send_mail(mail, st);
        return;
    }
    const int ns = ns_dev ? (int)*ns_dev : ns_host;          // the count may live on the device (no host sync)
    if (ns <= 0) {
        if (mail.dst && blockIdx.x == 0) icp_send_mail(mail, st);
        return;
    }
    if ((int)(blockIdx.x * kAccPts) >= ns) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x % kGroup;
    const int i = blockIdx.x * kAccPts + threadIdx.x / kGroup;
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
    if (i < ns) {
        float4 p = src[i], nrm = srcn[i];
        if (st->iters > 0) {       // transformPointCloudWithNormals with the previous estimate
            p = xform_point(st->T, p);
            nrm = xform_normal(st->T, nrm);
            if (sub == 0) { src[i] = p; srcn[i] = nrm; }
        }
        const NNBest b = nn_query_group(g, p.x, p.y, p.z, sub);
        const int bi = b.idx();
        const float4 t = tgt[bi], n = tgt_n[bi];
        const float sx = p.x, sy = p.y, sz = p.z, dx = t.x, dy = t.y, dz = t.z, nx = n.x, ny = n.y, nz = n.z;
        const double a = (double)(nz * sy - ny * sz);
        const double bb = (double)(nx * sz - nz * sx);
        const double c = (double)(ny * sx - nx * sy);
        const double d = (double)(nx * dx + ny * dy + nz * dz - nx * sx - ny * sy - nz * sz);
        // sum k (0..27) lives on lane sub = k % 8 as its (k / 8)-th value
        switch (sub) {
            case 0: w0 = a * a;   w1 = bb * nx;            w2 = (double)(nx * ny); w3 = nx * d; break;   // 0, 8, 16, 24
            case 1: w0 = a * bb;  w1 = bb * ny;            w2 = (double)(nx * nz); w3 = ny * d; break;   // 1, 9, 17, 25
            case 2: w0 = a * c;   w1 = bb * nz;            w2 = (double)(ny * ny); w3 = nz * d; break;   // 2, 10, 18, 26
            case 3: w0 = a * nx;  w1 = c * c;              w2 = (double)(ny * nz); w3 = (double)b.d2(); break;   // 3, 11, 19, 27
            case 4: w0 = a * ny;  w1 = c * nx;             w2 = (double)(nz * nz); break;                // 4, 12, 20
            case 5: w0 = a * nz;  w1 = c * ny;             w2 = a * d; break;                            // 5, 13, 21
            case 6: w0 = bb * bb; w1 = c * nz;             w2 = bb * d; break;                           // 6, 14, 22
            default: w0 = bb * c; w1 = (double)(nx * nx);  w2 = c * d; break;                            // 7, 15, 23
        }
    }
    // over the 8 points of the wave (lanes with equal `sub`), fixed order
#pragma unroll
    for (int o = kGroup; o < 64; o <<= 1) {
        w0 += __shfl_xor(w0, o); w1 += __shfl_xor(w1, o); w2 += __shfl_xor(w2, o); w3 += __shfl_xor(w3, o);
    }
    if (lane < kGroup) { sh[wave][lane] = w0; sh[wave][8 + lane] = w1; sh[wave][16 + lane] = w2; sh[wave][24 + lane] = w3; }
    __syncthreads();
    if (threadIdx.x < kNSums) {
        double acc = sh[0][threadIdx.x];
        for (int w = 1; w < kAccBlock / 64; ++w) acc += sh[w][threadIdx.x];
        // write-through (device-coherent) store: the partial sums reach the coherence point without a cache write-back
        __hip_atomic_store(&partials[(size_t)blockIdx.x * kNSums + threadIdx.x], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // last active block -> solve.  Only wave 0 (which stored the partials) goes on: drain its stores, count, and if it
    // is the last one the solve on that single wave (wave-level synchronisation only, see icp_solve_tail).  The partials
    // are write-through stores read back with device-coherent loads, so no fence (= L2 write-back + L1 invalidate, ~3.5 us
    // a pair) is needed on either side: draining the store queue before the count is the release.
    if (threadIdx.x >= 64) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned last = 0;
    if (threadIdx.x == 0) {
        const unsigned nact = (unsigned)((ns + kAccPts - 1) / kAccPts);
        const unsigned prev = atomicAdd(counter, 1u);
        last = (prev == nact - 1u) ? 1u : 0u;
        if (last) *counter = 0u;                             // re-armed for the next launch
    }
    last = (unsigned)__shfl((int)last, 0);
    if (!last) return;
    icp_solve_tail(st, partials, ns, mse_rel);
    if (mail.dst) {
        __threadfence();
        icp_send_mail(mail, st);
    }
}

// 6x6 inverse by LU with partial pivoting on ONE wave, operands in LDS.  Element (i,j) is owned by lane 6*i+j; every
// element goes through exactly the operations of the serial algorithm (devmath.h inv6) in the same order, so the
// result is bit-identical — only independent elements are updated side by side.  A is destroyed; inv receives A^-1.
__device__ __forceinline__ void inv6_wave(double (*A)[6], double (*inv)[6], int* piv, bool* singular) {
    const int t = threadIdx.x;
    if (t < 6) piv[t] = t;
    if (t == 0) *singular = false;
    wave_sync();
    for (int k = 0; k < 6; ++k) {
        if (t == 0) {
            int p = k;
            double best = fabs(A[k][k]);
            for (int i = k + 1; i < 6; ++i)
                if (fabs(A[i][k]) > best) { best = fabs(A[i][k]); p = i; }
            if (best == 0.0) *singular = true;
            piv[6] = p;                                   // scratch: pivot row of this step
        }
        wave_sync();
        const int p = piv[6];
        if (p != k && t < 6) { const double tmp = A[k][t]; A[k][t] = A[p][t]; A[p][t] = tmp; }
        if (p != k && t == 6) { const int tp = piv[k]; piv[k] = piv[p]; piv[p] = tp; }
        wave_sync();
        if (t > k && t < 6) A[t][k] = A[t][k] / A[k][k];
        wave_sync();
        if (t < 36) {
            const int i = t / 6, j = t % 6;
            if (i > k && j > k) A[i][j] = A[i][j] - A[i][k] * A[k][j];
        }
        wave_sync();
    }
    if (t < 6) {          // column t of the inverse: forward then back substitution (independent columns)
        double y[6];
        for (int i = 0; i < 6; ++i) {
            double s = (piv[i] == t) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) s = s - A[i][j] * y[j];
            y[i] = s;
        }
        for (int i = 5; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < 6; ++j) s = s - A[i][j] * inv[j][t];
            inv[i][t] = s / A[i][i];
        }
    }
    wave_sync();
    if (*singular && t < 36) inv[t / 6][t % 6] = NAN;
    wave_sync();
}

// fixed-order sum of the block partials, 6x6 LU inverse, x = inv*ATb, T from (alpha,beta,gamma,t), convergence tests of
// pcl::registration::DefaultConvergenceCriteria.  Runs on ONE wave (threadIdx.x < 64).
__device__ void icp_solve_tail(IcpState* st, const double* partials, int ns, double mse_rel) {
    const int nblocks = (ns + kAccPts - 1) / kAccPts;
    __shared__ double sums[kNSums], half[2][kNSums];
    __shared__ double A[6][6], inv[6][6], x[6], sc[6];
    __shared__ int piv[8];
    __shared__ bool singular;
    __shared__ float T[16], F[16];
    const int t = threadIdx.x;
    if (t < 2 * kNSums) {      // two lanes per sum (even / odd blocks), then one add: a fixed summation order
        const int k = t % kNSums, h = t / kNSums;
        double s = 0.0;
        // (written by other blocks of this launch: visible after the acquire fence in the caller.)  Many loads in
        // flight, then the adds in block order: the summation order stays fixed, the latency is paid once per pass
        int b = h;
        for (; b < nblocks; b += 64) {            // 32 guarded loads in flight (typical launches: one pass)
            double v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u)
                v[u] = (b + 2 * u < nblocks) ? __hip_atomic_load(&partials[(size_t)(b + 2 * u) * kNSums + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
#pragma unroll
            for (int u = 0; u < 32; ++u)
                if (b + 2 * u < nblocks) s += v[u];
        }
        half[h][k] = s;
    }
    wave_sync();
    if (t < kNSums) sums[t] = half[0][t] + half[1][t];
    wave_sync();
    if (t < 36) {           // symmetric fill from the 21 upper-triangle sums
        const int i = t / 6, j = t % 6, r = min(i, j), c = max(i, j);
        A[i][j] = sums[r * 6 - r * (r - 1) / 2 + (c - r)];
    }
    wave_sync();
    inv6_wave(A, inv, piv, &singular);
    if (t < 6) {
        double s = 0.0;
        for (int c = 0; c < 6; ++c) s += inv[t][c] * sums[21 + c];
        x[t] = s;
    }
    wave_sync();
    if (t < 3) { sc[t] = cos(x[t]); sc[3 + t] = sin(x[t]); }     // alpha, beta, gamma
    wave_sync();
    if (t == 0) {
        const double ca = sc[0], cb = sc[1], cg = sc[2], sa = sc[3], sb = sc[4], sg = sc[5];
        T[0] = (float)(cg * cb);
        T[1] = (float)(-sg * ca + cg * sb * sa);
        T[2] = (float)(sg * sa + cg * sb * ca);
        T[4] = (float)(sg * cb);
        T[5] = (float)(cg * ca + sg * sb * sa);
        T[6] = (float)(-cg * sa + sg * sb * ca);
        T[8] = (float)(-sb);
        T[9] = (float)(cb * sa);
        T[10] = (float)(cb * ca);
        T[3] = (float)x[3]; T[7] = (float)x[4]; T[11] = (float)x[5];
        T[12] = 0.f; T[13] = 0.f; T[14] = 0.f; T[15] = 1.f;
    }
    if (t < 16) F[t] = st->Tfinal[t];
    wave_sync();
    if (t < 16) {           // final = T * final (Eigen order), one element per lane
        const int i = t / 4, j = t % 4;
        float s = T[4 * i + 0] * F[0 + j];
        s = s + T[4 * i + 1] * F[4 + j];
        s = s + T[4 * i + 2] * F[8 + j];
        s = s + T[4 * i + 3] * F[12 + j];
        st->Tfinal[t] = s;
        st->T[t] = T[t];
    }
    if (t != 0) return;
    const int iters = st->iters + 1;
    st->iters = iters;
    // pcl::registration::DefaultConvergenceCriteria<float>::hasConverged()
    if (iters >= 100) { st->done = 1; st->reason = 1; return; }
    const double cos_angle = 0.5 * (double)(T[0] + T[5] + T[10] - 1.0f);
    const double translation_sqr = (double)(T[3] * T[3] + T[7] * T[7] + T[11] * T[11]);
    if (cos_angle >= 1.0 - 1e-8 && translation_sqr <= 1e-8) { st->done = 1; st->reason = 2; return; }
    const double mse = sums[27] / (double)ns;
    if (fabs(mse - st->prev_mse) < 1e-12) { st->done = 1; st->reason = 3; return; }
    if (fabs(mse - st->prev_mse) / st->prev_mse < mse_rel) { st->done = 1; st->reason = 4; return; }
    st->prev_mse = mse;
}

__global__ void k_icp_init(IcpState* st) {
    if (threadIdx.x == 0) {
        for (int k = 0; k < 16; ++k) {
            st->T[k] = (k % 5 == 0) ? 1.f : 0.f;
            st->Tfinal[k] = (k % 5 == 0) ? 1.f : 0.f;
        }
        st->iters = 0; st->done = 0; st->reason = 0; st->pad = 0;
        st->prev_mse = 1.7976931348623157e308;   // std::numeric_limits<double>::max()
    }
}

// ---- VCM (R.cpp:1273-1343) ------------------------------------------------------------------------------
constexpr int kVSums = 27;    // 21 ATA + 6 ATL

__device__ __forceinline__ void vcm_row(float4 q, float4 p, float4 n, double* a, double* L) {
    const double Qx = q.x, Qy = q.y, Qz = q.z, Px = p.x, Py = p.y, Pz = p.z, Nx = n.x, Ny = n.y, Nz = n.z;
    a[0] = Nz * Qy - Ny * Qz;
    a[1] = Nx * Qz - Nz * Qx;
    a[2] = Ny * Qx - Nx * Qy;
    a[3] = Nx; a[4] = Ny; a[5] = Nz;
    *L = Nx * (Px - Qx) + Ny * (Py - Qy) + Nz * (Pz - Qz);
}

// VCM step 1 in one launch: normal equations (group-cooperative like k_icp_iter: 8 lanes share a point's NN search and
// each keeps 4 of the 27 sums) and, on the block that finishes last, Qxx = (A^T A)^-1 and x = Qxx A^T L.
__global__ void __launch_bounds__(kAccBlock) k_vcm_normal(GridDesc g, const float4* __restrict__ tgt,
                                                          const float4* __restrict__ tgt_n,
                                                          const float4* __restrict__ src, int ns, int* __restrict__ match,
                                                          double* __restrict__ partials, unsigned* __restrict__ counter,
                                                          double* __restrict__ QX) {
    __shared__ double sh[kAccBlock / 64][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = threadIdx.x % kGroup;
    const int i = blockIdx.x * kAccPts + threadIdx.x / kGroup;
    double w0 = 0.0, w1 = 0.0, w2 = 0.0, w3 = 0.0;
    if (i < ns) {
        const float4 q = src[i];
        const NNBest b = nn_query_group(g, q.x, q.y, q.z, sub);
        const int bi = b.idx();
        if (sub == 0) match[i] = bi;
        double a[6], L;
        vcm_row(q, tgt[bi], tgt_n[bi], a, &L);
        // sum k (0..26; 21 upper-triangle products row by row, then a[r]*L) lives on lane sub = k % 8 as its (k / 8)-th value
        switch (sub) {
            case 0: w0 = a[0] * a[0]; w1 = a[1] * a[3]; w2 = a[3] * a[4]; w3 = a[3] * L; break;    // 0, 8, 16, 24
            case 1: w0 = a[0] * a[1]; w1 = a[1] * a[4]; w2 = a[3] * a[5]; w3 = a[4] * L; break;    // 1, 9, 17, 25
            case 2: w0 = a[0] * a[2]; w1 = a[1] * a[5]; w2 = a[4] * a[4]; w3 = a[5] * L; break;    // 2, 10, 18, 26
            case 3: w0 = a[0] * a[3]; w1 = a[2] * a[2]; w2 = a[4] * a[5]; break;                   // 3, 11, 19
            case 4: w0 = a[0] * a[4]; w1 = a[2] * a[3]; w2 = a[5] * a[5]; break;                   // 4, 12, 20
            case 5: w0 = a[0] * a[5]; w1 = a[2] * a[4]; w2 = a[0] * L; break;                      // 5, 13, 21
            case 6: w0 = a[1] * a[1]; w1 = a[2] * a[5]; w2 = a[1] * L; break;                      // 6, 14, 22
            default: w0 = a[1] * a[2]; w1 = a[3] * a[3]; w2 = a[2] * L; break;                     // 7, 15, 23
        }
    }
#pragma unroll
    for (int o = kGroup; o < 64; o <<= 1) {
        w0 += __shfl_xor(w0, o); w1 += __shfl_xor(w1, o); w2 += __shfl_xor(w2, o); w3 += __shfl_xor(w3, o);
    }
    if (lane < kGroup) { sh[wave][lane] = w0; sh[wave][8 + lane] = w1; sh[wave][16 + lane] = w2; sh[wave][24 + lane] = w3; }
    __syncthreads();
    if (threadIdx.x < kVSums) {
        double acc = sh[0][threadIdx.x];
        for (int w = 1; w < kAccBlock / 64; ++w) acc += sh[w][threadIdx.x];
        partials[(size_t)blockIdx.x * kNSums + threadIdx.x] = acc;
    }
    if (threadIdx.x >= 64) return;
    __threadfence();
    unsigned last = 0;
    if (threadIdx.x == 0) {
        last = (atomicAdd(counter, 1u) == gridDim.x - 1u) ? 1u : 0u;
        if (last) *counter = 0u;
    }
    last = (unsigned)__shfl((int)last, 0);
    if (!last) return;
    __threadfence();
    // ---- solve (one wave) ----
    __shared__ double sums[kVSums];
    __shared__ double A[6][6], Q[6][6];
    __shared__ int piv[8];
    __shared__ bool singular;
    const int t = threadIdx.x, nblocks = gridDim.x;
    if (t < kVSums) {
        double s = 0.0;
        int bk = 0;
        for (; bk + 8 <= nblocks; bk += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(bk + u) * kNSums + t];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; bk < nblocks; ++bk) s += partials[(size_t)bk * kNSums + t];
        sums[t] = s;
    }
    wave_sync();
    if (t < 36) {
        const int r0 = t / 6, c0 = t % 6, r = min(r0, c0), c = max(r0, c0);
        A[r0][c0] = sums[r * 6 - r * (r - 1) / 2 + (c - r)];
    }
    wave_sync();
    inv6_wave(A, Q, piv, &singular);
    if (t < 36) QX[t] = Q[t / 6][t % 6];
    if (t < 6) {
        double s = 0;
        for (int c = 0; c < 6; ++c) s += Q[t][c] * sums[21 + c];
        QX[36 + t] = s;
    }
}

// VCM step 2 in one launch: residuals v = A x - L, v^T v; the block that finishes last forms sigma0^2 * Qxx
// (R.cpp:1330-1340) and, when asked, publishes it together with the run's diagnostic counter to the host mailbox.
__global__ void __launch_bounds__(kBlock) k_vcm_finish(const float4* __restrict__ tgt, const float4* __restrict__ tgt_n,
                                                       const float4* __restrict__ src, int ns,
                                                       const int* __restrict__ match, const double* __restrict__ QX,
                                                       double* __restrict__ partials, unsigned* __restrict__ counter,
                                                       double* __restrict__ vcm, VcmMail mail) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double v[kNSums];
    v[0] = 0.0;
    if (i < ns) {
        double a[6], L;
        vcm_row(src[i], tgt[match[i]], tgt_n[match[i]], a, &L);
        double r = 0;
        for (int c = 0; c < 6; ++c) r += a[c] * QX[36 + c];
        r -= L;
        v[0] = r * r;
    }
    block_reduce_store(v, 1, partials + (size_t)blockIdx.x * kNSums);
    if (threadIdx.x >= 64) return;
    __threadfence();
    unsigned last = 0;
    if (threadIdx.x == 0) {
        last = (atomicAdd(counter, 1u) == gridDim.x - 1u) ? 1u : 0u;
        if (last) *counter = 0u;
    }
    last = (unsigned)__shfl((int)last, 0);
    if (!last) return;
    __threadfence();
    const int t = threadIdx.x, nblocks = gridDim.x;
    double vtpv = 0.0;
    {   // block order, eight loads in flight; every lane computes the same sum
        int bk = 0;
        for (; bk + 8 <= nblocks; bk += 8) {
            double w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = partials[(size_t)(bk + u) * kNSums];
#pragma unroll
            for (int u = 0; u < 8; ++u) vtpv += w[u];
        }
        for (; bk < nblocks; ++bk) vtpv += partials[(size_t)bk * kNSums];
    }
    const double STD0 = sqrt(vtpv / (double)(ns - 6));
    double out = 0.0;
    if (t < 36) { out = STD0 * STD0 * QX[t]; vcm[t] = out; }
    if (!mail.dst) return;
    // message: 36 doubles | the folded diagnostic counter (256 partial counters, 128 bytes apart) | seq
    unsigned long long ex = 0;
    for (int k = t; k < 256; k += 64) ex += mail.examined[(size_t)k * 16];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ex += __shfl_xor(ex, o);
    if (t < 36) {
        unsigned lo, hi;
        memcpy(&lo, &out, 4);
        memcpy(&hi, (const char*)&out + 4, 4);
        mail.dst[2 * t] = lo;
        mail.dst[2 * t + 1] = hi;
    }
    if (t == 0) {
        mail.dst[72] = (unsigned)(ex & 0xffffffffull);
        mail.dst[73] = (unsigned)(ex >> 32);
    }
    __threadfence_system();
    wave_sync();
    if (t == 0) __hip_atomic_store(mail.seq_ptr, mail.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

}  // namespace

// ==========================================================================================================
int IcpWork::reserve(pwicp_context* ctx, int ns_max) {
    int nb = std::max(div_up(std::max(ns_max, 1), kBlock), div_up(std::max(ns_max, 1), kAccPts));
    HIPCHK(ctx, src.reserve((size_t)std::max(ns_max, 1)));
    HIPCHK(ctx, srcn.reserve((size_t)std::max(ns_max, 1)));
    HIPCHK(ctx, match.reserve((size_t)std::max(ns_max, 1)));
    HIPCHK(ctx, partials.reserve((size_t)nb * kNSums));
    HIPCHK(ctx, state.reserve(1));
    HIPCHK(ctx, counter.reserve(1));
    HIPCHK(ctx, hipMemsetAsync(counter.p, 0, sizeof(unsigned), ctx->stream));
    HIPCHK(ctx, qx.reserve(48));
    HIPCHK(ctx, vcm.reserve(36));
    return PWICP_OK;
}

// Enqueues n_iter inner iterations (accumulate + solve each) on the stream; no host synchronisation.  The number
// of source points is ns_host, or *ns_dev when ns_dev != nullptr (then ns_max bounds the launch grid).
int pw_icp_enqueue(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
                   int ns_max, const unsigned* ns_dev, double euclid_eps, int n_iter, const IcpMail* mail) {
    if (ns_max <= 0) return PWICP_OK;
    const int nb = div_up(ns_max, kAccPts);
    IcpMail none{};
    for (int k = 0; k < n_iter; ++k)
        hipLaunchKernelGGL(k_icp_iter, dim3(nb), dim3(kAccBlock), 0, ctx->stream, g, d_tgt, d_tgt_n, w->src.p, w->srcn.p,
                           ns_max, ns_dev, w->state.p, w->partials.p, w->counter.p, euclid_eps,
                           (mail && k == n_iter - 1) ? *mail : none);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

// d_src / d_srcn: working copies (modified in place). Returns final T and the iteration count.
int pw_icp_run(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w, int ns,
               double euclid_eps, float* T16, int* iters_out) {
    for (int k = 0; k < 16; ++k) T16[k] = (k % 5 == 0) ? 1.f : 0.f;
    if (iters_out) *iters_out = 0;
    if (ns < 3 || g.fine.n <= 0) return PWICP_OK;      // min_number_correspondences_ = 3: no update
    hipLaunchKernelGGL(k_icp_init, dim3(1), dim3(64), 0, ctx->stream, w->state.p);
    IcpState h;
    for (int done_iters = 0; done_iters < 100;) {
        PWCHK(pw_icp_enqueue(ctx, g, d_tgt, d_tgt_n, w, ns, nullptr, euclid_eps, 4, nullptr));
        HIPCHK(ctx, hipMemcpyAsync(&h, w->state.p, sizeof(IcpState), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        done_iters = h.iters;
        if (h.done) break;
    }
    HIPCHK(ctx, hipGetLastError());
    memcpy(T16, h.Tfinal, sizeof(h.Tfinal));
    if (iters_out) *iters_out = h.iters;
    return PWICP_OK;
}

// enqueue only: the 6x6 result is left in w->vcm (device)
int pw_vcm_enqueue(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
                   const float4* d_src, int ns, const VcmMail* mail) {
    if (ns <= 0) return PWICP_OK;
    VcmMail none{};
    hipLaunchKernelGGL(k_vcm_normal, dim3(div_up(ns, kAccPts)), dim3(kAccBlock), 0, ctx->stream, g, d_tgt, d_tgt_n, d_src, ns,
                       w->match.p, w->partials.p, w->counter.p, w->qx.p);
    hipLaunchKernelGGL(k_vcm_finish, dim3(div_up(ns, kBlock)), dim3(kBlock), 0, ctx->stream, d_tgt, d_tgt_n, d_src, ns,
                       w->match.p, w->qx.p, w->partials.p, w->counter.p, w->vcm.p, mail ? *mail : none);
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}

int pw_vcm_run(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
               const float4* d_src, int ns, double* VCM36) {
    if (ns <= 0) { for (int i = 0; i < 36; ++i) VCM36[i] = NAN; return PWICP_OK; }
    PWCHK(pw_vcm_enqueue(ctx, g, d_tgt, d_tgt_n, w, d_src, ns, nullptr));
    HIPCHK(ctx, hipMemcpyAsync(VCM36, w->vcm.p, 36 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipGetLastError());
    return PWICP_OK;
}
