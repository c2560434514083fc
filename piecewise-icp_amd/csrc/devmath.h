// Small dense math used by the patch-statistics and ICP kernels (device side).
// Formulas follow PCL 1.8.1 (common/impl/eigen.hpp eigen33/computeRoots, common/impl/centroid.hpp,
// registration/impl/transformation_estimation_point_to_plane_lls.hpp) as called by the reference at
// src/CommonFunc.cpp:284-354 and src/Registration.cpp:1255-1343.  Compiled with -ffp-contract=off.
#pragma once

#include <hip/hip_runtime.h>

#include <cfloat>

namespace pwdev {

// float trig evaluated in double and rounded once (see DESIGN.md "numerics")
__device__ __forceinline__ float f_atan2(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ __forceinline__ float f_cos(float x) { return (float)cos((double)x); }
__device__ __forceinline__ float f_sin(float x) { return (float)sin((double)x); }

// ---- symmetric 3x3 eigen decomposition, cyclic Jacobi in double ---------------------------------
// A (row-major, symmetric) -> eigenvalues w ascending, eigenvectors as columns of V (row-major)
__device__ inline void jacobi3(const double* Ain, double* w, double* V) {
    double A[3][3], U[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[i][j] = Ain[3 * i + j];
            U[i][j] = (i == j) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-18 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double ukp = U[k][p], ukq = U[k][q];
                    U[k][p] = c * ukp - s * ukq;
                    U[k][q] = s * ukp + c * ukq;
                }
            }
    }
    int o0 = 0, o1 = 1, o2 = 2;
    double e0 = A[0][0], e1 = A[1][1], e2 = A[2][2];
    // bubble sort of three (stable, same comparisons as a 2-pass bubble sort)
    if (e0 > e1) { double t = e0; e0 = e1; e1 = t; int ti = o0; o0 = o1; o1 = ti; }
    if (e1 > e2) { double t = e1; e1 = e2; e2 = t; int ti = o1; o1 = o2; o2 = ti; }
    if (e0 > e1) { double t = e0; e0 = e1; e1 = t; int ti = o0; o0 = o1; o1 = ti; }
    w[0] = e0; w[1] = e1; w[2] = e2;
    for (int r = 0; r < 3; ++r) {
        V[3 * r + 0] = U[r][o0];
        V[3 * r + 1] = U[r][o1];
        V[3 * r + 2] = U[r][o2];
    }
}

// ---- PCL eigen33: smallest eigenpair of a symmetric PSD float 3x3 -----------------------------------
__device__ inline void compute_roots2(float b, float c, float* roots) {
    roots[0] = 0.0f;
    float d = (float)((double)(b * b) - 4.0 * (double)c);
    if (d < 0.0f) d = 0.0f;
    float sd = sqrtf(d);
    roots[2] = 0.5f * (b + sd);
    roots[1] = 0.5f * (b - sd);
}

__device__ inline void compute_roots(const float* m, float* roots) {
    float c0 = m[0] * m[4] * m[8] + 2.0f * m[1] * m[2] * m[5] - m[0] * m[5] * m[5] - m[4] * m[2] * m[2] -
               m[8] * m[1] * m[1];
    float c1 = m[0] * m[4] - m[1] * m[1] + m[0] * m[8] - m[2] * m[2] + m[4] * m[8] - m[5] * m[5];
    float c2 = m[0] + m[4] + m[8];
    if (fabsf(c0) < FLT_EPSILON) {
        compute_roots2(c2, c1, roots);
        return;
    }
    const float s_inv3 = (float)(1.0 / 3.0);
    const float s_sqrt3 = sqrtf(3.0f);
    float c2_over_3 = c2 * s_inv3;
    float a_over_3 = (c1 - c2 * c2_over_3) * s_inv3;
    if (a_over_3 > 0.0f) a_over_3 = 0.0f;
    float half_b = 0.5f * (c0 + c2_over_3 * (2.0f * c2_over_3 * c2_over_3 - c1));
    float q = half_b * half_b + a_over_3 * a_over_3 * a_over_3;
    if (q > 0.0f) q = 0.0f;
    float rho = sqrtf(-a_over_3);
    float theta = f_atan2(sqrtf(-q), half_b) * s_inv3;
    float cos_theta = f_cos(theta);
    float sin_theta = f_sin(theta);
    roots[0] = c2_over_3 + 2.0f * rho * cos_theta;
    roots[1] = c2_over_3 - rho * (cos_theta + s_sqrt3 * sin_theta);
    roots[2] = c2_over_3 - rho * (cos_theta - s_sqrt3 * sin_theta);
    float t;
    if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    if (roots[1] >= roots[2]) {
        t = roots[1]; roots[1] = roots[2]; roots[2] = t;
        if (roots[0] >= roots[1]) { t = roots[0]; roots[0] = roots[1]; roots[1] = t; }
    }
    if (roots[0] <= 0.0f) compute_roots2(c2, c1, roots);
}

__device__ __forceinline__ void cross3(const float* a, const float* b, float* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ inline void eigen33_smallest(const float* mat, float* vec) {
    float scale = 0.0f;
    for (int i = 0; i < 9; ++i) scale = fmaxf(scale, fabsf(mat[i]));
    if (scale <= FLT_MIN) scale = 1.0f;
    float s[9];
    for (int i = 0; i < 9; ++i) s[i] = mat[i] / scale;
    float ev[3];
    compute_roots(s, ev);
    s[0] -= ev[0]; s[4] -= ev[0]; s[8] -= ev[0];
    float v1[3], v2[3], v3[3];
    cross3(s + 0, s + 3, v1);
    cross3(s + 0, s + 6, v2);
    cross3(s + 3, s + 6, v3);
    float l1 = v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2];
    float l2 = v2[0] * v2[0] + v2[1] * v2[1] + v2[2] * v2[2];
    float l3 = v3[0] * v3[0] + v3[1] * v3[1] + v3[2] * v3[2];
    if (l1 >= l2 && l1 >= l3) {
        float sl = sqrtf(l1);
        vec[0] = v1[0] / sl; vec[1] = v1[1] / sl; vec[2] = v1[2] / sl;
    } else if (l2 >= l1 && l2 >= l3) {
        float sl = sqrtf(l2);
        vec[0] = v2[0] / sl; vec[1] = v2[1] / sl; vec[2] = v2[2] / sl;
    } else {
        float sl = sqrtf(l3);
        vec[0] = v3[0] / sl; vec[1] = v3[1] / sl; vec[2] = v3[2] / sl;
    }
}

// ---- 6x6 inverse, LU with partial pivoting (Eigen PartialPivLU::inverse semantics) -------------------
__device__ inline bool inv6(const double* Ain, double* inv) {
    double A[6][6];
    int piv[6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) A[i][j] = Ain[6 * i + j];
    for (int i = 0; i < 6; ++i) piv[i] = i;
    for (int k = 0; k < 6; ++k) {
        int p = k;
        double best = fabs(A[k][k]);
        for (int i = k + 1; i < 6; ++i)
            if (fabs(A[i][k]) > best) { best = fabs(A[i][k]); p = i; }
        if (best == 0.0) return false;
        if (p != k) {
            for (int j = 0; j < 6; ++j) { double t = A[k][j]; A[k][j] = A[p][j]; A[p][j] = t; }
            int tp = piv[k]; piv[k] = piv[p]; piv[p] = tp;
        }
        for (int i = k + 1; i < 6; ++i) {
            A[i][k] = A[i][k] / A[k][k];
            for (int j = k + 1; j < 6; ++j) A[i][j] = A[i][j] - A[i][k] * A[k][j];
        }
    }
    for (int c = 0; c < 6; ++c) {
        double y[6];
        for (int i = 0; i < 6; ++i) {
            double s = (piv[i] == c) ? 1.0 : 0.0;
            for (int j = 0; j < i; ++j) s = s - A[i][j] * y[j];
            y[i] = s;
        }
        for (int i = 5; i >= 0; --i) {
            double s = y[i];
            for (int j = i + 1; j < 6; ++j) s = s - A[i][j] * inv[6 * j + c];
            inv[6 * i + c] = s / A[i][i];
        }
    }
    return true;
}

// Eigen Matrix4f product, row-major storage, k = 0..3 in order
__device__ __host__ inline void mat4_mul(const float* A, const float* B, float* C) {
    float R[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = A[4 * i + 0] * B[0 + j];
            s = s + A[4 * i + 1] * B[4 + j];
            s = s + A[4 * i + 2] * B[8 + j];
            s = s + A[4 * i + 3] * B[12 + j];
            R[4 * i + j] = s;
        }
    for (int i = 0; i < 16; ++i) C[i] = R[i];
}

// pcl::transformPointCloud: x' = ((m00*x + m01*y) + m02*z) + m03
__device__ __forceinline__ float4 xform_point(const float* T, float4 p) {
    float4 r;
    r.x = T[0] * p.x + T[1] * p.y + T[2] * p.z + T[3];
    r.y = T[4] * p.x + T[5] * p.y + T[6] * p.z + T[7];
    r.z = T[8] * p.x + T[9] * p.y + T[10] * p.z + T[11];
    r.w = p.w;
    return r;
}
__device__ __forceinline__ float4 xform_normal(const float* T, float4 n) {
    float4 r;
    r.x = T[0] * n.x + T[1] * n.y + T[2] * n.z;
    r.y = T[4] * n.x + T[5] * n.y + T[6] * n.z;
    r.z = T[8] * n.x + T[9] * n.y + T[10] * n.z;
    r.w = n.w;
    return r;
}

}  // namespace pwdev
