// Steps (2)-(4) of PwICP_singleIteration for ONE source patch (R.cpp:750-862): level of detection, the seven point-to-plane
// distances against the matched target patches' normals, the reference's threshold comparisons verbatim.
#pragma once
#include "common.h"

struct ClassifyArgs {
    int m2;
    const int* mCT; const float* dCT;          // centroid matches / squared distances (front launch)
    const int* mBP; const float* dBP;          // boundary-point matches (6 per patch)
    const float* ctstd1; const float* bpstd2;  // sigma_CT of the target patches, sigma_BP of the source patches
    const float4* nrm1;                        // target patch normals, .w != 0: valid (calPatchNormal)
    const float4* ct1;                         // target centroids
    const float4* ct1n;                        // target centroid normals as the inner ICP sees them (C.cpp:357-382)
    const float4* ct2; const float4* bp2;      // source centroids / boundary points (current state)
    const float4* nrm2;                        // source patch normals (front launch)
    const int* off2;                           // source patch offsets
    float currDT, DTmin, DTctct;
};

namespace pwdev {

// what inner-ICP iteration 0 needs of a patch, gathered in the same two round trips as the classification's own operands
struct ClassifyRow { float4 q, t, tn; float dct; };

// returns the stable flag; *lod = the patch's level of detection; *row (optional): the operands of the patch's LLS row
__device__ __forceinline__ int classify_patch(const ClassifyArgs& a, int i, float* lod, ClassifyRow* row = nullptr) {
    // All loads that only need the patch index first (the seven matches, the patch's own points and sigma), then everything
    // that hangs on a match (the matched patches' sigma, normals, centroids): two memory round trips, whatever order the
    // arithmetic below consumes them in.
    const int j = max(a.mCT[i], 0);                  // (-1 = empty target: rejected on the host before the launch)
    int jb[6];
    float4 b[6], nn[6], tt[6];
    float db[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) { jb[k] = max(a.mBP[6 * i + k], 0); b[k] = a.bp2[6 * i + k]; db[k] = a.dBP[6 * i + k]; }
    const float s2 = a.bpstd2[i];
    const float4 q = a.ct2[i];
    const float dct = a.dCT[i];
    const float s1 = a.ctstd1[j];
    const float4 n = a.nrm1[j], t = a.ct1[j];
    if (row) { row->q = q; row->t = t; row->tn = a.ct1n[j]; row->dct = dct; }
#pragma unroll
    for (int k = 0; k < 6; ++k) { nn[k] = a.nrm1[jb[k]]; tt[k] = a.ct1[jb[k]]; }
    // (2) level of detection, R.cpp:756-766
    const float maxLoD = a.DTmin * 2.0f, minLoD = a.DTmin;
    float LoD = (float)(1.96 * (double)sqrtf(s1 * s1 + s2 * s2));
    if (LoD > maxLoD) LoD = maxLoD; else if (LoD < minLoD) LoD = minLoD;
    *lod = LoD;
    // (3) point-to-plane distances with the matched TARGET patch normal, R.cpp:781-812
    float resCT;
    if (n.w != 0.0f) {
        const float dx = t.x - q.x, dy = t.y - q.y, dz = t.z - q.z;
        resCT = fabsf(dx * n.x + dy * n.y + dz * n.z);
    } else resCT = sqrtf(dct);
    const float p2pt = sqrtf(dct);
    // (4) R.cpp:826-862; `thr < dist` fails, exactly the reference's comparisons
    const float thr = (a.currDT <= LoD) ? LoD : a.currDT;
    bool pass = !(thr < resCT);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        float res;
        if (nn[k].w != 0.0f) {
            const float dx = tt[k].x - b[k].x, dy = tt[k].y - b[k].y, dz = tt[k].z - b[k].z;
            res = fabsf(dx * nn[k].x + dy * nn[k].y + dz * nn[k].z);
        } else res = sqrtf(db[k]);
        if (thr < res) pass = false;
    }
    return (pass && (p2pt < a.DTctct)) ? 1 : 0;
}

}  // namespace pwdev
