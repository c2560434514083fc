// Device-resident set of selected patches of one cloud (CSR) + per-patch statistics.
#pragma once
#include "common.h"

struct PatchSet {
    int m = 0;      // number of patches
    int tot = 0;    // number of patch points
    DevBuf<float4> pat;   // [tot] refined patch points, concatenated
    DevBuf<int> off;      // [m+1]
    DevBuf<int> src;      // [tot] original point index
    DevBuf<float4> ct;    // [m]  centroids
    DevBuf<float4> bp;    // [6m] boundary points (Xmax,Xmin,Ymax,Ymin,Zmax,Zmin)
    DevBuf<float> bpstd;  // [m]
    DevBuf<float> ctstd;  // [m]
};

// armed by a front launch for the launches behind it (patch.hip: front_init)
struct FrontInit {
    unsigned* slot = nullptr;              // scalar slot of the iteration the front belongs to: [0] <- 0xffffffff, [1] <- 0
    unsigned long long* zero = nullptr;    // counters to clear (start of a run)
    int n_zero = 0;
};

int pw_patch_normals_launch(pwicp_context* ctx, const float4* d_pat, const int* d_off, int m, float4* d_nrm);
// source patch normals + 1-NN of (centroids | boundary points) among the target centroids, one launch
int pw_front_launch(pwicp_context* ctx, const float4* d_pat, const int* d_off, int m, float4* d_nrm, const GridDesc& g,
                    const float4* d_q, int nq, int* d_idx, float* d_d2, const struct FusedSelect* fs = nullptr,
                    const FrontInit* init = nullptr);
// transform update + the next iteration's front in one launch (k_xf_front); *_in: the arrays the moved values are read from
// (== the output arrays except in the first update of a run on a lazily reset pair)
struct IcpState;
int pw_xf_front_launch(pwicp_context* ctx, const float4* d_pat_in, float4* d_pat, const int* d_off, int m, float4* d_nrm,
                       const GridDesc& g, const float4* d_ctbp_in, float4* d_ctbp, int nq, int* d_idx, float* d_d2,
                       const float4* d_cloud_in, float4* d_cloud, int n, const IcpState* d_state, const unsigned* d_ns,
                       unsigned* d_bbox_part, unsigned* d_slot, const struct FusedSelect* fs = nullptr,
                       const FrontInit* init = nullptr, const unsigned* d_guard = nullptr, int npat = 0);
// d_nrm == nullptr (both front launches): no normals, the patch points (npat of them) are only moved
// d_guard (optional): the launch only runs if *d_guard == 1 (the stage guard's flag, stage_dev.h)
int pw_patch_stats_launch(pwicp_context* ctx, const float4* d_pat, const int* d_off, int m, float4* ct, float4* bp,
                          float* bpstd, float* ctstd);
int pw_select_patches_dev(pwicp_context* ctx, const float4* d_cloud, int n, const int* d_labels, int nsv,
                          PatchSet* out);
int pw_point_patch_ids_launch(pwicp_context* ctx, const int* d_off, int m, int* d_pid);
