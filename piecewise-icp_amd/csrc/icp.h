// Inner ICP / VCM device work buffers and launch wrappers (icp.hip).
#pragma once
#include "classify_dev.h"
#include "common.h"
#include "stage_dev.h"

struct IcpState {
    float T[16];        // last incremental transformation (row-major)
    float Tfinal[16];   // accumulated final transformation
    int iters;
    int done;
    int reason;         // 1 max iterations, 2 transform epsilon, 3 abs MSE, 4 rel MSE
    int pad;
    double prev_mse;
};

// optional mailbox message sent by the last launch of an ICP batch: a[0..na) | b[0..nb) | IcpState -> dst, then seq
struct IcpMail {
    const unsigned* a = nullptr;
    int na = 0;
    const unsigned* b = nullptr;
    int nb = 0;
    unsigned* dst = nullptr;        // nullptr: no message
    unsigned* seq_ptr = nullptr;
    unsigned seq = 0;
};

// optional final message of a run, sent by the VCM's last launch: VCM (72 words) | diagnostic counter (2 words)
struct VcmMail {
    const unsigned long long* examined = nullptr;    // 256 partial counters, 16 words apart
    unsigned* dst = nullptr;                          // nullptr: no message
    unsigned* seq_ptr = nullptr;
    unsigned seq = 0;
};

struct IcpWork {
    DevBuf<float4> src, srcn;      // working source centroids + normals (transformed in place)
    DevBuf<int> match;
    DevBuf<double> partials;
    DevBuf<IcpState> state;
    DevBuf<unsigned> counter;      // blocks finished in the current launch (last one solves)
    DevBuf<double> qx, vcm;
    DevBuf<unsigned long long> agg; // per-block aggregates of the fused classification launch (k_classify_icp0)
    unsigned epoch = 0;            // tag of the current fused launch's aggregates
    int reserve(pwicp_context* ctx, int ns_max);
};

// fs (optional): passes 1 and 2 of a percentile selection ride on the first two launches (n_iter >= 2)
struct FusedSelect;
// sg (optional, both launch wrappers): the stage guard the converging tail evaluates (stage_dev.h)
int pw_icp_enqueue(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
                   int ns_max, const unsigned* ns_dev, double euclid_eps, int n_iter, const IcpMail* mail = nullptr,
                   const FusedSelect* fs = nullptr, const StageGuard* sg = nullptr);
// classification + order-preserving compaction of the stable patches + inner-ICP iteration 0, one launch (icp.hip)
int pw_classify_icp0_launch(pwicp_context* ctx, const ClassifyArgs& a, int* d_stable, float4* d_stCT, float4* d_stN, IcpWork* w,
                            unsigned* d_slot, double euclid_eps, const IcpMail* mail = nullptr, const StageGuard* sg = nullptr);
int pw_icp_run(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w, int ns,
               double euclid_eps, float* T16, int* iters_out);
// have_match: d_src are the stable centroids the last pw_classify_icp0_launch on `w` compacted (their matches are in w->match)
int pw_vcm_enqueue(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
                   const float4* d_src, int ns, const VcmMail* mail = nullptr, bool have_match = false);
// the run's last transform update + the VCM in one launch (icp.hip: k_xf_vcm)
int pw_xf_vcm_launch(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
                     const float4* d_stct, int ns_max, const VcmMail* mail, unsigned stage3_bits, const float4* d_cloud_in,
                     const float4* d_ctbp_in, const float4* d_pat_in, float4* d_cloud, int n, float4* d_ctbp, int n_ctbp, float4* d_pat,
                     int n_pat, unsigned* d_bbox_part, unsigned* d_slot);
int pw_vcm_run(pwicp_context* ctx, const GridDesc& g, const float4* d_tgt, const float4* d_tgt_n, IcpWork* w,
               const float4* d_src, int ns, double* VCM36);
