// pwicp/Registration.h — C++ facade with the reference's own function names and argument meaning
// (yihui4d/Piecewise-ICP include/Registration.h, include/CommonFunc.h), forwarding to the C ABI of pwicp.h.
//
// Two layers:
//  * namespace pwicp: templates over "cloud-like" types (anything with a contiguous `.points` of 16-byte x,y,z,pad
//    structs — pcl::PointCloud<pcl::PointXYZ> qualifies — and matrix types indexable as m(r, c)).  They compile
//    without PCL (tests/facade_compile_check.cpp instantiates them with plain structs).
//  * when PCL and Eigen are available at the user's site (`__has_include`), global functions with EXACTLY the
//    reference's signatures (Registration.h:149-153, 213-214, 227-229, 116-129; CommonFunc.h) so that the reference's
//    src/Registration.cpp can be replaced by this header plus libpwicp.so.
// The context is per thread (the reference keeps module globals instead, Registration.cpp:11-14).
#pragma once

#include <cstdint>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <cstring>
#include <vector>

#include "../pwicp.h"

namespace pwicp {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// one context per thread, created on first use on device $PWICP_DEVICE / $LOCAL_RANK (default 0)
inline pwicp_context* thread_context() {
    struct Holder {
        pwicp_context* ctx = nullptr;
        ~Holder() { if (ctx) pwicp_destroy(ctx); }
    };
    static thread_local Holder h;
    if (!h.ctx) {
        const char* e = std::getenv("PWICP_DEVICE");
        if (!e) e = std::getenv("LOCAL_RANK");
        const int rc = pwicp_create(&h.ctx, e ? std::atoi(e) : 0);
        if (rc != PWICP_OK) throw Error(rc, "pwicp: no usable HIP device (there is no CPU fallback)");
    }
    return h.ctx;
}

inline void check(int rc) {
    if (rc != PWICP_OK) throw Error(rc, pwicp_last_error(thread_context()));
}

template <class Cloud>
inline const float* xyz4(const Cloud& c) {
    static_assert(sizeof(c.points[0]) == 16, "point type must be 16 bytes (x, y, z, pad) like pcl::PointXYZ");
    return reinterpret_cast<const float*>(c.points.data());
}

// Piecewise_ICP (Registration.h:149-153) given the supervoxel labels of both clouds.
// `labels*`: output of the segmentation front end (pwicp_frontend_segment_dev or the reference's own codelibrary).
template <class Cloud, class Mat4, class MatX>
void Piecewise_ICP_labelled(const Cloud& cloud1, Cloud& cloud2, const std::vector<int32_t>& labels1, int nsv1,
                            const std::vector<int32_t>& labels2, int nsv2, float Res1, float Res2, float SVRes1,
                            float SVRes2, bool isManualDTinit, float DTinit, float DTmin, std::vector<float>& DTseries,
                            Mat4& transMat, MatX& VCM) {
    pwicp_context* ctx = thread_context();
    pwicp_params prm{Res1, Res2, SVRes1, SVRes2, isManualDTinit ? 1 : 0, DTinit, DTmin};
    pwicp_pair* pair = nullptr;
    check(pwicp_pair_create(ctx, xyz4(cloud1), (int)cloud1.points.size(), labels1.data(), nsv1, xyz4(cloud2),
                            (int)cloud2.points.size(), labels2.data(), nsv2, &prm, &pair));
    pwicp_result res;
    const int rc = pwicp_pair_run(pair, &res);
    if (rc == PWICP_OK)      // the reference transforms cloud2 in place (Registration.cpp:943-945)
        pwicp_pair_download_source(pair, reinterpret_cast<float*>(cloud2.points.data()));
    pwicp_pair_destroy(pair);
    check(rc);
    DTseries.assign(res.DTseries, res.DTseries + res.n_outer + 1);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) transMat(r, c) = res.T16[4 * r + c];
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) VCM(r, c) = res.VCM[6 * r + c];
}

// Piecewise_ICP with the library's own front end (k-NN graph on the GPU + host fusion)
template <class Cloud, class Mat4, class MatX>
void Piecewise_ICP(const Cloud& cloud1, Cloud& cloud2, bool isSetResSVsize, float Res1, float Res2, float SVsize1,
                   float SVsize2, bool isManualDTinit, float DTinit, float DTmin, std::vector<float>& DTseries,
                   Mat4& transMat, MatX& VCM) {
    const float SVRes1 = isSetResSVsize ? SVsize1 : Res1 * 10, SVRes2 = isSetResSVsize ? SVsize2 : Res2 * 10;   // R.cpp:635-640
    pwicp_context* ctx = thread_context();
    std::vector<int32_t> l1(cloud1.points.size()), l2(cloud2.points.size());
    int n1 = 0, n2 = 0;
    check(pwicp_frontend_segment_dev(ctx, xyz4(cloud1), (int)l1.size(), SVRes1, 45, Res1, l1.data(), &n1));
    check(pwicp_frontend_segment_dev(ctx, xyz4(cloud2), (int)l2.size(), SVRes2, 45, Res2, l2.data(), &n2));
    Piecewise_ICP_labelled(cloud1, cloud2, l1, n1, l2, n2, Res1, Res2, SVRes1, SVRes2, isManualDTinit, DTinit, DTmin,
                           DTseries, transMat, VCM);
}

// calPercentileDistBetween2PC (CommonFunc.h; CommonFunc.cpp:266-281)
template <class Cloud>
double calPercentileDistBetween2PC(const Cloud& cloud1, const Cloud& cloud2, float percentile) {
    double d = 0;
    check(pwicp_percentile_dist(thread_context(), xyz4(cloud1), (int)cloud1.points.size(), xyz4(cloud2),
                                (int)cloud2.points.size(), percentile, &d));
    return d;
}

// calOverlapRatioByC2Cdist (Registration.h:116-129)
template <class Cloud>
float calOverlapRatioByC2Cdist(const Cloud& cloud1, const Cloud& cloud2, float DTinit) {
    float r = 0;
    check(pwicp_overlap_ratio(thread_context(), xyz4(cloud1), (int)cloud1.points.size(), xyz4(cloud2),
                              (int)cloud2.points.size(), DTinit, &r));
    return r;
}

// calPatchNormal (CommonFunc.h; CommonFunc.cpp:284-333) for ONE patch
template <class Cloud>
bool calPatchNormal(const Cloud& patch, float& nx, float& ny, float& nz) {
    const int32_t off[2] = {0, (int32_t)patch.points.size()};
    float n4[4] = {0, 0, 1, 0};
    uint8_t ok = 0;
    check(pwicp_patch_normals(thread_context(), xyz4(patch), off, 1, n4, &ok));
    nx = n4[0]; ny = n4[1]; nz = n4[2];
    return ok != 0;
}

// P2PICPwithPatchNormal / calTransParaVCM on PointNormal-like clouds (48-byte points: xyz pad | normal pad | curvature pad)
template <class NCloud>
inline void split_point_normal(const NCloud& c, std::vector<float>* xyz, std::vector<float>* nrm) {
    static_assert(sizeof(c.points[0]) == 48, "point type must be 48 bytes like pcl::PointNormal");
    const size_t n = c.points.size();
    xyz->resize(4 * n); nrm->resize(4 * n);
    const float* p = reinterpret_cast<const float*>(c.points.data());
    for (size_t i = 0; i < n; ++i)
        for (int k = 0; k < 4; ++k) { (*xyz)[4 * i + k] = p[12 * i + k]; (*nrm)[4 * i + k] = p[12 * i + 4 + k]; }
}

template <class NCloud, class Mat4>
Mat4 P2PICPwithPatchNormal(const NCloud& cloudTarget, const NCloud& cloudSource, double EucldEpsilon) {
    std::vector<float> t, tn, s, sn;
    split_point_normal(cloudTarget, &t, &tn);
    split_point_normal(cloudSource, &s, &sn);
    float T[16];
    check(pwicp_p2p_icp(thread_context(), t.data(), tn.data(), (int)(t.size() / 4), s.data(), sn.data(), (int)(s.size() / 4),
                        EucldEpsilon, T, nullptr));
    Mat4 M;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) M(r, c) = T[4 * r + c];
    return M;
}

template <class Cloud, class NCloud, class MatX>
void calTransParaVCM(const Cloud& cloudTarget, const NCloud& cloudTargetwithNormals, const Cloud& cloudSourceStable, MatX& VCM) {
    std::vector<float> t, tn;
    split_point_normal(cloudTargetwithNormals, &t, &tn);
    double V[36];
    check(pwicp_trans_para_vcm(thread_context(), xyz4(cloudTarget), tn.data(), (int)cloudTarget.points.size(),
                               xyz4(cloudSourceStable), (int)cloudSourceStable.points.size(), V));
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) VCM(r, c) = V[6 * r + c];
}

// ---- the stages before the loop -------------------------------------------------------------------------------------

// PCpreprocessing (CommonFunc.h:196-198; CommonFunc.cpp:423-452): VoxelGrid + SOR, on the GPU
template <class Cloud>
void PCpreprocessing(const Cloud& cloud_in, Cloud& cloud_out, bool isDownSamp, float voxelSize, int SOR_NeighborNum,
                     double SOR_StdMult) {
    const int n = (int)cloud_in.points.size();
    std::vector<float> out((size_t)(n > 0 ? n : 1) * 4);
    int m = 0;
    if (isDownSamp)
        check(pwicp_preprocess_dev(thread_context(), xyz4(cloud_in), n, voxelSize, SOR_NeighborNum, SOR_StdMult, out.data(), &m));
    else      // copyPointCloud + SORfilter (C.cpp:436-439)
        check(pwicp_sor_filter_dev(thread_context(), xyz4(cloud_in), n, SOR_NeighborNum, SOR_StdMult, 0.f, out.data(), &m));
    cloud_out.points.resize((size_t)m);
    if (m) std::memcpy(static_cast<void*>(cloud_out.points.data()), out.data(), (size_t)m * 16);
}

// calPCresolution (CommonFunc.h:116; CommonFunc.cpp:239-263)
template <class Cloud>
float calPCresolution(const Cloud& cloud) {
    float r = 0.f;
    check(pwicp_pc_resolution_dev(thread_context(), xyz4(cloud), (int)cloud.points.size(), &r));
    return r;
}

// PatchGenerationAndRefinement (Segmentation.h:399-403; Segmentation.cpp:11-192) + calBPandCTSTD (Segmentation.h:451-452):
// supervoxels (GPU k-NN graph + host passes), then patch extraction / refinement / selection and the per-patch
// centroids, boundary points and sigmas on the GPU.  `pointSpacing` replaces the reference's global resolution; the
// patches come back as a vector instead of a new[]-ed array (S.cpp:84), the sigmas of calBPandCTSTD with them.
template <class Cloud>
int PatchGenerationAndRefinement(const Cloud& cloud, float svResolution, float pointSpacing, Cloud& cloudCentroid,
                                 Cloud& cloudBoundary, std::vector<Cloud>& cloudPatches, std::vector<float>& stdBP,
                                 std::vector<float>& stdCT) {
    pwicp_context* ctx = thread_context();
    const int n = (int)cloud.points.size();
    std::vector<int32_t> lab((size_t)(n > 0 ? n : 1));
    int nsv = 0;
    check(pwicp_frontend_segment_dev(ctx, xyz4(cloud), n, svResolution, 45, pointSpacing, lab.data(), &nsv));
    int m = 0, tot = 0;
    check(pwicp_select_patches(ctx, xyz4(cloud), n, lab.data(), nsv, &m, &tot, nullptr, nullptr, nullptr, nullptr, nullptr,
                               nullptr, nullptr));
    std::vector<float> pat((size_t)(tot > 0 ? tot : 1) * 4);
    std::vector<int32_t> off((size_t)m + 1);
    cloudCentroid.points.resize((size_t)m);
    cloudBoundary.points.resize((size_t)m * 6);
    stdBP.resize((size_t)(m > 0 ? m : 1));
    stdCT.resize((size_t)(m > 0 ? m : 1));
    std::vector<float> ct((size_t)(m > 0 ? m : 1) * 4), bp((size_t)(m > 0 ? m : 1) * 24);
    check(pwicp_select_patches(ctx, xyz4(cloud), n, lab.data(), nsv, &m, &tot, pat.data(), off.data(), nullptr, ct.data(),
                               bp.data(), stdBP.data(), stdCT.data()));
    stdBP.resize((size_t)m);
    stdCT.resize((size_t)m);
    if (m) std::memcpy(static_cast<void*>(cloudCentroid.points.data()), ct.data(), (size_t)m * 16);
    if (m) std::memcpy(static_cast<void*>(cloudBoundary.points.data()), bp.data(), (size_t)m * 96);
    cloudPatches.assign((size_t)m, Cloud());
    for (int k = 0; k < m; ++k) {
        const int cnt = off[(size_t)k + 1] - off[(size_t)k];
        cloudPatches[(size_t)k].points.resize((size_t)cnt);
        if (cnt) std::memcpy(static_cast<void*>(cloudPatches[(size_t)k].points.data()), pat.data() + 4 * (size_t)off[(size_t)k], (size_t)cnt * 16);
    }
    return m;
}

// the reference's module globals g_toStage2 / g_toStage3 (R.cpp:11-14), per thread here; Piecewise_ICP resets them (R.cpp:623-624)
inline int& toStage2() { static thread_local int v = 0; return v; }
inline int& toStage3() { static thread_local int v = 0; return v; }

}  // namespace pwicp

// ---- the reference's exact signatures, when PCL + Eigen are present ----------------------------------------------------
#if defined(__has_include)
#if __has_include(<pcl/point_types.h>) && __has_include(<pcl/point_cloud.h>) && __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#define PWICP_HAVE_PCL 1

// Registration.h:149-153
inline void Piecewise_ICP(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2,
                          bool isSetResSVsize, float Res1, float Res2, float SVsize1, float SVsize2, bool isManualDTinit,
                          float DTinit, float DTmin, std::vector<float>& DTseries, Eigen::Matrix4f& transMat,
                          Eigen::MatrixXd& VCM) {
    pwicp::toStage2() = 0; pwicp::toStage3() = 0;                  // R.cpp:623-624
    VCM.resize(6, 6);
    pwicp::Piecewise_ICP(*cloud1, *cloud2, isSetResSVsize, Res1, Res2, SVsize1, SVsize2, isManualDTinit, DTinit, DTmin,
                         DTseries, transMat, VCM);
}
// Registration.h:213-214
inline Eigen::Matrix4f P2PICPwithPatchNormal(pcl::PointCloud<pcl::PointNormal>::Ptr cloudTarget,
                                             pcl::PointCloud<pcl::PointNormal>::Ptr cloudSource, double EucldEpsilon) {
    return pwicp::P2PICPwithPatchNormal<pcl::PointCloud<pcl::PointNormal>, Eigen::Matrix4f>(*cloudTarget, *cloudSource, EucldEpsilon);
}
// Registration.h:227-229
inline Eigen::MatrixXd calTransParaVCM(pcl::PointCloud<pcl::PointXYZ>::Ptr cloudTarget,
                                       pcl::PointCloud<pcl::PointNormal>::Ptr cloudTargetwithNormals,
                                       pcl::PointCloud<pcl::PointXYZ>::Ptr cloudSourceStable) {
    Eigen::MatrixXd V(6, 6);
    pwicp::calTransParaVCM(*cloudTarget, *cloudTargetwithNormals, *cloudSourceStable, V);
    return V;
}
// Registration.h:116-129
inline float calOverlapRatioByC2Cdist(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2,
                                      float DTinit) {
    return pwicp::calOverlapRatioByC2Cdist(*cloud1, *cloud2, DTinit);
}
// CommonFunc.h
inline double calPercentileDistBetween2PC(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2,
                                          float percentile) {
    return pwicp::calPercentileDistBetween2PC(*cloud1, *cloud2, percentile);
}
inline bool calPatchNormal(pcl::PointCloud<pcl::PointXYZ> cloud, float& nx, float& ny, float& nz) {
    return pwicp::calPatchNormal(cloud, nx, ny, nz);
}
// CommonFunc.h:196-198
inline void PCpreprocessing(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud_in, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud_out,
                            bool isDownSamp, float voxelSize, int SOR_NeighborNum, double SOR_StdMult) {
    pwicp::PCpreprocessing(*cloud_in, *cloud_out, isDownSamp, voxelSize, SOR_NeighborNum, SOR_StdMult);
    cloud_out->width = (uint32_t)cloud_out->points.size(); cloud_out->height = 1; cloud_out->is_dense = true;
}
// CommonFunc.h:116
inline float calPCresolution(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud) { return pwicp::calPCresolution(*cloud); }
// CommonFunc.h:209 (C.cpp:441-452)
inline void SORfilter(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud_in, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud_out, int SOR_NeighborNum,
                      double SOR_StdMult) {
    pwicp::PCpreprocessing(*cloud_in, *cloud_out, false, 0.f, SOR_NeighborNum, SOR_StdMult);
    cloud_out->width = (uint32_t)cloud_out->points.size(); cloud_out->height = 1; cloud_out->is_dense = true;
}
// CommonFunc.h:150 (C.cpp:336-354)
inline float calPatchSTD(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud) {
    const int32_t off[2] = {0, (int32_t)cloud->points.size()};
    float sd = 0.f;
    pwicp::check(pwicp_patch_stats(pwicp::thread_context(), pwicp::xyz4(*cloud), off, 1, nullptr, nullptr, &sd, nullptr));
    return sd;
}
// CommonFunc.h:161 (C.cpp:357-382): normals of all patches in one launch; > 6 points and a valid normal, else (0,0,1)
inline void generateCentroidCloudWithPatchNormals(pcl::PointCloud<pcl::PointXYZ>::Ptr cloudCentroids,
                                                  pcl::PointCloud<pcl::PointXYZ>* cloudPatch,
                                                  pcl::PointCloud<pcl::PointNormal>::Ptr cloudCentroids_normals) {
    const int m = (int)cloudCentroids->points.size();
    std::vector<int32_t> off((size_t)m + 1, 0);
    for (int i = 0; i < m; ++i) off[(size_t)i + 1] = off[(size_t)i] + (int32_t)cloudPatch[i].points.size();
    std::vector<float> pat((size_t)(off[(size_t)m] > 0 ? off[(size_t)m] : 1) * 4), nrm((size_t)(m > 0 ? m : 1) * 4);
    std::vector<uint8_t> ok((size_t)(m > 0 ? m : 1));
    for (int i = 0; i < m; ++i)
        if (!cloudPatch[i].points.empty())
            std::memcpy(pat.data() + 4 * (size_t)off[(size_t)i], cloudPatch[i].points.data(), cloudPatch[i].points.size() * 16);
    pwicp::check(pwicp_patch_normals(pwicp::thread_context(), pat.data(), off.data(), m, nrm.data(), ok.data()));
    cloudCentroids_normals->points.assign((size_t)m, pcl::PointNormal());
    for (int i = 0; i < m; ++i) {
        pcl::PointNormal& q = cloudCentroids_normals->points[(size_t)i];
        const pcl::PointXYZ& c = cloudCentroids->points[(size_t)i];
        q.x = c.x; q.y = c.y; q.z = c.z;
        const bool good = cloudPatch[i].points.size() > 6 && ok[(size_t)i];
        q.normal_x = good ? nrm[4 * (size_t)i] : 0.f; q.normal_y = good ? nrm[4 * (size_t)i + 1] : 0.f; q.normal_z = good ? nrm[4 * (size_t)i + 2] : 1.f;
    }
    cloudCentroids_normals->width = (uint32_t)m; cloudCentroids_normals->height = 1;
}
// CommonFunc.h:172 (C.cpp:385-407)
inline void matrix2angle(Eigen::Matrix4f transMat, Eigen::Vector3f& rotAngle) {
    float T[16], a[3];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T[4 * r + c] = transMat(r, c);
    pwicp_matrix2angle(T, a);
    rotAngle[0] = a[0]; rotAngle[1] = a[1]; rotAngle[2] = a[2];
}
// CommonFunc.h:183 (C.cpp:410-419)
inline float calBoundingBoxCornerChange(const double* boundingBox, const Eigen::Matrix4f transMat) {
    float T[16];
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T[4 * r + c] = transMat(r, c);
    return pwicp_bbox_corner_change(boundingBox, T);
}

// Segmentation.h:399-403 (S.cpp:11-192): returns the number of selected patches; cloudPatches = new[] array the CALLER
// delete[]s (R.cpp:696-697), sized like the reference's (number of supervoxels), the first `return value` entries filled.
inline int PatchGenerationAndRefinement(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud, float svResolution,
                                        pcl::PointCloud<pcl::PointXYZ>::Ptr cloudCentroid,
                                        pcl::PointCloud<pcl::PointXYZ>::Ptr cloudBoundary,
                                        pcl::PointCloud<pcl::PointXYZ>*& cloudPatches, bool isVis) {
    (void)isVis;                                   // visualisation is out of scope
    std::vector<pcl::PointCloud<pcl::PointXYZ>> patches;
    std::vector<float> sbp, sct;
    const int m = pwicp::PatchGenerationAndRefinement(*cloud, svResolution, 0.f, *cloudCentroid, *cloudBoundary, patches, sbp, sct);
    cloudPatches = new pcl::PointCloud<pcl::PointXYZ>[(size_t)(m > 0 ? m : 1)];
    for (int i = 0; i < m; ++i) { cloudPatches[i].points.swap(patches[(size_t)i].points); cloudPatches[i].width = (uint32_t)cloudPatches[i].points.size(); cloudPatches[i].height = 1; }
    cloudCentroid->width = (uint32_t)m; cloudCentroid->height = 1; cloudBoundary->width = (uint32_t)(6 * m); cloudBoundary->height = 1;
    return m;
}
// Segmentation.h:451-452 (S.cpp:306-321)
inline void calBPandCTSTD(pcl::PointCloud<pcl::PointXYZ>* cloudPatches, int patchNum, std::vector<float>& stdBP, std::vector<float>& stdCT) {
    std::vector<int32_t> off((size_t)patchNum + 1, 0);
    for (int i = 0; i < patchNum; ++i) off[(size_t)i + 1] = off[(size_t)i] + (int32_t)cloudPatches[i].points.size();
    std::vector<float> pat((size_t)(off[(size_t)patchNum] > 0 ? off[(size_t)patchNum] : 1) * 4);
    for (int i = 0; i < patchNum; ++i)
        if (!cloudPatches[i].points.empty())
            std::memcpy(pat.data() + 4 * (size_t)off[(size_t)i], cloudPatches[i].points.data(), cloudPatches[i].points.size() * 16);
    stdBP.assign((size_t)patchNum, 0.f); stdCT.assign((size_t)patchNum, 0.f);
    pwicp::check(pwicp_patch_stats(pwicp::thread_context(), pat.data(), off.data(), patchNum, nullptr, nullptr, stdBP.data(), stdCT.data()));
}

// Registration.h:181-188 (R.cpp:704-972).  The reference hands every array over on each call; so does this wrapper: a pair
// is built from the given clouds and patch arrays, ONE iteration runs on the GPU (pwicp_pair_step), and everything the
// reference transforms in place comes back (cloud2, CTcloud2, BPcloud2, SVcloud2[i]).  The centroids, boundary points and
// sigmas are the caller's arrays (pwicp_pair_create_from_arrays), as in the reference.  A resident pair stepped with
// pwicp_pair_step is the fast path; this signature costs an upload per call.
inline Eigen::Matrix4f PwICP_singleIteration(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2, float Res1,
                                             float Res2, float SVRes1, float SVRes2, pcl::PointCloud<pcl::PointXYZ>*& SVcloud1,
                                             pcl::PointCloud<pcl::PointXYZ>*& SVcloud2, pcl::PointCloud<pcl::PointXYZ>::Ptr CTcloud1,
                                             pcl::PointCloud<pcl::PointXYZ>::Ptr CTcloud2, pcl::PointCloud<pcl::PointXYZ>::Ptr BPcloud1,
                                             pcl::PointCloud<pcl::PointXYZ>::Ptr BPcloud2, std::vector<float> CTstd1, std::vector<float> BPstd2,
                                             float DTmin, float& currDT, float& BBchange_1, float& BBchange_2, Eigen::MatrixXd& VCM) {
    const int m1 = (int)CTcloud1->points.size(), m2 = (int)CTcloud2->points.size();
    auto flatten = [](pcl::PointCloud<pcl::PointXYZ>* sv, int m, std::vector<float>* pat, std::vector<int32_t>* off) {
        off->assign((size_t)m + 1, 0);
        for (int i = 0; i < m; ++i) (*off)[(size_t)i + 1] = (*off)[(size_t)i] + (int32_t)sv[i].points.size();
        pat->resize((size_t)((*off)[(size_t)m] > 0 ? (*off)[(size_t)m] : 1) * 4);
        for (int i = 0; i < m; ++i)
            if (!sv[i].points.empty()) std::memcpy(pat->data() + 4 * (size_t)(*off)[(size_t)i], sv[i].points.data(), sv[i].points.size() * 16);
    };
    std::vector<float> p1, p2;
    std::vector<int32_t> o1, o2;
    flatten(SVcloud1, m1, &p1, &o1);
    flatten(SVcloud2, m2, &p2, &o2);
    pwicp_params prm{Res1, Res2, SVRes1, SVRes2, 1, currDT, DTmin};
    pwicp_pair* pair = nullptr;
    if ((int)CTstd1.size() != m1 || (int)BPstd2.size() != m2 || (int)BPcloud1->points.size() != 6 * m1 || (int)BPcloud2->points.size() != 6 * m2)
        throw pwicp::Error(PWICP_E_INVALID, "PwICP_singleIteration: array sizes do not match the patch counts");
    pwicp::check(pwicp_pair_create_from_arrays(pwicp::thread_context(), pwicp::xyz4(*cloud1), (int)cloud1->points.size(), p1.data(), o1.data(), m1,
                                               pwicp::xyz4(*CTcloud1), pwicp::xyz4(*BPcloud1), nullptr, CTstd1.data(),
                                               pwicp::xyz4(*cloud2), (int)cloud2->points.size(), p2.data(), o2.data(), m2,
                                               pwicp::xyz4(*CTcloud2), pwicp::xyz4(*BPcloud2), BPstd2.data(), nullptr, &prm, &pair));
    pwicp_step st{};
    st.currDT = currDT; st.BBchange_1 = BBchange_1; st.BBchange_2 = BBchange_2;
    st.toStage2 = pwicp::toStage2(); st.toStage3 = pwicp::toStage3();
    const int rc = pwicp_pair_step(pair, &st);
    if (rc == PWICP_OK)
        pwicp_pair_download_state(pair, reinterpret_cast<float*>(cloud2->points.data()), reinterpret_cast<float*>(CTcloud2->points.data()),
                                  reinterpret_cast<float*>(BPcloud2->points.data()), p2.data());
    pwicp_pair_destroy(pair);
    pwicp::check(rc);                                  // (the reference calls std::exit here: R.cpp:728-731, 864-867)
    for (int i = 0; i < m2; ++i)
        if (!SVcloud2[i].points.empty()) std::memcpy(static_cast<void*>(SVcloud2[i].points.data()), p2.data() + 4 * (size_t)o2[(size_t)i], SVcloud2[i].points.size() * 16);
    currDT = st.currDT; BBchange_1 = st.BBchange_1; BBchange_2 = st.BBchange_2;
    pwicp::toStage2() = st.toStage2; pwicp::toStage3() = st.toStage3;
    if (st.toStage3) { VCM.resize(6, 6); for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) VCM(r, c) = st.VCM[6 * r + c]; }
    Eigen::Matrix4f T;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) T(r, c) = st.T16[4 * r + c];
    return T;
}

// Registration.h:74-78 (R.cpp:402-548)
inline bool Piecewise_ICP_4D(pcl::PointCloud<pcl::PointXYZ>::Ptr cloud1, pcl::PointCloud<pcl::PointXYZ>::Ptr cloud2, bool isSetResSVsize,
                             float Res1, float Res2, float SVsize1, float SVsize2, bool isManualDTinit, float DTinit, float DTmin,
                             std::string outfileIdx, Eigen::Matrix4f& transMat, std::vector<float>& transPara, Eigen::MatrixXd& VCM) {
    float T[16], para[6];
    double V[36];
    const int rc = pwicp_piecewise_icp_4d(pwicp::thread_context(), pwicp::xyz4(*cloud1), (int)cloud1->points.size(), pwicp::xyz4(*cloud2),
                                          (int)cloud2->points.size(), isSetResSVsize ? 1 : 0, Res1, Res2, SVsize1, SVsize2,
                                          isManualDTinit ? 1 : 0, DTinit, DTmin, outfileIdx.c_str(), T, para, V);
    if (rc != PWICP_OK) return false;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) transMat(r, c) = T[4 * r + c];
    transPara.assign(para, para + 6);
    VCM.resize(6, 6);
    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) VCM(r, c) = V[6 * r + c];
    return true;
}
// Registration.h:93-94 (R.cpp:552-589)
inline bool calAdaptivePairSequence(std::vector<std::string> fileNameList, int startEpoch, float DTinit, float ratioThd,
                                    std::map<int, int>& RegPairs, std::string adaptivePairFile) {
    std::vector<const char*> names;
    for (auto& f : fileNameList) names.push_back(f.c_str());
    const int n = (int)names.size() - startEpoch - 1;
    if (n <= 0) return false;
    std::vector<int32_t> tg((size_t)n);
    if (pwicp_adaptive_pair_sequence(pwicp::thread_context(), names.data(), (int)names.size(), startEpoch, DTinit, ratioThd, tg.data(),
                                     adaptivePairFile.c_str()) != PWICP_OK)
        return false;
    for (int k = 0; k < n; ++k) RegPairs.insert(std::make_pair(k + 1, (int)tg[(size_t)k]));
    return RegPairs.size() == fileNameList.size() - 1;                      // R.cpp:573-574
}
// Registration.h:127-129 (R.cpp:977-1153)
inline void calTransToReferenceEpoch(std::string transMatFile, int pairMode, std::string adaptivePairFile, int epochNum,
                                     std::string transMat2RefFile, std::string transPara2RefFile, std::vector<int>& timeStamp,
                                     std::vector<Eigen::Matrix4f>& allTransMat2Ref, std::vector<Eigen::MatrixXd>& allVCM2Ref) {
    std::vector<int32_t> st((size_t)(epochNum > 0 ? epochNum : 1));
    std::vector<float> T((size_t)(epochNum > 0 ? epochNum : 1) * 16);
    std::vector<double> V((size_t)(epochNum > 0 ? epochNum : 1) * 36);
    const int rc = pwicp_trans_to_reference_epoch(transMatFile.c_str(), pairMode, adaptivePairFile.c_str(), epochNum, transMat2RefFile.c_str(),
                                                  transPara2RefFile.c_str(), st.data(), T.data(), V.data());
    if (rc != PWICP_OK) throw pwicp::Error(rc, "calTransToReferenceEpoch: cannot read / write the transformation files");
    for (int i = 0; i < epochNum; ++i) {
        timeStamp.push_back(st[(size_t)i]);
        Eigen::Matrix4f M;
        Eigen::MatrixXd C(6, 6);
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) M(r, c) = T[16 * (size_t)i + 4 * r + c];
        for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) C(r, c) = V[36 * (size_t)i + 6 * r + c];
        allTransMat2Ref.push_back(M);
        allVCM2Ref.push_back(C);
    }
}
// Registration.h:198-199 (R.cpp:1157-1251)
inline void calAbsErrorOfTransPara(std::string transMatFile, std::string GTtransMatFile, int allEpochNum, int startEpoch,
                                   std::string transParaErrorFile) {
    const int rc = pwicp_abs_error_of_trans_para(transMatFile.c_str(), GTtransMatFile.c_str(), allEpochNum, startEpoch, transParaErrorFile.c_str());
    if (rc != PWICP_OK) throw pwicp::Error(rc, "calAbsErrorOfTransPara: cannot read / write the files");
}
#endif
#endif
