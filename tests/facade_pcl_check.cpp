// The block of include/pwicp/Registration.h with the reference's EXACT signatures (Registration.h, Segmentation.h,
// CommonFunc.h of yihui4d/Piecewise-ICP), compiled against the PCL / Eigen type shim of tests/shim and — with a GPU,
// argument "run <golden dir> <tmp dir>" — executed: every signature of SURVEY 8b's "C++ API to keep" is called once.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "pwicp/Registration.h"

#ifndef PWICP_HAVE_PCL
#error "the PCL-typed block was not compiled (shim not on the include path?)"
#endif

typedef pcl::PointCloud<pcl::PointXYZ> Cloud;
typedef pcl::PointCloud<pcl::PointNormal> NCloud;

static void write_pcd(const std::string& path, const Cloud& c) {
    std::ofstream o(path, std::ios::binary);
    o << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH " << c.size()
      << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << c.size() << "\nDATA binary\n";
    for (auto& p : c.points) o.write(reinterpret_cast<const char*>(&p), 12);
}

static Eigen::Matrix4f mul(const Eigen::Matrix4f& A, const Eigen::Matrix4f& B) {      // Eigen's float product order
    Eigen::Matrix4f R;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = A(i, 0) * B(0, j);
            s = s + A(i, 1) * B(1, j);
            s = s + A(i, 2) * B(2, j);
            s = s + A(i, 3) * B(3, j);
            R(i, j) = s;
        }
    return R;
}

int main(int argc, char** argv) {
    const bool run = argc > 3 && !std::strcmp(argv[1], "run");
    if (!run) {
        // take the address of every exact-signature function: compiled and linked without a device
        void* f[] = {(void*)&Piecewise_ICP, (void*)&PwICP_singleIteration, (void*)&P2PICPwithPatchNormal, (void*)&calTransParaVCM,
                     (void*)&Piecewise_ICP_4D, (void*)&calAdaptivePairSequence, (void*)&calOverlapRatioByC2Cdist,
                     (void*)&calTransToReferenceEpoch, (void*)&calAbsErrorOfTransPara, (void*)&PatchGenerationAndRefinement,
                     (void*)&calBPandCTSTD, (void*)&calPercentileDistBetween2PC, (void*)&calPatchNormal, (void*)&calPatchSTD,
                     (void*)&generateCentroidCloudWithPatchNormals, (void*)&matrix2angle, (void*)&calBoundingBoxCornerChange,
                     (void*)&PCpreprocessing, (void*)&SORfilter, (void*)&calPCresolution};
        int n = 0;
        for (void* p : f) n += p != nullptr;
        std::printf("exact signatures compiled: %d\n", n);
        return n == 20 ? 0 : 1;
    }
    const std::string gold = argv[2], tmp = argv[3];
    int bad = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAILED: %s (line %d)\n", #c, __LINE__); ++bad; } } while (0)
    Cloud::Ptr a(new Cloud), b(new Cloud);
    const int side = 260;
    const float r = 0.005f;
    for (int i = 0; i < side; ++i)
        for (int j = 0; j < side; ++j) {
            const float x = i * r + 0.3f * r * std::sin(12.9898f * i + 78.233f * j), y = j * r + 0.3f * r * std::cos(39.346f * i + 11.135f * j);
            const float z = 0.03f * std::sin(7 * x) * std::cos(5 * y) + 0.01f * std::sin(31 * x + 1) * std::sin(27 * y);
            a->push_back(pcl::PointXYZ(x, y, z));
            b->push_back(pcl::PointXYZ(x + 0.004f, y - 0.003f, z + 0.005f));
        }
    // ---- CommonFunc.h --------------------------------------------------------------------------------------------------
    Cloud::Ptr pa(new Cloud), pb(new Cloud), sa(new Cloud), sa2(new Cloud);
    PCpreprocessing(a, pa, true, r, 14, 5.0);
    PCpreprocessing(b, pb, true, r, 14, 5.0);
    PCpreprocessing(a, sa, false, r, 14, 5.0);
    SORfilter(a, sa2, 14, 5.0);
    EXPECT(pa->size() > 30000 && pa->size() <= a->size() && sa->size() == sa2->size() && sa->size() > pa->size() - 10);
    EXPECT(std::memcmp(sa->points.data(), sa2->points.data(), sa->size() * 16) == 0);
    const float res = calPCresolution(pa);
    EXPECT(res > 0.003f && res < 0.008f);
    const double p75 = calPercentileDistBetween2PC(pa, pb, 0.75f);
    EXPECT(p75 > 0.001 && p75 < 0.01);
    EXPECT(calOverlapRatioByC2Cdist(pa, pb, 0.05f) > 0.95f);
    Eigen::Matrix4f M = Eigen::Matrix4f::Identity();
    M(0, 3) = 0.1f; M(0, 0) = std::cos(0.01f); M(0, 1) = -std::sin(0.01f); M(1, 0) = std::sin(0.01f); M(1, 1) = std::cos(0.01f);
    Eigen::Vector3f ang;
    matrix2angle(M, ang);
    EXPECT(std::fabs(ang[2] - 0.01f) < 1e-6f && std::fabs(ang[0]) < 1e-7f && std::fabs(ang[1]) < 1e-7f);
    const double bb[6] = {0, 0, 0, 1, 1, 1};
    EXPECT(std::fabs(calBoundingBoxCornerChange(bb, M) - 0.1f) < 1e-6f);        // the min corner moves by t = 0.1, the max corner by less
    // ---- Segmentation.h ------------------------------------------------------------------------------------------------
    Cloud::Ptr ct1(new Cloud), bp1(new Cloud), ct2(new Cloud), bp2(new Cloud);
    Cloud *sv1 = nullptr, *sv2 = nullptr;
    const int m1 = PatchGenerationAndRefinement(pa, 10 * r, ct1, bp1, sv1, false);
    const int m2 = PatchGenerationAndRefinement(pb, 10 * r, ct2, bp2, sv2, false);
    EXPECT(m1 > 200 && m2 > 200 && (int)ct1->size() == m1 && (int)bp1->size() == 6 * m1 && sv1[0].size() >= 20);
    std::vector<float> bpstd1, ctstd1, bpstd2, ctstd2;
    calBPandCTSTD(sv1, m1, bpstd1, ctstd1);
    calBPandCTSTD(sv2, m2, bpstd2, ctstd2);
    Cloud::Ptr one(new Cloud(sv1[3]));
    EXPECT(calPatchSTD(one) == bpstd1[3] && ctstd1[3] == bpstd1[3] / (float)sv1[3].size());
    float nx, ny, nz;
    EXPECT(calPatchNormal(sv1[3], nx, ny, nz) && std::fabs(nx * nx + ny * ny + nz * nz - 1.f) < 1e-5f);
    NCloud::Ptr ctn1(new NCloud), ctn2(new NCloud);
    generateCentroidCloudWithPatchNormals(ct1, sv1, ctn1);
    generateCentroidCloudWithPatchNormals(ct2, sv2, ctn2);
    EXPECT((int)ctn1->size() == m1 && ctn1->points[3].normal_x == nx && ctn1->points[3].normal_z == nz && ctn1->points[3].x == ct1->points[3].x);
    // ---- Registration.h: inner ICP, VCM --------------------------------------------------------------------------------
    const Eigen::Matrix4f Ti = P2PICPwithPatchNormal(ctn1, ctn2, 1e-6);
    EXPECT(std::fabs(Ti(2, 3) + 0.005f) < 1e-3f);
    const Eigen::MatrixXd V0 = calTransParaVCM(ct1, ctn1, ct2);
    EXPECT(V0.rows() == 6 && V0(3, 3) > 0);
    // ---- Piecewise_ICP == the loop of R.cpp:668-700 written with PwICP_singleIteration ---------------------------------
    Cloud::Ptr pb_loop(new Cloud(*pb)), pb_whole(new Cloud(*pb));
    std::vector<float> DT;
    Eigen::Matrix4f Twhole;
    Eigen::MatrixXd Vwhole;
    Piecewise_ICP(pa, pb_whole, true, r, r, 10 * r, 10 * r, true, 10 * r, 0.8f * r, DT, Twhole, Vwhole);
    pwicp::toStage2() = 0; pwicp::toStage3() = 0;
    float currDT = 10 * r, BB1 = 0.f, BB2 = 0.f;
    Eigen::Matrix4f T = Eigen::Matrix4f::Identity();
    Eigen::MatrixXd Vloop;
    int iters = 0;
    while (!pwicp::toStage3() && iters < 50) {
        const Eigen::Matrix4f Tk = PwICP_singleIteration(pa, pb_loop, r, r, 10 * r, 10 * r, sv1, sv2, ct1, ct2, bp1, bp2, ctstd1, bpstd2, 0.8f * r,
                                                        currDT, BB1, BB2, Vloop);
        T = mul(Tk, T);
        ++iters;
        EXPECT(iters < (int)DT.size() && currDT == DT[(size_t)iters]);
    }
    EXPECT(iters == (int)DT.size() - 1);
    for (int k = 0; k < 16; ++k) EXPECT(T(k / 4, k % 4) == Twhole(k / 4, k % 4));
    for (int k = 0; k < 36; ++k) EXPECT(Vloop(k / 6, k % 6) == Vwhole(k / 6, k % 6));
    EXPECT(std::memcmp(pb_loop->points.data(), pb_whole->points.data(), pb_loop->size() * 16) == 0);
    EXPECT(std::fabs(T(0, 3) + 0.004f) < 5e-4f && std::fabs(T(1, 3) - 0.003f) < 5e-4f && std::fabs(T(2, 3) + 0.005f) < 5e-4f);
    delete[] sv1;
    delete[] sv2;
    // ---- Piecewise_ICP_4D + the 4D scheduling functions on files ----------------------------------------------------------
    Eigen::Matrix4f T4;
    std::vector<float> para;
    Eigen::MatrixXd V4;
    EXPECT(Piecewise_ICP_4D(a, b, true, r, r, 10 * r, 10 * r, true, 10 * r, 0.8f * r, tmp + "/p_", T4, para, V4));
    EXPECT(para.size() == 6 && std::fabs(T4(2, 3) + 0.005f) < 5e-4f && std::ifstream(tmp + "/p_TransMatrix.txt").good());
    std::vector<std::string> files = {tmp + "/Epoch_001.pcd", tmp + "/Epoch_002.pcd", tmp + "/Epoch_003.pcd"};
    write_pcd(files[0], *a); write_pcd(files[1], *b); write_pcd(files[2], *b);
    std::map<int, int> pairs;
    EXPECT(calAdaptivePairSequence(files, 0, 0.05f, 0.75f, pairs, tmp + "/RegPairFile.txt"));
    EXPECT(pairs.size() == 2 && pairs[1] == 0 && pairs[2] == 0);
    std::vector<int> stamps;
    std::vector<Eigen::Matrix4f> T2ref;
    std::vector<Eigen::MatrixXd> V2ref;
    calTransToReferenceEpoch(gold + "/TransMatrices.txt", 0, "", 19, tmp + "/toRef.txt", tmp + "/toRefPara.txt", stamps, T2ref, V2ref);
    EXPECT(stamps.size() == 19 && stamps[0] == 2 && T2ref.size() == 19 && V2ref[18].rows() == 6);
    calAbsErrorOfTransPara(tmp + "/toRef.txt", gold + "/defined_transformations.txt", 20, 0, tmp + "/err.txt");
    EXPECT(std::ifstream(tmp + "/err.txt").good());
    std::printf(bad ? "FACADE_PCL_MISMATCH (%d)\n" : "FACADE_PCL_OK\n", bad);
    return bad ? 1 : 0;
}
