// Compile-and-link check of include/pwicp/Registration.h without PCL: instantiates every facade template with plain
// structs that have the layout of pcl::PointXYZ / pcl::PointNormal / Eigen-style (r, c) matrices.  With a GPU it also
// runs a tiny registration (argument "run").
#include <cstdio>
#include <cmath>
#include <cstring>
#include <vector>

#include "pwicp/Registration.h"

struct PointXYZ { float x, y, z, pad; };
struct PointNormal { float x, y, z, pad, nx, ny, nz, pad2, curvature, pad3[3]; };
template <class P> struct Cloud { std::vector<P> points; };
template <int R, int C, class T> struct Mat {
    T v[R * C] = {};
    T& operator()(int r, int c) { return v[r * C + c]; }
};

int main(int argc, char** argv) {
    const bool run = argc > 1 && !std::strcmp(argv[1], "run");
    Cloud<PointXYZ> a, b;
    const int side = 120;
    for (int i = 0; i < side; ++i)
        for (int j = 0; j < side; ++j) {
            const float x = i * 0.005f, y = j * 0.005f, z = 0.02f * std::sin(9 * x) * std::cos(7 * y);
            a.points.push_back({x, y, z, 1.f});
            b.points.push_back({x + 0.002f, y - 0.001f, z + 0.0015f, 1.f});
        }
    Cloud<PointNormal> an, bn;
    for (auto& p : a.points) an.points.push_back({p.x, p.y, p.z, 1.f, 0.f, 0.f, 1.f, 0.f, 0.f, {0, 0, 0}});
    for (auto& p : b.points) bn.points.push_back({p.x, p.y, p.z, 1.f, 0.f, 0.f, 1.f, 0.f, 0.f, {0, 0, 0}});
    if (!run) {
        // reference every instantiation so that the templates are compiled and linked, without needing a device
        void (*f1)(const Cloud<PointXYZ>&, Cloud<PointXYZ>&, bool, float, float, float, float, bool, float, float,
                   std::vector<float>&, Mat<4, 4, float>&, Mat<6, 6, double>&) = &pwicp::Piecewise_ICP;
        double (*f2)(const Cloud<PointXYZ>&, const Cloud<PointXYZ>&, float) = &pwicp::calPercentileDistBetween2PC;
        float (*f3)(const Cloud<PointXYZ>&, const Cloud<PointXYZ>&, float) = &pwicp::calOverlapRatioByC2Cdist;
        bool (*f4)(const Cloud<PointXYZ>&, float&, float&, float&) = &pwicp::calPatchNormal;
        Mat<4, 4, float> (*f5)(const Cloud<PointNormal>&, const Cloud<PointNormal>&, double) =
            &pwicp::P2PICPwithPatchNormal<Cloud<PointNormal>, Mat<4, 4, float>>;
        void (*f6)(const Cloud<PointXYZ>&, const Cloud<PointNormal>&, const Cloud<PointXYZ>&, Mat<6, 6, double>&) = &pwicp::calTransParaVCM;
        void (*f7)(const Cloud<PointXYZ>&, Cloud<PointXYZ>&, bool, float, int, double) = &pwicp::PCpreprocessing;
        float (*f8)(const Cloud<PointXYZ>&) = &pwicp::calPCresolution;
        int (*f9)(const Cloud<PointXYZ>&, float, float, Cloud<PointXYZ>&, Cloud<PointXYZ>&, std::vector<Cloud<PointXYZ>>&,
                  std::vector<float>&, std::vector<float>&) = &pwicp::PatchGenerationAndRefinement;
        std::printf("facade templates instantiated: %d\n", (f1 && f2 && f3 && f4 && f5 && f6 && f7 && f8 && f9) ? 9 : 0);
        return 0;
    }
    std::vector<float> DT;
    Mat<4, 4, float> T;
    Mat<6, 6, double> V;
    pwicp::Piecewise_ICP(a, b, true, 0.005f, 0.005f, 0.05f, 0.05f, true, 0.05f, 0.004f, DT, T, V);
    std::printf("facade run: outer iterations %d, t = (%g %g %g)\n", (int)DT.size() - 1, T(0, 3), T(1, 3), T(2, 3));
    const bool ok = std::fabs(T(0, 3) + 0.002f) < 5e-4f && std::fabs(T(1, 3) - 0.001f) < 5e-4f && std::fabs(T(2, 3) + 0.0015f) < 5e-4f;
    // the stages before the loop through the facade
    Cloud<PointXYZ> pre, ct, bpc;
    pwicp::PCpreprocessing(a, pre, true, 0.005f, 14, 5.0);
    const float spacing = pwicp::calPCresolution(pre);
    std::vector<Cloud<PointXYZ>> patches;
    std::vector<float> sbp, sct;
    const int m = pwicp::PatchGenerationAndRefinement(pre, 0.05f, 0.005f, ct, bpc, patches, sbp, sct);
    std::printf("facade stages: %zu -> %zu points, spacing %g, %d patches\n", a.points.size(), pre.points.size(), (double)spacing, m);
    const bool ok2 = pre.points.size() > 1000 && pre.points.size() <= a.points.size() && spacing > 0.003f && spacing < 0.008f &&
                     m > 20 && (int)ct.points.size() == m && (int)bpc.points.size() == 6 * m && (int)patches.size() == m &&
                     patches[0].points.size() > 4 && (int)sbp.size() == m;
    std::printf((ok && ok2) ? "FACADE_OK\n" : "FACADE_MISMATCH\n");
    return (ok && ok2) ? 0 : 1;
}
