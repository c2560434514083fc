// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
//
// Driver that compiles the *reference's own* vendored segmentation front end
// (header-only `codelibrary/`, included from /root/reference where it lies —
// nothing is copied into this repo) into oracle/_ref/libref_frontend.so.
//
// It drives the headers exactly the way the reference does in
//   src/Segmentation.cpp:18-68   (kNN-45 -> PCAEstimateNormal -> SupervoxelSegmentation)
// with the distance metric of include/Segmentation.h:362-375 (VCCSMetric) restated
// here because that header drags in PCL (absent in this image).
//
// Used only by tests / fixture generators to (a) produce supervoxel labels for the
// golden end-to-end vectors and (b) check the product's own front end.
// Build: see oracle/Makefile (target `ref`).  Needs /root/reference at build time only.

#include <climits>
#include <cmath>
#include <cstring>
#include <memory>   // std::uninitialized_* (MSVC pulls it in transitively; libstdc++ does not)
#include <numeric>  // std::iota, same reason

#include "codelibrary/base/array.h"
#include "codelibrary/geometry/kernel/point_3d.h"
#include "codelibrary/geometry/util/distance_3d.h"   // as include/Segmentation.h:19
#include "codelibrary/geometry/point_cloud/pca_estimate_normals.h"
#include "codelibrary/geometry/point_cloud/supervoxel_segmentation.h"
#include "codelibrary/util/tree/kd_tree.h"

namespace {

// include/Segmentation.h:25-28
struct OrientedPoint : cl::RPoint3D {
    OrientedPoint() {}
    cl::RVector3D normal;
};

// include/Segmentation.h:362-375
class Metric {
public:
    explicit Metric(double resolution) : resolution_(resolution) {}
    double operator()(const OrientedPoint& p1, const OrientedPoint& p2) const {
        return 1.0 - std::fabs(p1.normal * p2.normal) +
               cl::geometry::Distance(p1, p2) / resolution_ * 0.4;
    }
private:
    double resolution_;
};

}  // namespace

extern "C" {

// xyz: n points, `stride` floats apart (3 or 4).  labels_out[n]: supervoxel id per point.
// normals_out (optional, n*3 doubles), neighbors_out (optional, n*knn ints).
// Returns the number of supervoxels, or <0 on error.
int ref_frontend_run(const float* xyz, int n, int stride, double sv_resolution, int knn,
                     int* labels_out, double* normals_out, int* neighbors_out)
{
    if (!xyz || n <= knn || knn <= 0 || !labels_out) return -1;

    // S.cpp:18-23
    cl::Array<cl::RPoint3D> points;
    for (int i = 0; i < n; ++i)
        points.emplace_back(xyz[(size_t)i * stride + 0], xyz[(size_t)i * stride + 1],
                            xyz[(size_t)i * stride + 2]);

    // S.cpp:30-46
    cl::KDTree<cl::RPoint3D> kdtree;
    kdtree.SwapPoints(&points);
    cl::Array<cl::RVector3D> normals(n);
    cl::Array<cl::Array<int> > neighbors(n);
    cl::Array<cl::RPoint3D> neighbor_points(knn);
    for (int i = 0; i < n; ++i) {
        kdtree.FindKNearestNeighbors(kdtree.points()[i], knn, &neighbors[i]);
        for (int k = 0; k < knn; ++k) neighbor_points[k] = kdtree.points()[neighbors[i][k]];
        cl::geometry::point_cloud::PCAEstimateNormal(neighbor_points.begin(),
                                                      neighbor_points.end(), &normals[i]);
    }
    kdtree.SwapPoints(&points);

    // S.cpp:51-68
    Metric metric(sv_resolution);
    cl::Array<int> supervoxels, labels;
    cl::Array<OrientedPoint> oriented(n);
    for (int i = 0; i < n; ++i) {
        oriented[i].x = points[i].x;
        oriented[i].y = points[i].y;
        oriented[i].z = points[i].z;
        oriented[i].normal = normals[i];
    }
    cl::geometry::point_cloud::SupervoxelSegmentation(oriented, neighbors, sv_resolution, metric,
                                                      &supervoxels, &labels);

    for (int i = 0; i < n; ++i) labels_out[i] = labels[i];
    if (normals_out)
        for (int i = 0; i < n; ++i) {
            normals_out[3 * (size_t)i + 0] = normals[i].x;
            normals_out[3 * (size_t)i + 1] = normals[i].y;
            normals_out[3 * (size_t)i + 2] = normals[i].z;
        }
    if (neighbors_out)
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < knn; ++k) neighbors_out[(size_t)i * knn + k] = neighbors[i][k];
    return supervoxels.size();
}

}  // extern "C"
