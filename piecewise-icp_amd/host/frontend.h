// Stages of the segmentation front end shared between the host pipeline (frontend.cpp) and the device pipeline
// (csrc/frontend.hip): S.cpp:18-68 = k-NN graph -> PCA normals -> supervoxel fusion -> boundary refinement -> relabel.
#pragma once

#include <cstdint>
#include <memory>
#include <vector>

namespace pwhost {

// one cache line per point: position and PCA normal (double, as in the reference's front end); same layout on the device
struct alignas(64) FePt {
    double x, y, z, nx, ny, nz;
};

// float -> double positions (S.cpp:18-22) and PCA normals (pca_estimate_normals.h:42-108) of all points, host threads
void fe_points_and_normals(const float* cloud_xyz4, int n, const int32_t* nb, int k, FePt* P);
// the eigen step of the PCA normals alone: S6 = xx xy xz yy yz zz of every neighbourhood (built on the device) -> unit normals
void fe_normals_from_scatter(const double* S6, int n, double* normals3);
// number of occupied cells of edge `resolution` (grid_sample.h:30-75) = the supervoxel count the fusion stops at
int fe_count_occupied_cells(const FePt* P, int n, double resolution);
// bounding box of n (x, y, z, -) points, as doubles (grid_sample.h:36-44), on the host threads
void fe_bounding_box(const float* xyz4, int n, double mn[3], double mx[3]);
// the serial fusion (supervoxel_segmentation.h:65-170): root point of every point and the roots in ascending order
int fe_fusion_host(const FePt* P, const int32_t* nb, int k, int n, double resolution, int n_supervoxels, std::vector<int>* root_of,
                   std::vector<int>* roots);

// the serial boundary refinement (supervoxel_segmentation.h:172-236) of root labels, in place
void fe_refine_host(const FePt* P, const int32_t* nb, int k, int n, double resolution, std::vector<int>* root_of);

}  // namespace pwhost

// csrc/frontend.hip: the whole front end of a cloud on the device - k-NN graph, neighbourhood scatter, occupied cells, fusion,
// refinement, relabel (cell_edge: of the k-NN search grid, <= 0 estimated)
struct pwicp_context;
int pw_frontend_segment_device(pwicp_context* ctx, const float* cloud_xyz4, int n, int k, float cell_edge, float sv_resolution,
                               int32_t* labels, int* n_supervoxels);
// frees the grow-only device / pinned work space the front end keeps with `ctx` (it is rebuilt by the next call)
void pw_frontend_release_workspace(pwicp_context* ctx);
// a slot the host stages may hang helpers on that should live as long as the context (destroyed with it)
std::shared_ptr<void>* pw_context_host_slot(pwicp_context* ctx);
