// Host pieces of the preprocessing stage shared with the device pipeline (csrc/prep.hip).
#ifndef PWICP_HOST_PREPROCESS_H
#define PWICP_HOST_PREPROCESS_H
#include <cstddef>

namespace pwhost {

struct VoxelEntry { unsigned idx; int pt; };      // pcl::VoxelGrid's cloud_point_index_idx

// PWICP_VOXEL_ORDER: "msvc" (default) sums the points of a voxel in the order the std::sort of the reference's released build
// leaves them in (msvc_sort.h); "input" sums them in input order.
bool voxel_order_is_msvc();
// false: depth budget of the sort exhausted, order not reproduced (caller falls back to input order)
bool voxel_sort_msvc(VoxelEntry* e, size_t n);

}  // namespace pwhost
#endif
