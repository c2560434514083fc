// TEST SHIM, not PCL: the layout of the two PCL point types on the path (PCL 1.8.1 point_types.hpp: PointXYZ = 16 bytes
// x,y,z + padding 1.0f; PointNormal = 48 bytes xyz pad | normal pad | curvature pad[3]), so that the block of
// include/pwicp/Registration.h with the reference's exact signatures is compiled and run without PCL installed.
#pragma once
namespace pcl {
struct alignas(16) PointXYZ {
    float x = 0, y = 0, z = 0, pad = 1.f;
    PointXYZ() = default;
    PointXYZ(float a, float b, float c) : x(a), y(b), z(c) {}
};
struct alignas(16) PointNormal {
    float x = 0, y = 0, z = 0, pad = 1.f;
    float normal_x = 0, normal_y = 0, normal_z = 0, pad_n = 0;
    float curvature = 0, pad_c[3] = {0, 0, 0};
};
static_assert(sizeof(PointXYZ) == 16 && sizeof(PointNormal) == 48, "PCL layouts");
}  // namespace pcl
