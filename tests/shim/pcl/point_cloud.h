// TEST SHIM, not PCL: pcl::PointCloud<T> reduced to what the reference's signatures use (points, width/height/is_dense,
// size/clear/push_back, Ptr).  PCL 1.8.1's Ptr is boost::shared_ptr; the facade only spells `::Ptr`.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
namespace pcl {
template <class T>
struct PointCloud {
    typedef std::shared_ptr<PointCloud<T>> Ptr;
    std::vector<T> points;
    uint32_t width = 0, height = 0;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    void clear() { points.clear(); width = height = 0; }
    void push_back(const T& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
    T& operator[](size_t i) { return points[i]; }
    const T& operator[](size_t i) const { return points[i]; }
};
}  // namespace pcl
