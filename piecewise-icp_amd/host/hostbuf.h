// Large host arrays of the setup stages (k-NN graphs: 180 MB per 1 M points) without value initialisation and on
// transparent huge pages where the system offers them on request (madvise mode): first-touch page faults, not copies or
// arithmetic, dominate the cost of such buffers (≈0.3 ms per MB with 4 KiB pages on the measured hosts).
#ifndef PWICP_HOSTBUF_H
#define PWICP_HOSTBUF_H
#include <cstddef>
#include <cstdlib>
#if defined(__linux__)
#include <sys/mman.h>
#endif

namespace pwhost {

template <typename T>
struct HostBuf {
    T* p = nullptr;
    size_t n = 0;
    HostBuf() = default;
    HostBuf(const HostBuf&) = delete;
    HostBuf& operator=(const HostBuf&) = delete;
    ~HostBuf() { release(); }
    void release() { std::free(p); p = nullptr; n = 0; }
    // contents are NOT preserved and NOT initialised
    bool reserve(size_t count) {
        if (count <= n && p) return true;
        release();
        const size_t align = (size_t)2 << 20;
        size_t bytes = (count ? count : 1) * sizeof(T);
        bytes = (bytes + align - 1) / align * align;
        void* q = std::aligned_alloc(align, bytes);
        if (!q) return false;
#if defined(__linux__) && defined(MADV_HUGEPAGE)
        (void)madvise(q, bytes, MADV_HUGEPAGE);
#endif
        p = (T*)q;
        n = count;
        return true;
    }
    T* data() { return p; }
    const T* data() const { return p; }
    void swap(HostBuf& o) { T* tp = p; p = o.p; o.p = tp; size_t tn = n; n = o.n; o.n = tn; }
};

}  // namespace pwhost
#endif
