// Static range split over host threads for the embarrassingly parallel parts of the setup stages (per-point normals,
// per-point minima).  Every index is computed by exactly one thread with the same scalar code as the serial loop, so
// results do not depend on the thread count.  PWICP_HOST_THREADS overrides the default (hardware threads, <= 32).
#ifndef PWICP_HOST_PARALLEL_H
#define PWICP_HOST_PARALLEL_H
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <vector>

namespace pwhost {

inline int host_threads() {
    if (const char* e = std::getenv("PWICP_HOST_THREADS")) {
        const int v = std::atoi(e);
        if (v > 0) return std::min(v, 256);
    }
    const unsigned hc = std::thread::hardware_concurrency();
    return (int)std::min<unsigned>(hc ? hc : 1u, 32u);
}

// fn(lo, hi) over [0, n)
template <typename F>
void parallel_for(long long n, F fn, long long min_chunk = 4096) {
    int nt = (int)std::min<long long>(host_threads(), (n + min_chunk - 1) / min_chunk);
    if (nt <= 1) { fn((long long)0, n); return; }
    std::vector<std::thread> th;
    th.reserve((size_t)nt);
    for (int t = 0; t < nt; ++t) {
        const long long lo = n * t / nt, hi = n * (t + 1) / nt;
        th.emplace_back([=] { fn(lo, hi); });
    }
    for (auto& t : th) t.join();
}

}  // namespace pwhost
#endif
