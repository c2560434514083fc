// The order std::sort leaves EQUAL keys in, for the standard library the reference's released binaries were built with.
//
// Why this exists: pcl::VoxelGrid (PCL 1.8.1 filters/impl/voxel_grid.hpp applyFilter, called from PCpreprocessing,
// /root/reference/src/CommonFunc.cpp:423-434) sorts its (voxel index, point index) entries with an unstable std::sort whose
// comparator looks at the voxel index only, then sums the points of a voxel in the order the sort left them in - in float.
// With three or more points in a voxel the centroid's last bit depends on that order, the order is a property of the
// std::sort implementation, and the reference's checked-in results (results/4DPCReg/*, SURVEY F3: MSVC 14.10) carry it: with
// the points of a voxel summed in input order 2 of the 19 Direct2Ref pairs miss the result files by 1.8e-5 / 8.7e-4 rad
// (one near-threshold decision flips downstream), with this order all 19 agree to <= 3e-7 rad
// (tests/golden/oracle_vs_reference.json, tools/rootcause_golden.py).
//
// The algorithm restated here is the published structure of that library's std::sort: introsort with a depth budget that
// shrinks by 3/4 per level, a "fat" three-way partition around a median-of-three (ninther above 40 elements) pivot guess that
// gathers the pivot's equals in the middle, recursion into the smaller side, insertion sort at <= 32 elements; a sub-range
// whose depth budget runs out is heap-sorted (make_heap bottom-up through "hole to the bottom along the larger child, then push
// the value back up", then pop after pop) - restated below from the same library's published <algorithm>; unlike the
// introsort part, which the reference's 57 result files vouch for, no fixture reaches it (it takes an adversarial key
// sequence), so product and oracle are only held to EACH OTHER there (tests/test_host_stages.py, budget forced small).
// Entries are sorted in place; `less` must be a strict weak order.
#ifndef PWICP_HOST_MSVC_SORT_H
#define PWICP_HOST_MSVC_SORT_H
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

#include "parallel.h"

namespace pwhost {
namespace msvc_order {

constexpr std::ptrdiff_t kInsertionMax = 32;

template <typename T, typename Less>
inline void med3(T* a, T* b, T* c, Less& less) {
    if (less(*b, *a)) std::swap(*b, *a);
    if (less(*c, *b)) {
        std::swap(*c, *b);
        if (less(*b, *a)) std::swap(*b, *a);
    }
}

template <typename T, typename Less>
inline void guess_median(T* first, T* mid, T* last, Less& less) {       // last = the final element, not one past it
    if (40 < last - first) {
        const std::size_t step = (std::size_t)(last - first + 1) / 8;
        med3(first, first + step, first + 2 * step, less);
        med3(mid - step, mid, mid + step, less);
        med3(last - 2 * step, last - step, last, less);
        med3(first + step, mid, last - step, less);
    } else {
        med3(first, mid, last, less);
    }
}

// [first, last) -> (begin, end) of the run equal to the pivot; everything left of it is smaller, right of it larger
template <typename T, typename Less>
std::pair<T*, T*> partition(T* first, T* last, Less& less) {
    T* mid = first + (last - first) / 2;
    guess_median(first, mid, last - 1, less);
    T* pf = mid;
    T* pl = pf + 1;
    while (first < pf && !less(*(pf - 1), *pf) && !less(*pf, *(pf - 1))) --pf;
    while (pl < last && !less(*pl, *pf) && !less(*pf, *pl)) ++pl;
    T* gf = pl;
    T* gl = pf;
    for (;;) {
        for (; gf < last; ++gf) {
            if (less(*pf, *gf)) continue;
            if (less(*gf, *pf)) break;
            if (pl++ != gf) std::swap(*(pl - 1), *gf);
        }
        for (; first < gl; --gl) {
            if (less(*(gl - 1), *pf)) continue;
            if (less(*pf, *(gl - 1))) break;
            if (--pf != gl - 1) std::swap(*pf, *(gl - 1));
        }
        if (gl == first && gf == last) return {pf, pl};
        if (gl == first) {                  // no room at the bottom: rotate the pivot run upward
            if (pl != gf) std::swap(*pf, *pl);
            ++pl;
            std::swap(*pf++, *gf++);
        } else if (gf == last) {            // no room at the top: rotate the pivot run downward
            if (--gl != --pf) std::swap(*gl, *pf);
            std::swap(*pf, *--pl);
        } else {
            std::swap(*gf++, *--gl);
        }
    }
}

template <typename T, typename Less>
void insertion_sort(T* first, T* last, Less& less) {
    if (first == last) return;
    for (T* next = first; ++next != last;) {
        T* hole = next;
        T val = *next;
        if (less(val, *first)) {
            std::move_backward(first, next, next + 1);
            *first = val;
        } else {
            for (T* prev = hole; less(val, *--prev); hole = prev) *hole = *prev;
            *hole = val;
        }
    }
}

// _Pop_heap_hole_by_index: the hole sinks to the bottom along the larger child, then `val` is pushed up from there
template <typename T, typename Less>
void heap_hole(T* first, std::ptrdiff_t hole, std::ptrdiff_t bottom, T val, Less& less) {
    const std::ptrdiff_t top = hole;
    std::ptrdiff_t idx = hole;
    const std::ptrdiff_t max_non_leaf = (bottom - 1) / 2;
    while (idx < max_non_leaf) {
        idx = 2 * idx + 2;
        if (less(first[idx], first[idx - 1])) --idx;
        first[hole] = first[idx];
        hole = idx;
    }
    if (idx == max_non_leaf && bottom % 2 == 0) {          // only child at the bottom
        first[hole] = first[bottom - 1];
        hole = bottom - 1;
    }
    for (std::ptrdiff_t i = (hole - 1) / 2; top < hole && less(first[i], val); i = (hole - 1) / 2) {
        first[hole] = first[i];
        hole = i;
    }
    first[hole] = val;
}

// std::make_heap + std::sort_heap of that library (the fall-back of its std::sort)
template <typename T, typename Less>
void heap_sort(T* first, T* last, Less& less) {
    const std::ptrdiff_t bottom = last - first;
    for (std::ptrdiff_t hole = bottom / 2; 0 < hole;) {
        --hole;
        T val = first[hole];
        heap_hole(first, hole, bottom, val, less);
    }
    for (; 2 <= last - first; --last) {
        T val = *(last - 1);
        *(last - 1) = *first;
        heap_hole(first, (std::ptrdiff_t)0, last - 1 - first, val, less);
    }
}

// Task pool for the independent sides of a partition (the order of the result does not depend on the schedule: every
// sub-range is sorted by the same sequential procedure, whoever runs it).
template <typename T, typename Less>
class Sorter {
  public:
    Sorter(Less less, int threads) : less_(less), nthreads_(std::max(threads, 1)) {}

    // budget: the library's initial depth budget is the element count; a smaller one (tests) makes sub-ranges take the
    // heap-sort fall-back.  Always returns true (kept for the callers' documented fall-back, which nothing reaches any more).
    bool sort(T* first, T* last, std::ptrdiff_t budget = -1) {
        ok_.store(true);
        if (budget < 0) budget = last - first;
        if (nthreads_ == 1 || last - first < kParallelMin) {
            run(first, last, budget, false);
            return ok_.load();
        }
        pending_.store(1);
        push(Task{first, last, budget});
        // (only sub-ranges of kParallelMin elements or more become tasks: more workers than that never have work, and starting a
        // thread is ~30 us - 32 of them for a 140 k-element sort were a quarter of its 4 ms)
        const int nt = (int)std::min<std::ptrdiff_t>(nthreads_, std::max<std::ptrdiff_t>(1, (last - first) / kParallelMin));
        // (on the process's pool, host/parallel.h: a late starter finds nothing pending and returns)
        pwhost::Pool::get().run(nt, [this](int) { worker(); });
        return ok_.load();
    }

  private:
    struct Task { T* first; T* last; std::ptrdiff_t ideal; };
    static constexpr std::ptrdiff_t kParallelMin = 1 << 15;

    void push(const Task& t) {
        { std::lock_guard<std::mutex> g(mu_); queue_.push_back(t); }
        cv_.notify_one();
    }

    void worker() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return !queue_.empty() || pending_.load() == 0; });
                if (queue_.empty()) return;
                t = queue_.back();
                queue_.pop_back();
            }
            run(t.first, t.last, t.ideal, true);
            if (pending_.fetch_sub(1) == 1) { std::lock_guard<std::mutex> g(mu_); cv_.notify_all(); }
        }
    }

    void run(T* first, T* last, std::ptrdiff_t ideal, bool spawn) {
        std::ptrdiff_t count;
        while (kInsertionMax < (count = last - first) && 0 < ideal) {
            const std::pair<T*, T*> mid = partition(first, last, less_);
            ideal /= 2;
            ideal += ideal / 2;
            T *of, *ol;                      // the side the library recurses into (the smaller one); it loops on the other
            if (mid.first - first < last - mid.second) { of = first; ol = mid.first; first = mid.second; }
            else { of = mid.second; ol = last; last = mid.first; }
            if (spawn && ol - of >= kParallelMin) {
                pending_.fetch_add(1);
                push(Task{of, ol, ideal});
            } else {
                run(of, ol, ideal, spawn);
            }
        }
        if (kInsertionMax < count) heap_sort(first, last, less_);
        else if (2 <= count) insertion_sort(first, last, less_);
    }

    Less less_;
    int nthreads_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Task> queue_;
    std::atomic<long long> pending_{0};
    std::atomic<bool> ok_{true};
};

template <typename T, typename Less>
bool sort(T* first, T* last, Less less, int threads = 1, std::ptrdiff_t budget = -1) {
    Sorter<T, Less> s(less, threads);
    return s.sort(first, last, budget);
}

}  // namespace msvc_order
}  // namespace pwhost
#endif
