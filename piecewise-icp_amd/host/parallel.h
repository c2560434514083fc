// Host threads for the embarrassingly parallel parts of the setup stages (per-point normals, per-point minima, the voxel sort).
// Every index is computed by exactly one thread with the same scalar code as the serial loop, so results do not depend on the
// thread count.
//
// How many: host_threads() = the CPUs this PROCESS may really use - the affinity mask, cut by the cgroup's CPU quota (v2 cpu.max /
// v1 cfs_quota_us), divided by $LOCAL_WORLD_SIZE (the ranks of a multi-process run share the node's CPUs: 8 ranks x 5 front-end
// streams x 32 threads on a 16-CPU box was what round 4 would have started), at most 32, at least 1.  $PWICP_HOST_THREADS overrides.
// Who: ONE pool per process, created at the first use, host_threads() - 1 workers that live until the process ends (a std::thread
// per call was ~30 us each, and five front-end streams each brought their own set).  The caller of parallel_for works too, and
// while it waits for its last chunks it takes chunks of OTHER callers from the queue, so the streams of a series share the pool
// without idling and a pool of one thread (one rank of eight on 16 CPUs) is still correct.
#ifndef PWICP_HOST_PARALLEL_H
#define PWICP_HOST_PARALLEL_H
#include <sched.h>
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace pwhost {

inline int detect_host_threads() {
    if (const char* e = std::getenv("PWICP_HOST_THREADS")) {
        const int v = std::atoi(e);
        if (v > 0) return std::min(v, 256);
    }
    const unsigned hc = std::thread::hardware_concurrency();
    double n = hc ? (double)hc : 1.0;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = std::min(n, (double)CPU_COUNT(&set));
    // cgroup CPU quota: v2 "<quota|max> <period>", v1 two files
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long long per = 0;
        if (std::fscanf(f, "%63s %lld", q, &per) == 2 && per > 0 && q[0] != 'm') n = std::min(n, (double)std::atoll(q) / (double)per);
        std::fclose(f);
    } else {
        long long q = -1, per = 0;
        if (FILE* fq = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(fq, "%lld", &q) != 1) q = -1; std::fclose(fq); }
        if (FILE* fp = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(fp, "%lld", &per) != 1) per = 0; std::fclose(fp); }
        if (q > 0 && per > 0) n = std::min(n, (double)q / (double)per);
    }
    if (const char* e = std::getenv("LOCAL_WORLD_SIZE")) {
        const int lws = std::atoi(e);
        if (lws > 1) n /= (double)lws;
    }
    return (int)std::min(std::max(n + 0.5, 1.0), 32.0);
}

inline int host_threads() {
    static const int n = detect_host_threads();
    return n;
}

class Pool {
  public:
    static Pool& get() {
        static Pool* p = new Pool(host_threads());     // (never destructed: its workers sleep in wait() until the process ends)
        return *p;
    }
    int size() const { return nthreads_; }

    // fn(t) for t in [0, n): n - 1 of them offered to the pool, the caller runs the rest and - while its own are still out - whatever
    // else is queued.  Returns when all n have finished.
    template <typename F>
    void run(int n, F fn) {
        if (n <= 1) { if (n == 1) fn(0); return; }
        struct Job { int left; std::mutex mu; std::condition_variable cv; std::exception_ptr err; };      // `left` under `mu`
        Job job;
        job.left = n - 1;
        {
            std::lock_guard<std::mutex> g(mu_);
            for (int t = 1; t < n; ++t)
                q_.emplace_back([&job, &fn, t] {
                    std::exception_ptr e;
                    try { fn(t); } catch (...) { e = std::current_exception(); }      // (handed to the caller, who rethrows it)
                    std::lock_guard<std::mutex> g2(job.mu);          // (the unlock is this task's last touch of the Job)
                    if (e && !job.err) job.err = e;
                    if (--job.left == 0) job.cv.notify_all();
                });
        }
        cv_.notify_all();
        // The queued tasks hold references to `job` and `fn`, which live on this stack frame: whatever the caller's own share (or a
        // task it helps out with) throws - std::bad_alloc inside a parallel_for body - the frame must outlive every one of them.
        // The first exception is kept, all n are waited for, then it is thrown again.
        std::exception_ptr err;
        try { fn(0); } catch (...) { err = std::current_exception(); }
        for (;;) {
            {
                std::lock_guard<std::mutex> g(job.mu);
                if (job.left == 0) break;
            }
            std::function<void()> task;
            {
                std::lock_guard<std::mutex> g(mu_);
                if (!q_.empty()) { task = std::move(q_.front()); q_.pop_front(); }
            }
            if (task) {
                task();                       // (a task keeps its own exception for ITS caller: nothing escapes here)
                continue;
            }
            std::unique_lock<std::mutex> g(job.mu);
            job.cv.wait(g, [&job] { return job.left == 0; });
            break;
        }
        if (!err) err = job.err;               // (all tasks are done: no lock needed)
        if (err) std::rethrow_exception(err);
    }

  private:
    explicit Pool(int n) : nthreads_(std::max(n, 1)) {
        for (int t = 1; t < nthreads_; ++t) std::thread([this] { loop(); }).detach();
    }
    void loop() {
        for (;;) {
            std::function<void()> task;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return !q_.empty(); });
                task = std::move(q_.front());
                q_.pop_front();
            }
            task();
        }
    }
    int nthreads_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
};

// fn(lo, hi) over [0, n)
template <typename F>
void parallel_for(long long n, F fn, long long min_chunk = 4096) {
    const int nt = (int)std::min<long long>(host_threads(), (n + min_chunk - 1) / min_chunk);
    if (nt <= 1) { fn((long long)0, n); return; }
    Pool::get().run(nt, [&](int t) { fn(n * t / nt, n * (t + 1) / nt); });
}

}  // namespace pwhost
#endif
