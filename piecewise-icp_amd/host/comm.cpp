// The one exchange of a multi-process 4D series — the all-gather of the 384-byte pair records (and, in adaptive mode, the
// broadcast of the pair map) — over RCCL directly: one process per GPU, no Python, no torch.  librccl.so is opened with
// dlopen on first use, so single-GPU users never load it.
//
// Reference: the pair loop of PiecewiseICP_4D_call (src/Registration.cpp:89-187) is sequential in one process; its
// iterations are independent, which is what this shards: pair p -> rank p mod world, results gathered once.
// Rendezvous: the ncclUniqueId travels through a file (single node).  Rank 0 removes whatever is left under that name,
// creates <id_file>.tmp exclusively (O_EXCL | O_NOFOLLOW, mode 0600), writes {magic, job token, time, id} and renames it;
// the other ranks poll and accept only a file that carries THEIR job token and a fresh time stamp, so a file left behind by a
// crashed run or written by another job is never taken for this job's id.  The job token is what all ranks of one launch
// share: $PWICP_JOB_ID, else $TORCHELASTIC_RUN_ID + the launcher's pid (getppid()).  Rank 0 removes the file on every failure
// path of the initialisation and in pwicp_comm_destroy.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <ctime>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "pwicp.h"
#include "../csrc/pwicp_internal.h"

namespace {

struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (h) return true;
        for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) { std::cerr << "Error: librccl.so not found (" << dlerror() << ")\n"; return false; }
        GetUniqueId = (decltype(GetUniqueId))dlsym(h, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(h, "ncclCommInitRank");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        Broadcast = (decltype(Broadcast))dlsym(h, "ncclBroadcast");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(h, "ncclGetErrorString");
        return GetUniqueId && CommInitRank && AllGather && Broadcast && CommDestroy;
    }
};
Rccl g_rccl;

// what every rank of one launch shares and no other launch does
uint64_t job_token() {
    std::string t;
    // launcher-agnostic: nothing of the process tree goes in (ranks started through per-rank wrappers, `mpirun bash -c`, a
    // container exec have different parents); $PWICP_JOB_ID tells two launches apart that share address, port and world size
    if (const char* e = std::getenv("PWICP_JOB_ID")) t = e;
    else if (const char* r = std::getenv("TORCHELASTIC_RUN_ID")) t = r;
    for (const char* k : {"MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE"})
        if (const char* e = std::getenv(k)) { t += "/"; t += e; }
    uint64_t h = 1469598103934665603ull;                 // FNV-1a
    for (unsigned char c : t) { h ^= c; h *= 1099511628211ull; }
    return h;
}

struct IdFile {                  // content of the rendezvous file
    uint64_t magic, token;
    int64_t written_at;          // seconds since the epoch
    ncclUniqueId id;
};
constexpr uint64_t kIdMagic = 0x5057494350494431ull;     // "PWICPID1"
constexpr int64_t kIdMaxAgeSeconds = 600;
// The ranks of one launch are started by one launcher within seconds of each other, and rank 0 writes the file after ITS start: a file
// written more than this long before the reading process started belongs to an earlier launch - also when its token is the same
// (torchrun's static rendezvous: TORCHELASTIC_RUN_ID "none", default port, same world size, no $PWICP_JOB_ID) and it is younger than
// kIdMaxAgeSeconds (an earlier launch that was killed a minute ago).
constexpr int64_t kLauncherStaggerSeconds = 30;
// How far apart the processes of ONE launch may start: 30 s by default, $PWICP_ID_STAGGER_S for launchers that start ranks minutes
// apart (srun / mpirun behind container pulls, ranks started by hand).  With $PWICP_JOB_ID set the token is unique per launch by
// the user's own word, an id file that carries it IS this launch's, and the start-time rule is not applied at all (a
// rank that started later than the stagger after rank 0's write used to refuse a valid id until the rendezvous timed out).
int64_t launcher_stagger_seconds() {
    static const int64_t v = [] {
        if (std::getenv("PWICP_JOB_ID")) return (int64_t)-1;
        if (const char* e = std::getenv("PWICP_ID_STAGGER_S")) return (int64_t)std::max(atoll(e), 0ll);
        return kLauncherStaggerSeconds;
    }();
    return v;
}

// start of this process, seconds since the epoch: /proc/self/stat field 22 (clock ticks since boot) + btime of /proc/stat; without
// procfs the first call into this file stands in for it
int64_t process_start_epoch() {
    static const int64_t t = [] {
        const int64_t fallback = (int64_t)std::time(nullptr);
        std::ifstream st("/proc/self/stat");
        std::string line;
        if (!st || !std::getline(st, line)) return fallback;
        const size_t rp = line.rfind(')');              // the command name may hold blanks and brackets
        if (rp == std::string::npos) return fallback;
        std::istringstream is(line.substr(rp + 1));
        std::string tok;
        unsigned long long ticks = 0;
        for (int field = 3; field <= 22 && (is >> tok); ++field)
            if (field == 22) ticks = std::strtoull(tok.c_str(), nullptr, 10);
        if (!ticks) return fallback;
        std::ifstream ps("/proc/stat");
        int64_t btime = 0;
        while (ps && std::getline(ps, line))
            if (line.compare(0, 6, "btime ") == 0) { btime = std::strtoll(line.c_str() + 6, nullptr, 10); break; }
        const long hz = sysconf(_SC_CLK_TCK);
        if (btime <= 0 || hz <= 0) return fallback;
        return std::min<int64_t>(btime + (int64_t)(ticks / (unsigned long long)hz), fallback);
    }();
    return t;
}

bool write_id_file(const std::string& path, const ncclUniqueId& id, int64_t written_at = 0) {
    const std::string tmp = path + ".tmp";
    (void)unlink(path.c_str());
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    if (fd < 0) return false;
    IdFile f;
    std::memset(&f, 0, sizeof(f));
    f.magic = kIdMagic; f.token = job_token(); f.written_at = written_at ? written_at : (int64_t)std::time(nullptr); f.id = id;
    const bool ok = write(fd, &f, sizeof(f)) == (ssize_t)sizeof(f);
    close(fd);
    if (!ok || std::rename(tmp.c_str(), path.c_str()) != 0) { (void)unlink(tmp.c_str()); return false; }
    return true;
}

// 1: taken, 0: nothing (acceptable) there yet
int read_id_file(const std::string& path, ncclUniqueId* id) {
    const int fd = open(path.c_str(), O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
    if (fd < 0) return 0;
    IdFile f;
    const bool ok = read(fd, &f, sizeof(f)) == (ssize_t)sizeof(f);
    close(fd);
    if (ok && f.magic == kIdMagic && f.token != job_token()) {
        static bool told = false;
        if (!told) {
            told = true;
            std::cerr << "pwicp: " << path << " holds the id of another launch (job token differs) - waiting for this launch's; if "
                         "the ranks of ONE launch see different $PWICP_JOB_ID / $TORCHELASTIC_RUN_ID / MASTER_ADDR / MASTER_PORT / "
                         "WORLD_SIZE, export one PWICP_JOB_ID for all of them\n";
        }
    }
    if (!ok || f.magic != kIdMagic || f.token != job_token()) return 0;                 // another job's, or a torn write
    if ((int64_t)std::time(nullptr) - f.written_at > kIdMaxAgeSeconds) return 0;        // left behind by an earlier run
    if (launcher_stagger_seconds() >= 0 && f.written_at < process_start_epoch() - launcher_stagger_seconds()) {    // ... also one with THIS launch's token
        static bool told_stale = false;
        if (!told_stale) {
            told_stale = true;
            std::cerr << "pwicp: " << path << " was written " << (process_start_epoch() - f.written_at) << " s before this process started: "
                         "an earlier launch's id (same job token) - waiting for rank 0 to replace it ($PWICP_ID_STAGGER_S widens the "
                         "allowed start-time spread of one launch, $PWICP_JOB_ID switches the rule off)\n";
        }
        return 0;
    }
    *id = f.id;
    return 1;
}

}  // namespace

struct pwicp_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    std::string id_file;
    void* stage = nullptr;       // device staging of the host-buffer collectives, grow-only
    size_t stage_bytes = 0;
    void* staging(size_t bytes) {
        if (bytes <= stage_bytes) return stage;
        if (stage) { (void)hipFree(stage); stage = nullptr; stage_bytes = 0; }
        const size_t want = std::max<size_t>(bytes, 1 << 16);
        if (hipMalloc(&stage, want) != hipSuccess) { stage = nullptr; return nullptr; }
        stage_bytes = want;
        return stage;
    }
};

extern "C" {

PWICP_API int pwicp_comm_init(int rank, int world, int device, const char* id_file, pwicp_comm** out) {
    if (!out || world < 1 || rank < 0 || rank >= world || (world > 1 && (!id_file || !*id_file))) return PWICP_E_INVALID;
    *out = nullptr;
    if (!g_rccl.load()) return PWICP_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return PWICP_E_NO_DEVICE;
    ncclUniqueId id;
    std::memset(&id, 0, sizeof(id));
    const std::string path = id_file ? id_file : "";
    if (rank == 0) {
        if (world > 1) { (void)unlink(path.c_str()); (void)unlink((path + ".tmp").c_str()); }     // nothing stale may be picked up
        if (g_rccl.GetUniqueId(&id) != ncclSuccess) return PWICP_E_NO_DEVICE;
        if (world > 1 && !write_id_file(path, id)) {
            std::cerr << "Error: cannot create the RCCL id file " << path << "\n";
            return PWICP_E_INTERNAL;
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        while (!read_id_file(path, &id)) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
                std::cerr << "Error: rank " << rank << " never saw this job's RCCL id file " << path << "\n";
                return PWICP_E_INTERNAL;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    pwicp_comm* c = new pwicp_comm;
    c->rank = rank; c->world = world; c->device = device; c->id_file = path;
    const ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        std::cerr << "Error: ncclCommInitRank failed: " << (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?") << "\n";
        if (rank == 0 && world > 1) (void)unlink(path.c_str());
        delete c;
        return PWICP_E_NO_DEVICE;
    }
    if (hipStreamCreate(&c->stream) != hipSuccess) {
        g_rccl.CommDestroy(c->comm);
        if (rank == 0 && world > 1) (void)unlink(path.c_str());
        delete c;
        return PWICP_E_NO_DEVICE;
    }
    *out = c;
    return PWICP_OK;
}

// Test hook of the rendezvous file (no GPU, no RCCL): op 0 writes an id file (zero id, this process's job token) dated `age_s` seconds
// back, op 1 reads it as a rank != 0 would - 1: accepted, 0: not (yet) acceptable.
PWICP_API int pwicp_comm_debug_id_file(const char* path, int op, long age_s) {
    if (!path || !*path) return PWICP_E_INVALID;
    ncclUniqueId id;
    std::memset(&id, 0, sizeof(id));
    if (op == 0) return write_id_file(path, id, (int64_t)std::time(nullptr) - (int64_t)age_s) ? 1 : 0;
    return read_id_file(path, &id);
}

PWICP_API void pwicp_comm_destroy(pwicp_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->comm) g_rccl.CommDestroy(c->comm);
    if (c->stage) (void)hipFree(c->stage);
    if (c->rank == 0 && c->world > 1 && !c->id_file.empty()) std::remove(c->id_file.c_str());
    delete c;
}

PWICP_API int pwicp_comm_rank(const pwicp_comm* c) { return c ? c->rank : -1; }
PWICP_API int pwicp_comm_world(const pwicp_comm* c) { return c ? c->world : 0; }

// recv holds world * bytes; host buffers, staged through device memory (RCCL moves device memory over xGMI)
PWICP_API int pwicp_comm_allgather(pwicp_comm* c, const void* send, size_t bytes, void* recv) {
    if (!c || !send || !recv || bytes == 0) return PWICP_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return PWICP_E_NO_DEVICE;
    // one staging block: [send | recv]; 256-byte aligned halves
    const size_t off = (bytes + 255) / 256 * 256;
    char* st = (char*)c->staging(off + bytes * (size_t)c->world);
    if (!st) return PWICP_E_NOMEM;
    void *ds = st, *dr = st + off;
    int rc = PWICP_OK;
    if (hipMemcpyAsync(ds, send, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = PWICP_E_NO_DEVICE;
    if (rc == PWICP_OK && g_rccl.AllGather(ds, dr, bytes, ncclChar, c->comm, c->stream) != ncclSuccess) rc = PWICP_E_INTERNAL;
    if (rc == PWICP_OK && hipMemcpyAsync(recv, dr, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = PWICP_E_NO_DEVICE;
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == PWICP_OK) rc = PWICP_E_NO_DEVICE;
    return rc;
}

PWICP_API int pwicp_comm_broadcast(pwicp_comm* c, void* buf, size_t bytes, int root) {
    if (!c || !buf || bytes == 0 || root < 0 || root >= c->world) return PWICP_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return PWICP_E_NO_DEVICE;
    void* d = c->staging(bytes);
    if (!d) return PWICP_E_NOMEM;
    int rc = PWICP_OK;
    if (c->rank == root && hipMemcpyAsync(d, buf, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = PWICP_E_NO_DEVICE;
    if (rc == PWICP_OK && g_rccl.Broadcast(d, d, bytes, ncclChar, root, c->comm, c->stream) != ncclSuccess) rc = PWICP_E_INTERNAL;
    if (rc == PWICP_OK && hipMemcpyAsync(buf, d, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = PWICP_E_NO_DEVICE;
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == PWICP_OK) rc = PWICP_E_NO_DEVICE;
    return rc;
}

// One rank of a 4D series sharded over `world` processes (one GPU each): PiecewiseICP_4D_call (R.cpp:17-215) with its pair
// loop (R.cpp:89-187) dealt out as pair p -> rank p mod world.  In adaptive mode the overlap ratios behind the pair map are
// dealt out too and every rank replays calAdaptivePairSequence (R.cpp:552-589) on the gathered table; every rank runs its pairs; ONE all-gather of the 384-byte
// records; rank 0 writes the reference's result files.  Every collective is preceded by an agreement on the ranks' status
// (a 4-byte all-gather), so a rank that failed locally makes all ranks return false instead of leaving them in a collective.
PWICP_API bool pwicp_series_run_distributed(const char* confile, int startEpoch, int epochNum, int pairMode, float overlapThd,
                                            int rank, int world, int device, const char* id_file) {
    if (!confile || world < 1 || rank < 0 || rank >= world) return false;
    pwicp_comm* comm = nullptr;
    if (pwicp_comm_init(rank, world, device, id_file, &comm) != PWICP_OK) return false;
    auto agree = [&](bool mine) -> bool {
        int32_t v = mine ? 1 : 0;
        std::vector<int32_t> all((size_t)world, 0);
        if (pwicp_comm_allgather(comm, &v, sizeof(v), all.data()) != PWICP_OK) return false;
        for (int32_t x : all) if (!x) return false;
        return true;
    };
    pwicp_series* s = nullptr;
    bool ok = true, result = false;
    std::vector<int32_t> targets;
    int32_t n_t = 0;
    do {
        if (pairMode < 0 && world > 1) {
            // adaptive mode: the overlap ratios of the candidate pairs (source j against the targets j-W .. j-1, R.cpp:593-614) are
            // independent - dealt to the ranks, all-gathered as one #files x #files table - and every rank replays the
            // sequential target scan (R.cpp:552-589) on the same table: the same map everywhere, nothing to broadcast.  A
            // ratio the scan needs beyond the window is computed on the spot (by every rank alike).
            ok = pwicp_series_open(confile, startEpoch, epochNum, pairMode, overlapThd, device, nullptr, -1, &s) == PWICP_OK;
            if (!agree(ok)) break;
            const int nf = pwicp_series_num_scans(s);
            int W = 6;
            if (const char* e = getenv("PWICP_ADAPTIVE_WINDOW")) W = std::max(1, atoi(e));
            std::vector<float> mine_tab((size_t)nf * nf, NAN);
            {
                // dealt in CONTIGUOUS blocks of the (source, target) order: a rank reads the scans of its block of sources plus
                // the W before it, and its scan cache (bounded, registration.cpp: series_overlap) walks along with it
                std::vector<int32_t> all_ij, ij;
                for (int j = startEpoch + 1; j < nf; ++j)
                    for (int i = std::max(startEpoch, j - W); i < j; ++i) { all_ij.push_back(i); all_ij.push_back(j); }
                const long long nc = (long long)(all_ij.size() / 2);
                for (long long p = nc * rank / world; p < nc * (rank + 1) / world; ++p) { ij.push_back(all_ij[2 * p]); ij.push_back(all_ij[2 * p + 1]); }
                std::vector<float> r(ij.size() / 2 + 1);
                ok = pwicp_series_overlap_ratios(s, ij.data(), (int)(ij.size() / 2), r.data()) == PWICP_OK;
                for (size_t k = 0; ok && k < ij.size() / 2; ++k) mine_tab[(size_t)ij[2 * k] * nf + ij[2 * k + 1]] = r[k];
            }
            if (!agree(ok)) break;
            std::vector<float> all_tab((size_t)nf * nf * world);
            ok = pwicp_comm_allgather(comm, mine_tab.data(), sizeof(float) * mine_tab.size(), all_tab.data()) == PWICP_OK;
            if (!agree(ok)) break;
            for (int r = 0; r < world; ++r)
                for (size_t k = 0; k < mine_tab.size(); ++k) {
                    const float v = all_tab[(size_t)r * mine_tab.size() + k];
                    if (v == v) mine_tab[k] = v;
                }
            ok = pwicp_series_adaptive_from_ratios(s, mine_tab.data(), overlapThd, rank == 0 ? 1 : 0) == PWICP_OK;
            if (!agree(ok)) break;
        }
        if (!s) ok = pwicp_series_open(confile, startEpoch, epochNum, pairMode, overlapThd, device, targets.empty() ? nullptr : targets.data(),
                                       (int)n_t, &s) == PWICP_OK;
        if (!agree(ok)) break;
        const int n = pwicp_series_num_pairs(s);
        const int slots = (n + world - 1) / world;
        std::vector<int32_t> mine;
        for (int p = rank; p < n; p += world) mine.push_back(p);
        std::vector<pwicp_pair_record> loc((size_t)std::max(slots, 1));
        for (auto& r : loc) { std::memset(&r, 0, sizeof(r)); r.pair = -1; }
        // Direct2Ref: the pairs of every rank have the SAME target (the reference epoch).  Rank 0 - the owner of pair 0 - segments
        // it as part of its run; the others preprocess it themselves and take its labels (4 bytes per point) from a broadcast
        // that a helper thread runs beside the rank's own preparation (pwicp_series_*_target_labels).  Every rank takes part in
        // the two broadcasts whether or not it has pairs; a failure on rank 0 travels as m = -1 and the others segment for themselves.
        const bool share_target = pairMode == 0 && world > 1 && n > 0 && !(getenv("PWICP_SHARE_TARGET") && atoi(getenv("PWICP_SHARE_TARGET")) == 0);
        std::thread label_thread;
        if (share_target) {
            if (rank != 0) (void)pwicp_series_expect_target_labels(s, startEpoch);
            label_thread = std::thread([&, s] {
                int32_t hdr[2] = {-1, 0};
                std::vector<int32_t> lab;
                if (rank == 0) {
                    int m = 0, nsv = 0;
                    if (pwicp_series_wait_target_labels(s, startEpoch, 3600 * 1000, &m, &nsv, nullptr, 0) == PWICP_OK) {
                        lab.resize((size_t)std::max(m, 1));
                        if (pwicp_series_wait_target_labels(s, startEpoch, 0, &m, &nsv, lab.data(), m) == PWICP_OK) { hdr[0] = m; hdr[1] = nsv; }
                    }
                }
                bool got = pwicp_comm_broadcast(comm, hdr, sizeof(hdr), 0) == PWICP_OK;
                if (got && hdr[0] > 0) {
                    lab.resize((size_t)hdr[0]);
                    got = pwicp_comm_broadcast(comm, lab.data(), sizeof(int32_t) * (size_t)hdr[0], 0) == PWICP_OK;
                }
                if (rank != 0) (void)pwicp_series_supply_target_labels(s, startEpoch, (got && hdr[0] > 0) ? hdr[0] : -1, hdr[1], lab.data());
            });
        }
        if (!mine.empty()) ok = pwicp_series_run_pairs(s, mine.data(), (int)mine.size(), loc.data()) != PWICP_E_NO_DEVICE;
        if (share_target) {
            pwicp_series_close_target_labels(s);          // (rank 0: a run that never reached its target wakes the helper)
            label_thread.join();
        }
        if (!agree(ok)) break;
        std::vector<pwicp_pair_record> all((size_t)std::max(slots, 1) * (size_t)world);
        ok = slots == 0 || pwicp_comm_allgather(comm, loc.data(), sizeof(pwicp_pair_record) * (size_t)slots, all.data()) == PWICP_OK;
        if (!agree(ok)) break;
        if (rank == 0) {
            std::vector<pwicp_pair_record> recs;
            for (auto& r : all) if (r.pair >= 0) recs.push_back(r);
            ok = (int)recs.size() == n && pwicp_series_write_results(s, recs.data(), (int)recs.size()) == PWICP_OK;
            for (auto& r : recs) ok = ok && r.status == PWICP_OK;
        }
        result = agree(ok);
    } while (0);
    if (s) pwicp_series_close(s);
    pwicp_comm_destroy(comm);
    return result;
}

}  // extern "C"
