// The one exchange of a multi-process 4D series — the all-gather of the 384-byte pair records (and, in adaptive mode, the
// broadcast of the pair map) — over RCCL directly: one process per GPU, no Python, no torch.  librccl.so is opened with
// dlopen on first use, so single-GPU users never load it.
//
// Reference: the pair loop of PiecewiseICP_4D_call (src/Registration.cpp:89-187) is sequential in one process; its
// iterations are independent, which is what this shards: pair p -> rank p mod world, results gathered once.
// Rendezvous: the ncclUniqueId travels through a file (single node): rank 0 writes <id_file>.tmp and renames it, the other
// ranks poll for it.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <thread>
#include <vector>

#include "pwicp.h"

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

}  // namespace

struct pwicp_comm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    std::string id_file;
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
        if (g_rccl.GetUniqueId(&id) != ncclSuccess) return PWICP_E_NO_DEVICE;
        if (world > 1) {
            const std::string tmp = path + ".tmp";
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(&id, sizeof(id), 1, f) != 1) { if (f) std::fclose(f); return PWICP_E_INTERNAL; }
            std::fclose(f);
            if (std::rename(tmp.c_str(), path.c_str()) != 0) return PWICP_E_INTERNAL;
        }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            FILE* f = std::fopen(path.c_str(), "rb");
            if (f) {
                const size_t got = std::fread(&id, sizeof(id), 1, f);
                std::fclose(f);
                if (got == 1) break;
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
                std::cerr << "Error: rank " << rank << " never saw the RCCL id file " << path << "\n";
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
        delete c;
        return PWICP_E_NO_DEVICE;
    }
    if (hipStreamCreate(&c->stream) != hipSuccess) { g_rccl.CommDestroy(c->comm); delete c; return PWICP_E_NO_DEVICE; }
    *out = c;
    return PWICP_OK;
}

PWICP_API void pwicp_comm_destroy(pwicp_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->comm) g_rccl.CommDestroy(c->comm);
    if (c->rank == 0 && c->world > 1 && !c->id_file.empty()) std::remove(c->id_file.c_str());
    delete c;
}

PWICP_API int pwicp_comm_rank(const pwicp_comm* c) { return c ? c->rank : -1; }
PWICP_API int pwicp_comm_world(const pwicp_comm* c) { return c ? c->world : 0; }

// recv holds world * bytes; host buffers, staged through device memory (RCCL moves device memory over xGMI)
PWICP_API int pwicp_comm_allgather(pwicp_comm* c, const void* send, size_t bytes, void* recv) {
    if (!c || !send || !recv || bytes == 0) return PWICP_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return PWICP_E_NO_DEVICE;
    void *ds = nullptr, *dr = nullptr;
    int rc = PWICP_OK;
    if (hipMalloc(&ds, bytes) != hipSuccess || hipMalloc(&dr, bytes * (size_t)c->world) != hipSuccess) rc = PWICP_E_NOMEM;
    if (rc == PWICP_OK && hipMemcpyAsync(ds, send, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = PWICP_E_NO_DEVICE;
    if (rc == PWICP_OK && g_rccl.AllGather(ds, dr, bytes, ncclChar, c->comm, c->stream) != ncclSuccess) rc = PWICP_E_INTERNAL;
    if (rc == PWICP_OK && hipMemcpyAsync(recv, dr, bytes * (size_t)c->world, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = PWICP_E_NO_DEVICE;
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == PWICP_OK) rc = PWICP_E_NO_DEVICE;
    if (ds) (void)hipFree(ds);
    if (dr) (void)hipFree(dr);
    return rc;
}

PWICP_API int pwicp_comm_broadcast(pwicp_comm* c, void* buf, size_t bytes, int root) {
    if (!c || !buf || bytes == 0 || root < 0 || root >= c->world) return PWICP_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return PWICP_E_NO_DEVICE;
    void* d = nullptr;
    int rc = PWICP_OK;
    if (hipMalloc(&d, bytes) != hipSuccess) rc = PWICP_E_NOMEM;
    if (rc == PWICP_OK && c->rank == root && hipMemcpyAsync(d, buf, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = PWICP_E_NO_DEVICE;
    if (rc == PWICP_OK && g_rccl.Broadcast(d, d, bytes, ncclChar, root, c->comm, c->stream) != ncclSuccess) rc = PWICP_E_INTERNAL;
    if (rc == PWICP_OK && hipMemcpyAsync(buf, d, bytes, hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = PWICP_E_NO_DEVICE;
    if (hipStreamSynchronize(c->stream) != hipSuccess && rc == PWICP_OK) rc = PWICP_E_NO_DEVICE;
    if (d) (void)hipFree(d);
    return rc;
}

// One rank of a 4D series sharded over `world` processes (one GPU each): PiecewiseICP_4D_call (R.cpp:17-215) with its pair
// loop (R.cpp:89-187) dealt out as pair p -> rank p mod world.  In adaptive mode rank 0 determines the pair map
// (calAdaptivePairSequence, R.cpp:552-589) and broadcasts it; every rank runs its pairs; ONE all-gather of the 384-byte
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
            if (rank == 0) {
                ok = pwicp_series_open(confile, startEpoch, epochNum, pairMode, overlapThd, device, nullptr, 0, &s) == PWICP_OK;
                if (ok) {
                    n_t = pwicp_series_num_scans(s) - startEpoch - 1;
                    targets.resize((size_t)std::max(n_t, 1));
                    ok = n_t > 0 && pwicp_series_adaptive_targets(s, targets.data(), n_t) == PWICP_OK;
                }
            }
            if (!agree(ok)) break;
            if (pwicp_comm_broadcast(comm, &n_t, sizeof(n_t), 0) != PWICP_OK) { ok = false; }
            if (ok) targets.resize((size_t)std::max(n_t, 1));
            if (ok && pwicp_comm_broadcast(comm, targets.data(), sizeof(int32_t) * (size_t)n_t, 0) != PWICP_OK) ok = false;
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
        if (!mine.empty()) ok = pwicp_series_run_pairs(s, mine.data(), (int)mine.size(), loc.data()) != PWICP_E_NO_DEVICE;
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
