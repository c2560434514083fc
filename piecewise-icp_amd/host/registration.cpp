// The reference's two exported entry points, on top of the core C ABI (pwicp.h):
//   bool PiecewiseICP_pair_call(const char* confile, const char* outfile)            include/Registration.h:49
//   bool PiecewiseICP_4D_call(const char* confile, int startEpoch, int epochNum, int pairMode, float overlapThd)
//                                                                                     include/Registration.h:36
// Same names, argument meaning, side effects (result files) and `bool` return; never exit()s.
//
// Reference: PiecewiseICP_pair_call src/Registration.cpp:219-398, PiecewiseICP_4D_call Registration.cpp:17-215,
// Piecewise_ICP_4D Registration.cpp:402-548, calAdaptivePairSequence Registration.cpp:552-589,
// calTransToReferenceEpoch Registration.cpp:977-1153, calAbsErrorOfTransPara Registration.cpp:1157-1251.
// Pipeline per pair: load PCD -> VoxelGrid + SOR -> subtract the target centroid -> supervoxel labels (host front
// end) -> pwicp_pair_create / pwicp_pair_run (the fine-registration loop on the GPU) -> T_final = S^-1 T S -> files.
#include <hip/hip_runtime_api.h>      // (hipGetDevice / hipSetDevice around the release of parked contexts only)
#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <condition_variable>
#include <map>
#include <set>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "frontend.h"
#include "hostbuf.h"
#include "io.h"
#include "parallel.h"
#include "pwicp.h"


using namespace pwhost;

namespace {

constexpr int kNN = 45;      // include/CommonFunc.h:41

struct PairOutput {
    float T[16];
    float para[6];           // Rx,Ry,Rz [gon], tx,ty,tz [m]
    double VCM[36];
    pwicp_result res;
};

// One preprocessed, centroid-reduced cloud with its supervoxel labels.  Preparing it has a GPU part (VoxelGrid + SOR,
// k-NN graph) and a host part (PCA normals, supervoxel fusion, boundary refinement: serial per cloud, ~1.2 s per 1 M
// points) — kept apart so that the host parts of several clouds can run side by side on host threads.
struct Prepared {
    std::vector<float> p;          // preprocessed points, shifted
    int m = 0;
    float shift[3] = {0, 0, 0};    // the translation that was applied (minus the centroid of the pair's target)
    float Res = 0.f, SVRes = 0.f;
    double sor_mult = 0.0;
    HostBuf<int32_t> nb;           // k-NN graph, m x kNN (host front end only; released by prepare_labels)
    std::vector<int32_t> lab;
    int nsv = 0;
    bool segmented = false;        // labels already made by the device pipeline
    pwicp_target* dev = nullptr;   // device side of a TARGET (cloud, patches, normals, grids), built at its first pair
    Prepared() = default;
    Prepared(const Prepared&) = delete;
    Prepared& operator=(const Prepared&) = delete;
    ~Prepared() { if (dev) pwicp_target_destroy(dev); }
};

// $PWICP_FRONTEND=host keeps the serial host passes (same labels); default: the device pipeline
bool frontend_on_device() {
    const char* e = std::getenv("PWICP_FRONTEND");
    return !(e && std::string(e) == "host");
}

// $PWICP_TRACE_TIMELINE=1: wall-clock stamps (ms since the first one) of the series' events on stderr
void timeline(const char* what, int k = -1) {
    static const bool on = std::getenv("PWICP_TRACE_TIMELINE") != nullptr;
    if (!on) return;
    static const auto t0 = std::chrono::steady_clock::now();
    static std::mutex mu;
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::lock_guard<std::mutex> g(mu);
    std::fprintf(stderr, "[pwicp timeline] %9.2f ms  %s %d\n", ms, what, k);
}

// GPU part.  shift_in == nullptr: the cloud is a target and is reduced by its own centroid (R.cpp:419-436:
// pcl::compute3DCentroid float sums, float shift); otherwise the target's shift is applied.
// VoxelGrid + SOR of a raw scan do not depend on the role the scan plays in a pair (only the shift that follows does): in the
// adaptive and the fixed-interval mode every scan of a window is a source AND a target, and is preprocessed once (keyed by scan,
// leaf size and SOR multiplier; the reference's own run in those modes: the one-after-the-other GPU stage 230 -> 120 ms).
struct PreCache {
    struct Key {
        int scan; float res; double sor;
        bool operator<(const Key& o) const { return std::tie(scan, res, sor) < std::tie(o.scan, o.res, o.sor); }
    };
    struct Out { std::vector<float> p; int m = 0; };
    std::map<Key, std::shared_ptr<const Out>> done;
    std::map<int, float> resolution;         // calPCresolution of a raw scan (when the configuration does not give it)
};

bool prepare_gpu(pwicp_context* ctx, const std::vector<float>& raw, float Res, float SVRes, double sor_mult, const float* shift_in,
                 Prepared* c, PreCache* cache = nullptr, int scan = -1) {
    const int n = (int)(raw.size() / 4);
    c->Res = Res; c->SVRes = SVRes; c->sor_mult = sor_mult;
    std::shared_ptr<const PreCache::Out> hit;
    if (cache && scan >= 0) {
        auto it = cache->done.find(PreCache::Key{scan, Res, sor_mult});
        if (it != cache->done.end()) hit = it->second;
    }
    if (hit) {
        c->p = hit->p;
        c->m = hit->m;
    } else {
        c->p.resize((size_t)std::max(n, 1) * 4);
        if (pwicp_preprocess_dev(ctx, raw.data(), n, Res, 14, sor_mult, c->p.data(), &c->m) != PWICP_OK) {      // R.cpp:412-416
            std::cerr << "Error: preprocessing failed: " << pwicp_last_error(ctx) << "\n";
            return false;
        }
        c->p.resize((size_t)std::max(c->m, 1) * 4);
        if (cache && scan >= 0) {
            auto o = std::make_shared<PreCache::Out>();
            o->p = c->p; o->m = c->m;
            cache->done[PreCache::Key{scan, Res, sor_mult}] = o;
        }
    }
    const int m = c->m;
    if (m < kNN + 1) { std::cerr << "Error: too few points after preprocessing.\n"; return false; }
    if (shift_in) {
        for (int d = 0; d < 3; ++d) c->shift[d] = shift_in[d];
    } else {
        float acc[3] = {0, 0, 0};
        for (int i = 0; i < m; ++i) { acc[0] += c->p[4 * (size_t)i]; acc[1] += c->p[4 * (size_t)i + 1]; acc[2] += c->p[4 * (size_t)i + 2]; }
        for (int d = 0; d < 3; ++d) c->shift[d] = -1 * (acc[d] / (float)m);
    }
    const float* sh = c->shift;
    const float S[16] = {1, 0, 0, sh[0], 0, 1, 0, sh[1], 0, 0, 1, sh[2], 0, 0, 0, 1};
    for (int i = 0; i < m; ++i) {                                  // pcl::transformPointCloud with a pure translation
        float* q = c->p.data() + 4 * (size_t)i;
        const float x = q[0], y = q[1], z = q[2];
        q[0] = S[0] * x + S[1] * y + S[2] * z + S[3];
        q[1] = S[4] * x + S[5] * y + S[6] * z + S[7];
        q[2] = S[8] * x + S[9] * y + S[10] * z + S[11];
    }
    if (frontend_on_device()) return true;                         // S.cpp:30-68 follow in prepare_labels, on a stream of their own
    if (!c->nb.reserve((size_t)m * kNN)) { std::cerr << "Error: out of host memory.\n"; return false; }
    if (pwicp_knn(ctx, c->p.data(), m, kNN, 2.0f * Res, c->nb.data()) != PWICP_OK) {                          // S.cpp:30-41
        std::cerr << "Error: supervoxel segmentation failed: " << pwicp_last_error(ctx) << "\n";
        return false;
    }
    return true;
}

// Auxiliary contexts (= streams) of one device: the front ends of several clouds run side by side, each on its own stream
// and host thread (a front end is hundreds of small dependent launches: one alone leaves most of the GPU idle).
struct AuxContexts {
    int device = 0, limit = 3, made = 0;          // (limit: raised by size_for once the clouds' size is known)
    bool limit_from_env = false;
    std::mutex m;
    std::condition_variable cv;
    std::vector<pwicp_context*> idle;
    explicit AuxContexts(int dev) : device(dev) {
        if (const char* e = std::getenv("PWICP_FRONTEND_STREAMS")) { limit = std::max(1, std::min(atoi(e), 16)); limit_from_env = true; }
    }
    // Streams by the size of the clouds they will segment: the front end of a 1 M-point cloud fills the device for most of its
    // time and three or four of them side by side saturate it (3 / 4 / 5 streams: 0.387 / 0.377 / 0.379 s for 8 pairs); that of a 140 k-point
    // scan is a chain of small launches - the reference's 19 pairs: 0.28 - 0.32 s with three streams, 0.23 - 0.25 s with six
    // ($PWICP_FRONTEND_STREAMS decides when it is set).
    void size_for(long long points_per_cloud) {
        if (limit_from_env) return;
        std::lock_guard<std::mutex> lk(m);
        limit = std::max(limit, points_per_cloud < 400000 ? 6 : 4);
    }
    // a parked set taken over by the next series: the limit starts over (streams beyond it stay idle in the list - a stream that was
    // sized for small scans is not handed a 5 M-point cloud just because an earlier series had six of them)
    void reset_limit() {
        if (limit_from_env) return;
        std::lock_guard<std::mutex> lk(m);
        limit = 3;
    }
    AuxContexts(const AuxContexts&) = delete;
    AuxContexts& operator=(const AuxContexts&) = delete;
    ~AuxContexts() { for (pwicp_context* c : idle) pwicp_destroy(c); }
    pwicp_context* acquire() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            if (!idle.empty()) { pwicp_context* c = idle.back(); idle.pop_back(); return c; }
            if (made < limit) {
                pwicp_context* c = nullptr;
                if (pwicp_create(&c, device) == PWICP_OK) { ++made; return c; }
                if (made == 0) return nullptr;
            }
            cv.wait(lk);
        }
    }
    void release(pwicp_context* c) {
        { std::lock_guard<std::mutex> lk(m); idle.push_back(c); }
        cv.notify_one();
    }
};

// supervoxel labels of a prepared cloud (S.cpp:30-68; thread-safe): the device pipeline on a stream of its own, or
// ($PWICP_FRONTEND=host) the serial host passes from the k-NN graph prepare_gpu left
bool prepare_labels(Prepared* c, AuxContexts* aux) {
    if (c->segmented) return true;
    if (frontend_on_device()) {
        timeline("front end: waiting for a stream");
        pwicp_context* ctx = aux->acquire();
        timeline("front end: stream acquired");
        if (!ctx) { std::cerr << "Error: no usable HIP device (pwicp has no CPU fallback).\n"; return false; }
        c->lab.resize((size_t)c->m);
        // the device pipeline indexes its lists with 32 bits (n * k < 2^31) and wants ~1 KB of work space per point; a cloud
        // beyond either limit takes the serial passes of the same front end (host/frontend.cpp: same labels) from a device
        // k-NN graph instead of failing - what the fusion does anyway when a search outgrows its queue
        const bool fits = (long long)c->m * kNN <= (long long)INT_MAX - 64;
        int rc = fits ? pw_frontend_segment_device(ctx, c->p.data(), c->m, kNN, 2.0f * c->Res, c->SVRes, c->lab.data(), &c->nsv)
                      : PWICP_E_NOMEM;
        if (rc == PWICP_E_NOMEM) {
            std::cerr << "[pwicp] front end: " << (fits ? "device work space does not fit" : "cloud beyond the 32-bit list limit of the device pipeline")
                      << " (" << c->m << " points); serial fusion / refinement on the host for this cloud.\n";
            pw_frontend_release_workspace(ctx);
            rc = c->nb.reserve((size_t)c->m * kNN) ? PWICP_OK : PWICP_E_NOMEM;
            if (rc == PWICP_OK) rc = pwicp_knn(ctx, c->p.data(), c->m, kNN, 2.0f * c->Res, c->nb.data());
            if (rc == PWICP_OK) rc = segment_from_knn(c->p.data(), c->m, c->nb.data(), kNN, c->SVRes, c->lab.data(), &c->nsv);
            c->nb.release();
        }
        if (rc != PWICP_OK) std::cerr << "Error: supervoxel segmentation failed: " << pwicp_last_error(ctx) << "\n";
        aux->release(ctx);
        timeline("front end: done");
        c->segmented = rc == PWICP_OK;
        return c->segmented;
    }
    c->lab.resize((size_t)c->m);
    const int rc = segment_from_knn(c->p.data(), c->m, c->nb.data(), kNN, c->SVRes, c->lab.data(), &c->nsv);
    c->nb.release();
    return rc == PWICP_OK;
}

// the registration of two prepared clouds: Piecewise_ICP (R.cpp:618-700) on the GPU, then T_final = S^-1 T S (R.cpp:461)
bool run_prepared(pwicp_context* ctx, Prepared& t, const Prepared& s, const ConfigPara& cfg, PairOutput* out) {
    StageTimer tm;
    std::cout << "Preprocessed PC-1 point number: " << t.m << "\tPreprocessed PC-2 point number: " << s.m << std::endl << std::endl;
    std::cout << "--->>> " << t.nsv << " / " << s.nsv << " supervoxels are generated." << std::endl;
    pwicp_params prm{t.Res, s.Res, t.SVRes, s.SVRes, cfg.isSetDTinit ? 1 : 0, cfg.DTinit, cfg.DTmin};
    pwicp_pair* pair = nullptr;
    // the target's device side is built once and shared by all pairs with this target (R.cpp:653 rebuilds it per pair)
    if (!t.dev && pwicp_target_create(ctx, t.p.data(), t.m, t.lab.data(), t.nsv, t.Res, t.SVRes, &t.dev) != PWICP_OK) {
        std::cerr << "Error: " << pwicp_last_error(ctx) << "\n";
        return false;
    }
    if (pwicp_pair_create_with_target(t.dev, s.p.data(), s.m, s.lab.data(), s.nsv, &prm, &pair) != PWICP_OK) {
        std::cerr << "Error: " << pwicp_last_error(ctx) << "\n";
        return false;
    }
    tm.lap("upload, patches, grids (GPU)");
    int M1 = 0, M2 = 0;
    pwicp_pair_num_patches(pair, &M1, &M2);
    std::cout << "PC-1 selected patch number: " << M1 << "\tPC-2 selected patch number: " << M2 << std::endl;
    const int rc = pwicp_pair_run(pair, &out->res);
    tm.lap("registration loop (GPU)");
    pwicp_pair_destroy(pair);
    if (rc != PWICP_OK) {
        std::cerr << "Error: registration failed (status " << rc << "): " << pwicp_last_error(ctx) << "\n";
        return false;
    }
    for (int k = 0; k < out->res.n_outer; ++k)
        std::cout << "--->>> Iteration No." << k + 1 << " | Current DT = " << out->res.DTseries[k + 1] * 100 << " cm. \n";
    const float* sh = t.shift;
    const float S[16] = {1, 0, 0, sh[0], 0, 1, 0, sh[1], 0, 0, 1, sh[2], 0, 0, 0, 1};
    const float Sinv[16] = {1, 0, 0, -1 * sh[0], 0, 1, 0, -1 * sh[1], 0, 0, 1, -1 * sh[2], 0, 0, 0, 1};
    float tmpM[16];
    mat4_mul(Sinv, out->res.T16, tmpM);
    mat4_mul(tmpM, S, out->T);
    float ang[3];
    matrix2angle(out->T, ang);                                                     // R.cpp:464-480
    out->para[0] = (float)(ang[0] * ARC_TO_GON); out->para[1] = (float)(ang[1] * ARC_TO_GON); out->para[2] = (float)(ang[2] * ARC_TO_GON);
    out->para[3] = out->T[3]; out->para[4] = out->T[7]; out->para[5] = out->T[11];
    std::memcpy(out->VCM, out->res.VCM, sizeof(out->VCM));
    return true;
}

// Piecewise_ICP_4D without its file output (R.cpp:402-480) for ONE pair; sor_mult 5.0 (4D) or 2.7 (pair)
bool register_pair(pwicp_context* ctx, const std::vector<float>& cloud1, const std::vector<float>& cloud2,
                   const ConfigPara& cfg, float Res1, float Res2, double sor_mult, PairOutput* out) {
    StageTimer tm;
    std::cout << "Original PC-1 point number: " << cloud1.size() / 4 << "\t Original PC-2 point number: " << cloud2.size() / 4 << std::endl;
    std::cout << "PC-1 avg. point spacing: " << Res1 << "\t PC-2 avg. point spacing: " << Res2 << std::endl << std::endl;
    const float SVRes1 = cfg.isSetResSVsize ? cfg.SVsize1 : Res1 * 10, SVRes2 = cfg.isSetResSVsize ? cfg.SVsize2 : Res2 * 10;   // R.cpp:635-640
    Prepared t, s;
    // the streams (and GB-sized work spaces) of the front ends are kept with the context, not rebuilt per call
    std::shared_ptr<void>* slot = pw_context_host_slot(ctx);
    if (!slot) return false;
    if (!*slot) *slot = std::shared_ptr<void>(new AuxContexts(pwicp_context_device(ctx)), [](void* p) { delete static_cast<AuxContexts*>(p); });
    AuxContexts& aux = *static_cast<AuxContexts*>(slot->get());
    if (!prepare_gpu(ctx, cloud1, Res1, SVRes1, sor_mult, nullptr, &t)) return false;
    bool ok1 = true;
    std::thread th([&] { ok1 = prepare_labels(&t, &aux); });    // the target's front end runs beside the source's preparation and front end
    const bool ok2 = prepare_gpu(ctx, cloud2, Res2, SVRes2, sor_mult, t.shift, &s) && prepare_labels(&s, &aux);
    th.join();
    tm.lap("preparation (GPU: voxel grid, SOR, k-NN; host: supervoxels)");
    if (!ok1 || !ok2) { std::cerr << "Error: supervoxel segmentation failed.\n"; return false; }
    return run_prepared(ctx, t, s, cfg, out);
}

// calTransToReferenceEpoch, R.cpp:977-1153 (re-reads the pairwise file, exactly as the reference does)
bool trans_to_reference(const std::string& tm_file, int pairMode, const std::map<int, int>& reg_pair, int n,
                        const std::string& out_tm, const std::string& out_tp, int32_t* o_stamp = nullptr, float* o_T = nullptr,
                        double* o_V = nullptr) {
    std::vector<int> stamps;
    std::vector<std::array<float, 16>> Ts;
    std::vector<std::array<double, 36>> Vs;
    if (!read_transmatrices(tm_file, n, &stamps, &Ts, &Vs)) { std::cerr << "Error: Cannot open transMatFile!\n"; return false; }
    std::ofstream oTM(out_tm.c_str()), oTP(out_tp.c_str());
    if (!oTM || !oTP) { std::cerr << "Error: Cannot open transMat2RefFile!\n"; return false; }
    oTP << trans_parameters_header() << std::endl;
    for (int i = 0; i < n; ++i) {
        std::array<float, 16> accT;
        std::array<double, 36> accV;
        if (pairMode < 0) {
            accT = Ts[(size_t)i];
            accV = Vs[(size_t)i];
            int idx = i + 1;
            for (int j = 0; j < i + 1; ++j) {
                auto it = reg_pair.find(idx);
                idx = it == reg_pair.end() ? 0 : it->second;
                if (idx == 0) break;
                const std::array<float, 16>& M = Ts[(size_t)idx - 1];
                float nt[16];
                mat4_mul(M.data(), accT.data(), nt);
                std::memcpy(accT.data(), nt, sizeof(nt));
                // adjoint [[R,0],[t x R, R]] in double (R.cpp:1074-1083)
                double R[9], t[3];
                for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = (double)M[(size_t)(4 * r + c)]; t[r] = (double)M[(size_t)(4 * r + 3)]; }
                const double SS[9] = {0, -t[2], t[1], t[2], 0, -t[0], -t[1], t[0], 0};
                double txR[9];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) { double s = 0; for (int k = 0; k < 3; ++k) s += SS[3 * r + k] * R[3 * k + c]; txR[3 * r + c] = s; }
                double Adj[36] = {0};
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) { Adj[6 * r + c] = R[3 * r + c]; Adj[6 * (r + 3) + c + 3] = R[3 * r + c]; Adj[6 * (r + 3) + c] = txR[3 * r + c]; }
                double AV[36], AVA[36];
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c) { double s = 0; for (int k = 0; k < 6; ++k) s += Adj[6 * r + k] * accV[(size_t)(6 * k + c)]; AV[6 * r + c] = s; }
                for (int r = 0; r < 6; ++r)
                    for (int c = 0; c < 6; ++c) { double s = 0; for (int k = 0; k < 6; ++k) s += AV[6 * r + k] * Adj[6 * c + k]; AVA[6 * r + c] = s; }
                for (int k = 0; k < 36; ++k) accV[(size_t)k] = Vs[(size_t)idx - 1][(size_t)k] + AVA[k];
            }
        } else if (pairMode == 0 || i < pairMode) {
            accT = Ts[(size_t)i];
            accV = Vs[(size_t)i];
        } else {
            accT = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
            accV.fill(0.0);
            for (int j = 0; j < n; ++j) {
                const int src = i - pairMode * j;
                float nt[16];
                mat4_mul(Ts[(size_t)src].data(), accT.data(), nt);
                std::memcpy(accT.data(), nt, sizeof(nt));
                for (int k = 0; k < 36; ++k) accV[(size_t)k] = Vs[(size_t)src][(size_t)k] + accV[(size_t)k];
                if (src < pairMode) break;
            }
        }
        // R.cpp:1112-1149 (this file uses std::endl per row)
        oTM << std::fixed << std::setprecision(12);
        oTM << stamps[(size_t)i] << "\n";
        for (int r = 0; r < 4; ++r) { for (int c = 0; c < 4; ++c) oTM << accT[(size_t)(4 * r + c)] << " "; oTM << std::endl; }
        for (int r = 0; r < 6; ++r) { for (int c = 0; c < 6; ++c) oTM << accV[(size_t)(6 * r + c)] << " "; oTM << std::endl; }
        float ang[3];
        matrix2angle(accT.data(), ang);
        const float para[6] = {(float)(ang[0] * ARC_TO_GON), (float)(ang[1] * ARC_TO_GON), (float)(ang[2] * ARC_TO_GON),
                               accT[3], accT[7], accT[11]};
        append_transparameters(oTP, stamps[(size_t)i], para, accV.data());
        if (o_stamp) o_stamp[i] = stamps[(size_t)i];
        if (o_T) std::memcpy(o_T + 16 * (size_t)i, accT.data(), 16 * sizeof(float));
        if (o_V) std::memcpy(o_V + 36 * (size_t)i, accV.data(), 36 * sizeof(double));
    }
    return true;
}

// calAbsErrorOfTransPara, R.cpp:1157-1251; optional here (the reference hard-codes the GT path and exits if missing)
bool abs_error_report(const std::string& toref_file, const std::string& gt_file, int all_epochs, int start, const std::string& out_file) {
    const int n = all_epochs - start - 1;
    std::vector<int> stamps;
    std::vector<std::array<float, 16>> Ts;
    std::vector<std::array<double, 36>> Vs;
    if (!read_transmatrices(toref_file, n, &stamps, &Ts, &Vs)) return false;
    std::ifstream gt(gt_file);
    if (!gt) return false;
    std::vector<std::array<float, 16>> G;
    for (int i = 0; i < all_epochs; ++i) {
        int stamp;
        std::array<float, 16> T;
        if (!(gt >> stamp)) return false;
        for (int k = 0; k < 16; ++k) if (!(gt >> T[(size_t)k])) return false;
        G.push_back(T);
    }
    std::ofstream o(out_file);
    if (!o) return false;
    o << "Err_Rx[mgon]  Err_Ry[mgon]  Err_Rz[mgon]  Err_tx[mm]  Err_ty[mm]  Err_tz[mm]" << std::endl;
    for (int i = 0; i < n; ++i) {
        float a[3], b[3];
        matrix2angle(Ts[(size_t)i].data(), a);
        matrix2angle(G[(size_t)(start + 1 + i)].data(), b);
        const float e[6] = {1000 * std::fabs((float)(b[0] * ARC_TO_GON) - (float)(a[0] * ARC_TO_GON)),
                            1000 * std::fabs((float)(b[1] * ARC_TO_GON) - (float)(a[1] * ARC_TO_GON)),
                            1000 * std::fabs((float)(b[2] * ARC_TO_GON) - (float)(a[2] * ARC_TO_GON)),
                            1000 * std::fabs(G[(size_t)(start + 1 + i)][3] - Ts[(size_t)i][3]),
                            1000 * std::fabs(G[(size_t)(start + 1 + i)][7] - Ts[(size_t)i][7]),
                            1000 * std::fabs(G[(size_t)(start + 1 + i)][11] - Ts[(size_t)i][11])};
        o << e[0] << " " << e[1] << " " << e[2] << " " << e[3] << " " << e[4] << " " << e[5] << " " << std::endl;
    }
    return true;
}

int env_device() {
    const char* e = getenv("PWICP_DEVICE");
    if (!e) e = getenv("LOCAL_RANK");
    return e ? atoi(e) : 0;
}

// Devices of an entry-point call: $PWICP_DEVICES = "all" or a comma-separated list (duplicates allowed: two workers on one
// GPU); else $PWICP_DEVICE / $LOCAL_RANK (one rank of a multi-process launch owns one GPU); else every visible device.
std::vector<int32_t> env_devices() {
    std::vector<int32_t> d;
    const char* e = getenv("PWICP_DEVICES");
    if (e && std::strcmp(e, "all") != 0) {
        std::stringstream ss(e);
        std::string tok;
        while (std::getline(ss, tok, ',')) if (!tok.empty()) d.push_back(atoi(tok.c_str()));
    } else if (!e && (getenv("PWICP_DEVICE") || getenv("LOCAL_RANK"))) {
        d.push_back(env_device());
    } else {
        const int n = pwicp_device_count();
        for (int i = 0; i < std::max(n, 1); ++i) d.push_back(i);
    }
    if (d.empty()) d.push_back(0);
    return d;
}

}  // namespace

// the context of the pair entry point (with the front-end streams and work spaces that hang on it) is parked between calls
// like a series worker's (WorkerParking, below): a process that calls PiecewiseICP_pair_call in a loop sets them up once
pwicp_context* pair_context_take(int device);
void pair_context_park(pwicp_context* ctx);

extern "C" {

PWICP_API bool PiecewiseICP_pair_call(const char* confile, const char* outfile) {
    if (!confile || !outfile) return false;
    ConfigPara cfg;
    std::cout << "Loading parameter configuration file: " << confile << "\n\n";
    if (!read_config(confile, &cfg)) { std::cerr << "Error: Cannot open configuration file! Aborting.\n\n"; return false; }
    std::vector<float> c1, c2;
    if (!load_pcd(cfg.FolderFilePath1, &c1) || !load_pcd(cfg.FolderFilePath2, &c2) || c1.empty() || c2.empty()) return false;   // R.cpp:252-256
    pwicp_context* ctx = pair_context_take(env_device());
    if (!ctx) { std::cerr << "Error: no usable HIP device (pwicp has no CPU fallback).\n"; return false; }
    float Res1 = cfg.PCres1, Res2 = cfg.PCres2;
    if (!cfg.isSetResSVsize &&
        (pwicp_pc_resolution_dev(ctx, c1.data(), (int)(c1.size() / 4), &Res1) != PWICP_OK ||
         pwicp_pc_resolution_dev(ctx, c2.data(), (int)(c2.size() / 4), &Res2) != PWICP_OK)) {
        std::cerr << "Error: " << pwicp_last_error(ctx) << "\n";
        pwicp_destroy(ctx);
        return false;
    }
    PairOutput out;
    const bool ok = register_pair(ctx, c1, c2, cfg, Res1, Res2, 2.7, &out);          // SOR multiplier 2.7 (R.cpp:272-273)
    if (ok) pair_context_park(ctx); else pwicp_destroy(ctx);
    if (!ok) return false;
    if (!write_transmatrix_file(std::string(outfile) + "TransMatrix.txt", out.T, out.VCM)) return false;
    std::cout << "--->>> Transformation results saved.\n";
    // registered source cloud (R.cpp:331-333, 391-394): the ORIGINAL source transformed by T_final
    const int n2 = (int)(c2.size() / 4);
    std::vector<float> moved(c2);
    for (int i = 0; i < n2; ++i) {
        float* q = moved.data() + 4 * (size_t)i;
        const float x = q[0], y = q[1], z = q[2];
        q[0] = out.T[0] * x + out.T[1] * y + out.T[2] * z + out.T[3];
        q[1] = out.T[4] * x + out.T[5] * y + out.T[6] * z + out.T[7];
        q[2] = out.T[8] * x + out.T[9] * y + out.T[10] * z + out.T[11];
    }
    if (!save_pcd_binary(std::string(outfile) + "RegisteredSourceCloud.pcd", moved.data(), n2)) return false;
    std::cout << "--->>> Registered source cloud saved.\n\n";
    return true;
}

// ---- the remaining functions of include/Registration.h as C entry points (the PCL-typed wrappers with the reference's exact
// signatures live in include/pwicp/Registration.h) ----------------------------------------------------------------------------

// Piecewise_ICP_4D (R.cpp:402-548; decl R.h:74-78): PCpreprocessing (SOR multiplier 5.0) of both clouds, reduction by the
// target centroid, Piecewise_ICP, T_final = S^-1 T S, transformation parameters [gon, m], and — when outfileIdx is not
// NULL — "<outfileIdx>TransMatrix.txt".  isSetResSVsize == 0: patch size 10 x the given point spacing (R.cpp:635-640).
PWICP_API int pwicp_piecewise_icp_4d(pwicp_context* ctx, const float* cloud1_xyz4, int n1, const float* cloud2_xyz4, int n2,
                                     int isSetResSVsize, float Res1, float Res2, float SVsize1, float SVsize2, int isManualDTinit,
                                     float DTinit, float DTmin, const char* outfileIdx, float* transMat16, float* transPara6,
                                     double* VCM36) {
    if (!ctx || !cloud1_xyz4 || !cloud2_xyz4 || n1 <= 0 || n2 <= 0 || !transMat16 || !transPara6 || !VCM36) return PWICP_E_INVALID;
    ConfigPara cfg;
    cfg.isSetResSVsize = isSetResSVsize != 0; cfg.PCres1 = Res1; cfg.PCres2 = Res2; cfg.SVsize1 = SVsize1; cfg.SVsize2 = SVsize2;
    cfg.isSetDTinit = isManualDTinit != 0; cfg.DTinit = DTinit; cfg.DTmin = DTmin;
    std::vector<float> c1(cloud1_xyz4, cloud1_xyz4 + 4 * (size_t)n1), c2(cloud2_xyz4, cloud2_xyz4 + 4 * (size_t)n2);
    PairOutput out;
    std::memset(&out, 0, sizeof(out));
    if (!register_pair(ctx, c1, c2, cfg, Res1, Res2, 5.0, &out)) return out.res.status != 0 ? out.res.status : PWICP_E_INTERNAL;
    std::memcpy(transMat16, out.T, sizeof(out.T));
    std::memcpy(transPara6, out.para, sizeof(out.para));
    std::memcpy(VCM36, out.VCM, sizeof(out.VCM));
    if (outfileIdx && !write_transmatrix_file(std::string(outfileIdx) + "TransMatrix.txt", out.T, out.VCM)) {
        std::cerr << "Cannot open TransMatrix.txt for writing!\n\n";
        return PWICP_E_INTERNAL;
    }
    return PWICP_OK;
}

// calAdaptivePairSequence (R.cpp:552-589; decl R.h:93-94): targets[k] = target of source k+1, both relative to startEpoch
// (n_files - startEpoch - 1 entries); the pair file (may be NULL) gets the reference's "source target" lines.
PWICP_API int pwicp_adaptive_pair_sequence(pwicp_context* ctx, const char* const* fileNameList, int n_files, int startEpoch, float DTinit,
                                           float ratioThd, int32_t* targets, const char* adaptivePairFile) {
    if (!ctx || !fileNameList || !targets || n_files < 2 || startEpoch < 0 || startEpoch >= n_files - 1) return PWICP_E_INVALID;
    int idxTarget = startEpoch;
    std::vector<std::vector<float>> cache((size_t)n_files);
    auto cloud = [&](int i) -> std::vector<float>& { if (cache[(size_t)i].empty()) load_pcd(fileNameList[i], &cache[(size_t)i]); return cache[(size_t)i]; };
    for (int j = startEpoch + 1; j < n_files; ++j) {
        float ratio = 0;
        for (int i = idxTarget; i < j; ++i) {
            std::vector<float>&a = cloud(i), &b = cloud(j);
            if (a.empty() || b.empty()) return PWICP_E_INVALID;
            const int rc = pwicp_overlap_ratio(ctx, a.data(), (int)(a.size() / 4), b.data(), (int)(b.size() / 4), DTinit, &ratio);
            if (rc != PWICP_OK) return rc;
            idxTarget = i;
            if (ratio > ratioThd) break;
        }
        targets[j - startEpoch - 1] = idxTarget - startEpoch;
        std::cout << "Pair: " << idxTarget - startEpoch << " - " << j - startEpoch << ";  Overlap ratio = " << 100 * ratio << "% \n";
        if (idxTarget > startEpoch) cache[(size_t)idxTarget - 1].clear();
    }
    if (adaptivePairFile) {
        std::ofstream pf(adaptivePairFile);
        if (!pf) { std::cerr << "Error: Cannot open adaptivePairFile!\n"; return PWICP_E_INTERNAL; }
        for (int k = 0; k < n_files - startEpoch - 1; ++k) pf << k + 1 << " " << targets[k] << std::endl;
    }
    return PWICP_OK;
}

// calTransToReferenceEpoch (R.cpp:977-1153; decl R.h:127-129): reads the pairwise file (and, in adaptive mode, the pair file),
// writes the two "toRef" files and returns the composed matrices.  Optional outputs hold epochNum entries.
PWICP_API int pwicp_trans_to_reference_epoch(const char* transMatFile, int pairMode, const char* adaptivePairFile, int epochNum,
                                             const char* transMat2RefFile, const char* transPara2RefFile, int32_t* timeStamp,
                                             float* allTransMat2Ref16, double* allVCM2Ref36) {
    if (!transMatFile || !transMat2RefFile || !transPara2RefFile || epochNum <= 0) return PWICP_E_INVALID;
    std::map<int, int> reg;
    if (pairMode < 0) {
        if (!adaptivePairFile) return PWICP_E_INVALID;
        std::ifstream in(adaptivePairFile);
        if (!in) { std::cerr << "Error: Cannot open adaptivePairFile!\n"; return PWICP_E_INVALID; }
        for (int i = 0; i < epochNum; ++i) {                      // R.cpp:1024-1028
            int a, b;
            if (!(in >> a >> b)) break;
            reg.insert(std::make_pair(a, b));
        }
    }
    return trans_to_reference(transMatFile, pairMode, reg, epochNum, transMat2RefFile, transPara2RefFile, timeStamp, allTransMat2Ref16,
                              allVCM2Ref36) ? PWICP_OK : PWICP_E_INTERNAL;
}

// calAbsErrorOfTransPara (R.cpp:1157-1251; decl R.h:198-199)
PWICP_API int pwicp_abs_error_of_trans_para(const char* transMatFile, const char* GTtransMatFile, int allEpochNum, int startEpoch,
                                            const char* transParaErrorFile) {
    if (!transMatFile || !GTtransMatFile || !transParaErrorFile) return PWICP_E_INVALID;
    return abs_error_report(transMatFile, GTtransMatFile, allEpochNum, startEpoch, transParaErrorFile) ? PWICP_OK : PWICP_E_INVALID;
}

// matrix2angle (C.cpp:385-407; decl C.h:172): rotation angles [rad] about x, y, z of a row-major 4x4
PWICP_API void pwicp_matrix2angle(const float* T16, float* rotAngle3) { matrix2angle(T16, rotAngle3); }
PWICP_API int pwicp_write_trans_matrix_file(const char* path, const float* T16, const double* VCM36) {
    if (!path || !T16 || !VCM36) return PWICP_E_INVALID;
    return write_transmatrix_file(path, T16, VCM36) ? PWICP_OK : PWICP_E_INVALID;
}

// ---- 4D series as a handle: the pairs of R.cpp:89-187 are independent, so any subset can run on any GPU ---------
}  // extern "C"

// One GPU of a series: its context and the target clouds prepared on it.  A series owns one worker per device it was
// opened on; the pairs handed to pwicp_series_run_pairs are dealt to the workers (pair k of the call -> worker k mod G),
// each worker runs on a host thread of its own (the pairs are independent, R.cpp:89-187).
// What a series worker needs of a device - its context and the auxiliary contexts of the front ends with their work spaces
// (~2 GB each at 1 M points, pinned staging buffers, loaded kernels) - costs 0.15 - 0.25 s to set up from nothing: more than a
// third of an 8 x 1 M-point series.  A worker that closes PARKS them per device and the next series of the process on that device
// takes them over warm: the second PiecewiseICP_4D_call of a process costs what its clouds cost.  One set per device is kept
// ($PWICP_SERIES_KEEP=0: none) until pwicp_series_release_parked() or the end of the process.
struct ParkedWorker { pwicp_context* ctx = nullptr; std::unique_ptr<AuxContexts> aux; };
bool pw_set_release_parked_hook(bool (*fn)(int device));        // csrc/api.hip (library-internal)

struct WorkerParking {
    std::mutex mu;
    std::map<int, ParkedWorker> by_device;
    std::map<int, pwicp_context*> pair_by_device;      // the context of PiecewiseICP_pair_call (its front-end streams hang on it)
    bool exit_hook = false;
    static WorkerParking& get() { static WorkerParking* p = new WorkerParking; return *p; }       // (the object itself is never deleted)
    // destroys what is parked (pwicp_series_release_parked, and once at exit: registered when the first set is parked, i.e. after
    // the HIP runtime has registered its own exit work, so it runs before the runtime goes away)
    // `device` >= 0: only what is parked for that device (the out-of-memory hook of PwPool::take - the other devices' sets are
    // not what the failing allocation competes with).  pwicp_destroy selects the context's device: the caller's is put back.
    bool release_all(int device = -1) {
        std::map<int, ParkedWorker> take;
        std::map<int, pwicp_context*> take_pair;
        {
            std::lock_guard<std::mutex> g(mu);
            if (device < 0) { take.swap(by_device); take_pair.swap(pair_by_device); }
            else {
                auto a = by_device.find(device);
                if (a != by_device.end()) { take[device] = std::move(a->second); by_device.erase(a); }
                auto b = pair_by_device.find(device);
                if (b != pair_by_device.end()) { take_pair[device] = b->second; pair_by_device.erase(b); }
            }
        }
        if (take.empty() && take_pair.empty()) return false;
        int cur = -1;
        const bool have_cur = hipGetDevice(&cur) == hipSuccess;
        for (auto& kv : take) { kv.second.aux.reset(); if (kv.second.ctx) pwicp_destroy(kv.second.ctx); }
        for (auto& kv : take_pair) if (kv.second) pwicp_destroy(kv.second);
        if (have_cur) (void)hipSetDevice(cur);
        return true;
    }
    void hook_exit() {          // (call with mu held)
        if (!exit_hook) {
            exit_hook = true;
            std::atexit([] { (void)WorkerParking::get().release_all(); });
            // an allocation anywhere in the process that runs out of device memory gets the parked sets back before it fails
            // (csrc/common.h PwPool::take: after its own cache and every other cache of the device)
            pw_set_release_parked_hook([](int device) { return WorkerParking::get().release_all(device); });
        }
    }
    static bool enabled() { static const bool on = !(std::getenv("PWICP_SERIES_KEEP") && atoi(std::getenv("PWICP_SERIES_KEEP")) == 0); return on; }
};

pwicp_context* pair_context_take(int device) {
    if (WorkerParking::enabled()) {
        WorkerParking& pk = WorkerParking::get();
        std::lock_guard<std::mutex> g(pk.mu);
        auto it = pk.pair_by_device.find(device);
        if (it != pk.pair_by_device.end()) { pwicp_context* c = it->second; pk.pair_by_device.erase(it); return c; }
    }
    pwicp_context* c = nullptr;
    return pwicp_create(&c, device) == PWICP_OK ? c : nullptr;
}
void pair_context_park(pwicp_context* ctx) {
    if (WorkerParking::enabled()) {
        WorkerParking& pk = WorkerParking::get();
        std::lock_guard<std::mutex> g(pk.mu);
        const int device = pwicp_context_device(ctx);
        if (!pk.pair_by_device.count(device)) { pk.pair_by_device[device] = ctx; pk.hook_exit(); return; }
    }
    pwicp_destroy(ctx);
}

struct SeriesWorker {
    int device = 0;
    pwicp_context* ctx = nullptr;         // created by the first call that needs the GPU
    // prepared target clouds by epoch index (in the Direct2Ref mode every pair has the same target, R.cpp:94-103; in the
    // adaptive mode runs of pairs share one): prepared once PER DEVICE, kept while the following pairs use them
    std::map<int, std::shared_ptr<Prepared>> targets;
    std::unique_ptr<AuxContexts> aux;     // streams for the front ends of the clouds of a window
    bool need_ctx() {
        if (!ctx && !aux && WorkerParking::enabled()) {           // a warm set parked by an earlier series of this process?
            WorkerParking& pk = WorkerParking::get();
            std::lock_guard<std::mutex> g(pk.mu);
            auto it = pk.by_device.find(device);
            if (it != pk.by_device.end()) {
                ctx = it->second.ctx; aux = std::move(it->second.aux); pk.by_device.erase(it);
                if (aux) aux->reset_limit();                     // (sized again for THIS series' clouds, size_for)
            }
        }
        if (!aux) aux.reset(new AuxContexts(device));
        if (ctx) return true;
        if (pwicp_create(&ctx, device) != PWICP_OK) { std::cerr << "Error: no usable HIP device (pwicp has no CPU fallback).\n"; ctx = nullptr; return false; }
        return true;
    }
    void close() {
        targets.clear();                  // device-side targets go before their context
        if (ctx && aux && WorkerParking::enabled()) {
            WorkerParking& pk = WorkerParking::get();
            std::lock_guard<std::mutex> g(pk.mu);
            if (!pk.by_device.count(device)) {
                ParkedWorker& slot = pk.by_device[device];
                slot.ctx = ctx; slot.aux = std::move(aux);
                ctx = nullptr;
                pk.hook_exit();
            }
        }
        aux.reset();
        if (ctx) pwicp_destroy(ctx);
        ctx = nullptr;
    }
};

struct pwicp_series {
    ConfigPara cfg;
    std::string outputFolder;
    std::vector<std::string> files;
    std::vector<long> times;
    int startEpoch = 0, epochNum = 0, pairMode = 0, device = 0;
    std::map<int, int> regPairs;          // adaptive mode: source -> target, relative to startEpoch (R.cpp:570)
    bool adaptive_deferred = false;       // adaptive mode opened without its pair map (a sharded computation supplies it)
    std::mutex stage_mu;
    double stage_ms[4] = {0, 0, 0, 0};    // wall time spent so far: reading scans | GPU preparation | front ends (rest) | registrations
    long long stage_bytes = 0;            // bytes of raw scans handed to the GPU
    std::vector<std::vector<float>> scan_cache;   // raw scans read for the overlap ratios: at most scan_cache_cap() at a time,
    std::vector<int> scan_lru;                    // least recently used first; all dropped once the map is known
    std::vector<std::unique_ptr<SeriesWorker>> workers;      // [0] = `device`; more after pwicp_series_set_devices

    // The supervoxel labels of a target that several PROCESSES share (Direct2Ref: every pair has the reference epoch as its target;
    // the reference rebuilds it per pair, R.cpp:653): one rank segments it, the others preprocess it themselves (17 ms at 1 M points,
    // they need its centroid before their source) and take the labels - 4 bytes per point - from that rank instead of running the
    // 60 ms front end once more per rank.  pwicp_series_expect / supply / wait_target_labels, include/pwicp.h; the labelling is a
    // pure function of the preprocessed cloud, so the records are the ones every rank would have computed alone.
    struct LabelExchange {
        std::mutex mu;
        std::condition_variable cv;
        std::set<int> expected;                                        // scans whose labels another rank supplies
        struct Supplied { int m = -1, nsv = 0; std::vector<int32_t> lab; };
        std::map<int, Supplied> supplied;                              // ... as they arrived (m < 0: the supplier failed)
        std::map<int, std::shared_ptr<Prepared>> done;                 // targets segmented HERE (nullptr: failed), for wait_target_labels
        bool closed = false;                                           // the run is over: nothing more will become ready
        int n_received = 0, n_segmented = 0;                           // targets of this series whose labels arrived / were made here
    } lx;

    SeriesWorker* w0() {
        if (workers.empty()) { workers.emplace_back(new SeriesWorker); workers[0]->device = device; }
        return workers[0].get();
    }
    int num_pairs() const { return epochNum - startEpoch - 1; }
    int ref_index(int pair) const {                   // R.cpp:94-103
        const int i = startEpoch + pair, step = pair + 1;
        if (pairMode > 0) return (pairMode >= step) ? startEpoch : (i + 1 - pairMode);
        if (pairMode < 0) { auto it = regPairs.find(i + 1); return it == regPairs.end() ? -1 : it->second; }
        return startEpoch;
    }
};

namespace {

// calOverlapRatioByC2Cdist (R.cpp:593-614) of the raw scans i (target) and j (source), on the GPU; scans cached in the series
bool series_overlap(pwicp_series* s, int i, int j, float* ratio) {
    if (!s->w0()->need_ctx()) return false;
    const int fileCount = (int)s->files.size();
    if (i < 0 || j < 0 || i >= fileCount || j >= fileCount) return false;
    if ((int)s->scan_cache.size() != fileCount) { s->scan_cache.assign((size_t)fileCount, {}); s->scan_lru.clear(); }
    // The candidates of a source j are the W targets before it (R.cpp:593-614) and callers walk the sources in order (a rank of a
    // sharded run walks ITS contiguous block of them, host/comm.cpp): W + 2 scans cover the working set, whatever the series'
    // length - a few hundred 5 M-point epochs must not end up resident in every rank ($PWICP_SCAN_CACHE: scans kept, >= 2).
    static const int cap = [] { const char* e = getenv("PWICP_SCAN_CACHE"); return std::max(2, e ? atoi(e) : 8); }();
    auto cloud = [&](int f, int keep) -> std::vector<float>& {
        auto it = std::find(s->scan_lru.begin(), s->scan_lru.end(), f);
        if (it != s->scan_lru.end()) s->scan_lru.erase(it);
        else {
            while ((int)s->scan_lru.size() >= cap) {
                auto victim = s->scan_lru.begin();
                if (*victim == keep) ++victim;
                std::vector<float>().swap(s->scan_cache[(size_t)*victim]);
                s->scan_lru.erase(victim);
            }
            load_pcd(s->files[(size_t)f], &s->scan_cache[(size_t)f]);
        }
        s->scan_lru.push_back(f);
        return s->scan_cache[(size_t)f];
    };
    std::vector<float>& a = cloud(i, j);
    std::vector<float>& b = cloud(j, i);
    return pwicp_overlap_ratio(s->w0()->ctx, a.data(), (int)(a.size() / 4), b.data(), (int)(b.size() / 4), s->cfg.DTinit, ratio) == PWICP_OK;
}

// calAdaptivePairSequence (R.cpp:552-589).  table (optional): fileCount x fileCount overlap ratios computed elsewhere, entry
// [i * fileCount + j] for target i, source j, NaN = not computed; whatever the scan needs beyond it is computed on the spot.
bool adaptive_pair_sequence(pwicp_series* s, float overlapThd, const std::string& pairFile, const float* table = nullptr) {
    const int fileCount = (int)s->files.size(), startEpoch = s->startEpoch;
    int idxTarget = startEpoch;
    s->regPairs.clear();
    for (int j = startEpoch + 1; j < fileCount; ++j) {
        float ratio = 0;
        for (int i = idxTarget; i < j; ++i) {
            const float known = table ? table[(size_t)i * fileCount + j] : NAN;
            if (known == known) ratio = known;
            else if (!series_overlap(s, i, j, &ratio)) return false;
            idxTarget = i;
            if (ratio > overlapThd) break;
        }
        s->regPairs[j - startEpoch] = idxTarget - startEpoch;
        std::cout << "Pair: " << idxTarget - startEpoch << " - " << j - startEpoch << ";  Overlap ratio = " << 100 * ratio << "% \n";
    }
    std::vector<std::vector<float>>().swap(s->scan_cache);
    s->scan_lru.clear();
    if (!pairFile.empty()) {
        std::ofstream pf(pairFile);
        if (!pf) { std::cerr << "Error: Cannot open adaptivePairFile!\n"; return false; }
        for (auto& kv : s->regPairs) pf << kv.first << " " << kv.second << std::endl;
    }
    s->adaptive_deferred = false;
    return true;
}

// The HIP runtime serves a process's streams from FOUR hardware queues unless $GPU_MAX_HW_QUEUES says otherwise (read when the
// runtime starts); a series worker runs five streams.  The library never edits the host's environment (csrc/api.hip): the
// processes that are ours set the variable themselves (bench.py, pwicp_demo, the Python binding at import), any other host gets
// this line once ($PWICP_QUIET silences it).
void hw_queue_hint() {
    static std::once_flag once;
    std::call_once(once, [] {
        if (getenv("PWICP_QUIET")) return;
        const char* v = getenv("GPU_MAX_HW_QUEUES");
        const int q = v ? atoi(v) : 4;
        if (q < 5)
            std::cerr << "pwicp: hint: the HIP runtime serves this process's streams from " << q << " hardware queues and a series "
                         "worker runs five streams; GPU_MAX_HW_QUEUES=8 in the environment before the first HIP call makes a series ~10 % faster\n";
    });
}

}  // namespace

extern "C" {

PWICP_API int pwicp_series_open(const char* confile, int startEpoch, int epochNum, int pairMode, float overlapThd, int device,
                                const int32_t* adaptive_targets, int n_adaptive, pwicp_series** out) {
    if (!confile || !out) return PWICP_E_INVALID;
    *out = nullptr;
    hw_queue_hint();
    std::unique_ptr<pwicp_series> s(new pwicp_series);
    std::cout << "Loading parameter configuration file: " << confile << "\n\n";
    if (!read_config(confile, &s->cfg)) { std::cerr << "Error: Cannot open configuration file! Aborting.\n\n"; return PWICP_E_INVALID; }
    s->outputFolder = s->cfg.FolderFilePath2;
    const int fileCount = extract_all_files(s->cfg.FolderFilePath1, &s->files, &s->times);
    std::cout << "--->>> " << fileCount << " scan files are successfully extracted. \n\n";
    if (startEpoch < 0 || epochNum > fileCount || startEpoch >= epochNum) { std::cerr << "Error: epoch range outside the folder content.\n"; return PWICP_E_INVALID; }
    s->startEpoch = startEpoch; s->epochNum = epochNum; s->pairMode = pairMode; s->device = device;
    if (pairMode < 0) {
        if (adaptive_targets) {                       // map computed elsewhere
            if (n_adaptive != fileCount - startEpoch - 1) { std::cerr << "Error: adaptive pair map has the wrong length.\n"; return PWICP_E_INVALID; }
            for (int k = 0; k < n_adaptive; ++k) s->regPairs[k + 1] = adaptive_targets[k];
        } else if (n_adaptive < 0) {                  // deferred: pwicp_series_overlap_ratios / pwicp_series_adaptive_from_ratios follow
            s->adaptive_deferred = true;
        } else {
            std::cout << "--->>> Adaptive pair sequence determination... \n";
            if (!adaptive_pair_sequence(s.get(), overlapThd, "RegPairFile.txt")) { for (auto& w : s->workers) w->close(); return PWICP_E_INTERNAL; }
        }
    }
    *out = s.release();
    return PWICP_OK;
}

PWICP_API int pwicp_host_threads(void) { return pwhost::host_threads(); }

// frees what closed series have left parked per device (contexts, front-end work spaces): for hosts that are done with series
PWICP_API void pwicp_series_release_parked(void) { (void)WorkerParking::get().release_all(); }

PWICP_API void pwicp_series_close(pwicp_series* s) {
    if (!s) return;
    { std::lock_guard<std::mutex> g(s->lx.mu); s->lx.done.clear(); s->lx.closed = true; }      // (device-side targets go before their context)
    s->lx.cv.notify_all();
    for (auto& w : s->workers) w->close();
    delete s;
}

PWICP_API int pwicp_series_num_pairs(const pwicp_series* s) { return s ? s->num_pairs() : 0; }
PWICP_API int pwicp_series_num_scans(const pwicp_series* s) { return s ? (int)s->files.size() : 0; }

PWICP_API int pwicp_series_pair_epochs(const pwicp_series* s, int pair, int* target_index, int* source_index, long* source_stamp) {
    if (!s || pair < 0 || pair >= s->num_pairs()) return PWICP_E_INVALID;
    if (target_index) *target_index = s->ref_index(pair);
    if (source_index) *source_index = s->startEpoch + pair + 1;
    if (source_stamp) *source_stamp = s->times[(size_t)(s->startEpoch + pair + 1)];
    return PWICP_OK;
}

// The adaptive pair map in pieces, for a computation sharded over processes (host/comm.cpp, pwicp_amd/series.py): the overlap
// ratios of candidate pairs are independent of each other (R.cpp:593-614), only the scan that picks the targets is sequential
// (R.cpp:552-589).  ij: n pairs of file indices (target, source).
PWICP_API int pwicp_series_overlap_ratios(pwicp_series* s, const int32_t* ij, int n, float* ratios) {
    if (!s || !ij || !ratios || n < 0) return PWICP_E_INVALID;
    for (int k = 0; k < n; ++k)
        if (!series_overlap(s, ij[2 * k], ij[2 * k + 1], &ratios[k])) return s->w0()->ctx ? PWICP_E_INVALID : PWICP_E_NO_DEVICE;
    return PWICP_OK;
}

// table: #files x #files ratios, [i * #files + j] for target i and source j, NaN where unknown (computed on the spot if the scan
// gets there).  write_pair_file != 0: RegPairFile.txt in the working directory, as the reference does (R.cpp:578-586).
PWICP_API int pwicp_series_adaptive_from_ratios(pwicp_series* s, const float* table, float overlapThd, int write_pair_file) {
    if (!s || s->pairMode >= 0) return PWICP_E_INVALID;
    std::cout << "--->>> Adaptive pair sequence determination... \n";
    return adaptive_pair_sequence(s, overlapThd, write_pair_file ? "RegPairFile.txt" : "", table) ? PWICP_OK : PWICP_E_INTERNAL;
}

// wall time the pairs run so far have spent per stage (ms): [0] reading scans, [1] GPU preparation (voxel grid, SOR, reduction;
// the front ends of earlier clouds run beside it on their own streams), [2] what was left of the front ends after that,
// [3] registrations (upload, patches, loop); [4] = raw scan bytes handed to the GPU.  With several devices the stages of the
// workers overlap: the entries are sums over the workers.
PWICP_API int pwicp_series_stage_times(pwicp_series* s, double* ms5) {
    if (!s || !ms5) return PWICP_E_INVALID;
    std::lock_guard<std::mutex> g(s->stage_mu);
    for (int k = 0; k < 4; ++k) ms5[k] = s->stage_ms[k];
    ms5[4] = (double)s->stage_bytes;
    return PWICP_OK;
}

// adaptive map as n = (#files - startEpoch - 1) targets, entry k = target of source k+1 (relative to startEpoch)
PWICP_API int pwicp_series_adaptive_targets(const pwicp_series* s, int32_t* targets, int n) {
    if (!s || !targets || n != (int)s->regPairs.size()) return PWICP_E_INVALID;
    for (int k = 0; k < n; ++k) { auto it = s->regPairs.find(k + 1); targets[k] = it == s->regPairs.end() ? -1 : it->second; }
    return PWICP_OK;
}

// labels of a target cloud of a series: taken from the rank that segments it when they were announced
// (pwicp_series_expect_target_labels), else made here - and then offered to pwicp_series_wait_target_labels
static bool target_labels(pwicp_series* s, int scan, const std::shared_ptr<Prepared>& t, AuxContexts* aux) {
    auto& lx = s->lx;
    bool expected;
    { std::lock_guard<std::mutex> g(lx.mu); expected = lx.expected.count(scan) > 0; }
    if (expected) {
        double tmo_s = 600.0;
        if (const char* e = std::getenv("PWICP_LABEL_TIMEOUT_S")) tmo_s = std::max(atof(e), 1.0);
        timeline("target labels: waiting for the rank that segments it", scan);
        std::unique_lock<std::mutex> g(lx.mu);
        const bool came = lx.cv.wait_for(g, std::chrono::duration<double>(tmo_s), [&] { return lx.supplied.count(scan) > 0; });
        if (came && lx.supplied[scan].m == t->m && (int)lx.supplied[scan].lab.size() == t->m) {
            // COPIED, and kept until the series closes: the exchange supplies a scan once, but it may have several takers in this
            // process - one worker per device prepares the shared target on ITS device (pwicp_series_set_devices), and a target
            // dropped after a failed window is prepared again in a later one.  (a second taker used to wait out the
            // whole time-out, 600 s, for labels the first one had moved away.)
            t->lab = lx.supplied[scan].lab;
            t->nsv = lx.supplied[scan].nsv;
            t->segmented = true;
            ++lx.n_received;
            g.unlock();
            timeline("target labels: received", scan);
            return true;
        }
        std::cerr << "[pwicp] labels of target scan " << scan << (came ? " do not fit this rank's preprocessed cloud" : " did not arrive")
                  << ": segmenting it here.\n";
        lx.supplied.erase(scan);
        lx.expected.erase(scan);           // ... and later takers of this process segment it at once instead of waiting again
    }
    const bool ok = prepare_labels(t.get(), aux);
    { std::lock_guard<std::mutex> g(lx.mu); lx.done[scan] = ok ? t : nullptr; if (ok) ++lx.n_segmented; }
    lx.cv.notify_all();
    return ok;
}

// Iterations of the pair loop R.cpp:89-150 (without their file output) for any subset of the pairs.  The pairs are
// independent, and so are the setup stages of their clouds: the pairs are taken in windows; within a window the scans
// are read on host threads, the GPU parts of the preparation run one after the other (tens of ms each), the serial
// host parts (~1.2 s per 1 M points and cloud) run side by side on host threads, then the registrations run on the
// GPU.  The results do not depend on the window (every stage is a pure function of its cloud).
static int run_pairs_on(pwicp_series* s, SeriesWorker* w, const int32_t* pairs, int n_pairs, pwicp_pair_record* recs) {
    if (!s || !pairs || !recs || n_pairs < 0) return PWICP_E_INVALID;
    for (int k = 0; k < n_pairs; ++k) {
        std::memset(&recs[k], 0, sizeof(recs[k]));
        recs[k].pair = pairs[k];
        recs[k].status = PWICP_E_INTERNAL;
        if (pairs[k] < 0 || pairs[k] >= s->num_pairs()) return PWICP_E_INVALID;
    }
    if (n_pairs == 0) return PWICP_OK;
    if (s->adaptive_deferred) { std::cerr << "Error: the adaptive pair map of this series has not been determined yet.\n"; return PWICP_E_INVALID; }
    timeline("run_pairs: begins");
    if (!w->need_ctx()) { for (int k = 0; k < n_pairs; ++k) recs[k].status = PWICP_E_NO_DEVICE; return PWICP_E_NO_DEVICE; }
    timeline("run_pairs: worker context ready");
    const ConfigPara& cfg = s->cfg;
    const double sor_mult = 5.0;                                           // R.cpp:415-416
    int window = 32;         // (clouds whose setup stages overlap; what runs side by side inside it: $PWICP_FRONTEND_STREAMS, the CPU budget)
    if (const char* e = std::getenv("PWICP_SERIES_WINDOW")) window = std::max(1, atoi(e));
    for (int w0 = 0; w0 < n_pairs; w0 += window) {
        const int w1 = std::min(n_pairs, w0 + window), nw = w1 - w0;
        const auto t0 = std::chrono::steady_clock::now();
        StageTimer tm;
        auto t_stage = std::chrono::steady_clock::now();
        auto stage = [&](int which) {
            const auto now = std::chrono::steady_clock::now();
            std::lock_guard<std::mutex> g(s->stage_mu);
            s->stage_ms[which] += std::chrono::duration<double, std::milli>(now - t_stage).count();
            t_stage = now;
        };
        // ---- scans of this window: sources, and targets that are not prepared yet -------------------------------
        std::vector<int> refIdx((size_t)nw);
        std::vector<std::vector<float>> raw2((size_t)nw);
        std::map<int, std::vector<float>> raw1;
        for (int k = 0; k < nw; ++k) {
            refIdx[(size_t)k] = s->ref_index(pairs[w0 + k]);
            if (refIdx[(size_t)k] < 0 || refIdx[(size_t)k] >= (int)s->files.size()) return PWICP_E_INVALID;
            if (!w->targets.count(refIdx[(size_t)k])) raw1[refIdx[(size_t)k]];
        }
        {
            std::vector<std::thread> th;
            for (int k = 0; k < nw; ++k)
                th.emplace_back([&, k] { load_pcd(s->files[(size_t)(s->startEpoch + pairs[w0 + k] + 1)], &raw2[(size_t)k]); });
            for (auto& kv : raw1) th.emplace_back([&s, &kv] { load_pcd(s->files[(size_t)kv.first], &kv.second); });
            for (auto& t : th) t.join();
        }
        tm.lap("read scans (host threads)");
        timeline("scans read");
        stage(0);
        {
            long long b = 0;
            for (auto& v : raw2) b += (long long)v.size() * 4;
            for (auto& kv : raw1) b += (long long)kv.second.size() * 4;
            std::lock_guard<std::mutex> g(s->stage_mu);
            s->stage_bytes += b;
        }
        {
            long long pts = 0, cnt = 0;
            for (auto& v : raw2) if (!v.empty()) { pts += (long long)v.size() / 4; ++cnt; }
            if (cnt) w->aux->size_for(pts / cnt);
        }
        // ---- GPU parts one after the other; the host part of a cloud starts on its own thread as soon as its k-NN graph
        //      is down, so the serial host passes of earlier clouds run while the GPU prepares the later ones -------------
        PreCache pre;                                                  // (of this window: a scan that is a source and a target)
        bool share_pre = false;                                        // (no scan plays two roles: no copies are kept)
        {
            std::map<int, int> uses;
            for (int k = 0; k < nw; ++k) ++uses[s->startEpoch + pairs[w0 + k] + 1];
            for (auto& kv : raw1) ++uses[kv.first];
            for (auto& kv : uses) share_pre = share_pre || kv.second > 1;
        }
        auto resolution_of = [&](int scan, const std::vector<float>& raw, float* out) {
            auto it = pre.resolution.find(scan);
            if (it != pre.resolution.end()) { *out = it->second; return true; }
            if (pwicp_pc_resolution_dev(w->ctx, raw.data(), (int)(raw.size() / 4), out) != PWICP_OK) return false;
            pre.resolution[scan] = *out;
            return true;
        };
        std::vector<char> ok((size_t)nw, 1);
        std::vector<std::thread> th;                                   // front ends of the targets prepared in this window
        std::vector<std::thread> th_src((size_t)nw);                   // front end of source k (not joinable: none was started)
        std::vector<char> okt(raw1.size(), 1);
        std::vector<Prepared> src((size_t)nw);
        th.reserve(raw1.size());
        {
            size_t ti = 0;
            for (auto& kv : raw1) {
                auto t = std::make_shared<Prepared>();
                float Res1 = cfg.PCres1;
                bool good = !kv.second.empty();
                if (good && !cfg.isSetResSVsize && !resolution_of(kv.first, kv.second, &Res1)) good = false;
                const float SVRes1 = cfg.isSetResSVsize ? cfg.SVsize1 : Res1 * 10;                       // R.cpp:635-640
                if (good) good = prepare_gpu(w->ctx, kv.second, Res1, SVRes1, sor_mult, nullptr, t.get(), share_pre ? &pre : nullptr, kv.first);
                if (good) {
                    w->targets[kv.first] = t;
                    char* flag = &okt[ti];
                    AuxContexts* aux = w->aux.get();
                    const int scan = kv.first;
                    th.emplace_back([s, scan, t, flag, aux] { *flag = target_labels(s, scan, t, aux) ? 1 : 0; });
                } else {
                    { std::lock_guard<std::mutex> g(s->lx.mu); s->lx.done[kv.first] = nullptr; }      // (a waiter learns it at once)
                    s->lx.cv.notify_all();
                }
                std::vector<float>().swap(kv.second);
                ++ti;
            }
        }
        for (int k = 0; k < nw; ++k) {
            auto it = w->targets.find(refIdx[(size_t)k]);
            float Res2 = cfg.PCres2;
            bool good = it != w->targets.end() && !raw2[(size_t)k].empty();
            const int scan2 = s->startEpoch + pairs[w0 + k] + 1;
            if (good && !cfg.isSetResSVsize && !resolution_of(scan2, raw2[(size_t)k], &Res2)) good = false;
            const float SVRes2 = cfg.isSetResSVsize ? cfg.SVsize2 : Res2 * 10;
            timeline("prep of source begins", k);
            if (good) good = prepare_gpu(w->ctx, raw2[(size_t)k], Res2, SVRes2, sor_mult, it->second->shift, &src[(size_t)k], share_pre ? &pre : nullptr, scan2);
            timeline("prep of source done", k);
            ok[(size_t)k] = good ? 1 : 0;
            if (good) th_src[(size_t)k] = std::thread([&ok, &src, k, w] { ok[(size_t)k] = prepare_labels(&src[(size_t)k], w->aux.get()) ? 1 : 0; });
            std::vector<float>().swap(raw2[(size_t)k]);
        }
        tm.lap("voxel grid + SOR, k-NN graphs (GPU)");
        stage(1);
        for (auto& t : th) t.join();                                   // (the targets' front ends were started first)
        {
            size_t ti = 0;
            for (auto& kv : raw1) { if (!okt[ti]) w->targets.erase(kv.first); ++ti; }
        }
        // ---- registrations: pair k as soon as ITS source is segmented, on the worker's own stream, beside the front ends of the
        //      later sources (the front ends finish roughly in the order they were started; records and printed lines keep the
        //      order of the pairs) ----------------------------------------------------------------------------------------
        double reg_ms = 0.0;
        std::vector<float> reg_each((size_t)nw, 0.f);
        for (int k = 0; k < nw; ++k) {
            if (th_src[(size_t)k].joinable()) th_src[(size_t)k].join();
            timeline("registration begins", k);
            pwicp_pair_record* rec = &recs[w0 + k];
            const int pair = pairs[w0 + k], step = pair + 1, i = s->startEpoch + pair;
            const auto tp = std::chrono::steady_clock::now();
            struct RegTime {                       // (time of this registration, whichever way the iteration is left)
                double* sum; float* mine; std::chrono::steady_clock::time_point t;
                ~RegTime() { const double d = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); *sum += d; *mine = (float)d; }
            } reg_time{&reg_ms, &reg_each[(size_t)k], tp};
            std::cout << "\n//////////////////////  Process Pair_" << step << ":  Epoch-" << s->times[(size_t)refIdx[(size_t)k]] << " and Epoch-"
                      << s->times[(size_t)i + 1] << "   //////////////////////////////////////////// \n\n";
            auto it = w->targets.find(refIdx[(size_t)k]);
            PairOutput out;
            std::memset(&out, 0, sizeof(out));
            if (!ok[(size_t)k] || it == w->targets.end() || !run_prepared(w->ctx, *it->second, src[(size_t)k], cfg, &out)) {
                std::cerr << "Step " << step << " failed. Skipping to next.\n\n";                          // R.cpp:145-147
                rec->status = out.res.status != 0 ? out.res.status : PWICP_E_INTERNAL;
                continue;
            }
            rec->status = PWICP_OK;
            rec->n_outer = out.res.n_outer;
            rec->n_inner = out.res.n_inner_total;
            std::memcpy(rec->T, out.T, sizeof(rec->T));
            std::memcpy(rec->VCM, out.VCM, sizeof(rec->VCM));
            rec->n_corr = out.res.n_corr;
            rec->t_loop_ms = (float)out.res.t_loop_ms;
            std::vector<float>().swap(src[(size_t)k].p);                                                     // release early
        }
        tm.lap("normals + supervoxels (host threads, rest) with the registrations beside them");
        {
            // wall time of this stretch: the registrations' own time under [3], what is left (front ends) under [2]
            const auto now = std::chrono::steady_clock::now();
            const double all = std::chrono::duration<double, std::milli>(now - t_stage).count();
            std::lock_guard<std::mutex> g(s->stage_mu);
            s->stage_ms[3] += std::min(reg_ms, all);
            s->stage_ms[2] += std::max(all - reg_ms, 0.0);
            t_stage = now;
        }
        {
            const double setup_each = std::max(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - reg_ms, 0.0) / nw;
            for (int k = 0; k < nw; ++k)
                if (recs[w0 + k].status == PWICP_OK) recs[w0 + k].t_pair_ms = (float)(setup_each + reg_each[(size_t)k]);
        }
        // keep the reference epoch and the targets of this window, drop older ones
        for (auto it = w->targets.begin(); it != w->targets.end();) {
            bool used = it->first == s->startEpoch;
            for (int k = 0; k < nw; ++k) used = used || refIdx[(size_t)k] == it->first;
            if (used) ++it; else it = w->targets.erase(it);
        }
    }
    return PWICP_OK;
}

// ---- labels of a shared target across processes (see pwicp_series::LabelExchange) -----------------------------------------------
PWICP_API int pwicp_series_expect_target_labels(pwicp_series* s, int scan) {
    if (!s || scan < 0 || scan >= (int)s->files.size()) return PWICP_E_INVALID;
    std::lock_guard<std::mutex> g(s->lx.mu);
    s->lx.expected.insert(scan);
    return PWICP_OK;
}

PWICP_API int pwicp_series_supply_target_labels(pwicp_series* s, int scan, int m, int nsv, const int32_t* labels) {
    if (!s || scan < 0 || scan >= (int)s->files.size() || (m > 0 && !labels)) return PWICP_E_INVALID;
    {
        std::lock_guard<std::mutex> g(s->lx.mu);
        auto& sp = s->lx.supplied[scan];
        sp.m = m; sp.nsv = nsv;
        sp.lab.assign(labels, labels + std::max(m, 0));
    }
    s->lx.cv.notify_all();
    return PWICP_OK;
}

// blocks until a pwicp_series_run_pairs of this process (another thread) has segmented target `scan`, or failed to, or ended
// (PWICP_E_INTERNAL), or timeout_ms has passed (PWICP_E_INVALID).  labels == nullptr: only the sizes.
PWICP_API int pwicp_series_wait_target_labels(pwicp_series* s, int scan, int timeout_ms, int* m, int* nsv, int32_t* labels, int cap) {
    if (!s || scan < 0 || scan >= (int)s->files.size()) return PWICP_E_INVALID;
    auto& lx = s->lx;
    std::unique_lock<std::mutex> g(lx.mu);
    const bool came = lx.cv.wait_for(g, std::chrono::milliseconds(std::max(timeout_ms, 0)), [&] { return lx.done.count(scan) > 0 || lx.closed; });
    if (!came) return PWICP_E_INVALID;
    auto it = lx.done.find(scan);
    if (it == lx.done.end() || !it->second) return PWICP_E_INTERNAL;
    const Prepared& t = *it->second;
    if (m) *m = t.m;
    if (nsv) *nsv = t.nsv;
    if (labels) {
        if (cap < t.m) return PWICP_E_INVALID;
        std::memcpy(labels, t.lab.data(), (size_t)t.m * sizeof(int32_t));
    }
    return PWICP_OK;
}

// how many targets of this series took their labels from another rank / were segmented by this one (so far)
PWICP_API int pwicp_series_target_label_counts(pwicp_series* s, int* received, int* segmented) {
    if (!s) return PWICP_E_INVALID;
    std::lock_guard<std::mutex> g(s->lx.mu);
    if (received) *received = s->lx.n_received;
    if (segmented) *segmented = s->lx.n_segmented;
    return PWICP_OK;
}

// tells waiters that no run is (any longer) going to segment anything: called by the client after its pwicp_series_run_pairs
PWICP_API void pwicp_series_close_target_labels(pwicp_series* s) {
    if (!s) return;
    { std::lock_guard<std::mutex> g(s->lx.mu); s->lx.closed = true; }
    s->lx.cv.notify_all();
}

PWICP_API int pwicp_series_run_pairs(pwicp_series* s, const int32_t* pairs, int n_pairs, pwicp_pair_record* recs) {
    if (!s || !pairs || !recs || n_pairs < 0) return PWICP_E_INVALID;
    s->w0();
    const int G = (int)s->workers.size();
    if (G == 1 || n_pairs <= 1) return run_pairs_on(s, s->workers[0].get(), pairs, n_pairs, recs);
    // several devices: pair k of this call -> worker k mod G, one host thread per worker; every worker prepares the targets it
    // needs on its own device.  Same records as on one device (every stage is a pure function of its clouds).
    std::vector<std::vector<int32_t>> part((size_t)G);
    std::vector<std::vector<int>> where((size_t)G);
    for (int k = 0; k < n_pairs; ++k) { part[(size_t)(k % G)].push_back(pairs[k]); where[(size_t)(k % G)].push_back(k); }
    std::vector<std::vector<pwicp_pair_record>> out((size_t)G);
    std::vector<int> rc((size_t)G, PWICP_OK);
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g) {
        out[(size_t)g].resize(part[(size_t)g].size() ? part[(size_t)g].size() : 1);
        th.emplace_back([&, g] {
            rc[(size_t)g] = run_pairs_on(s, s->workers[(size_t)g].get(), part[(size_t)g].data(), (int)part[(size_t)g].size(), out[(size_t)g].data());
        });
    }
    for (auto& t : th) t.join();
    int ret = PWICP_OK;
    for (int g = 0; g < G; ++g) {
        for (size_t j = 0; j < part[(size_t)g].size(); ++j) recs[where[(size_t)g][j]] = out[(size_t)g][j];
        if (rc[(size_t)g] != PWICP_OK && ret == PWICP_OK) ret = rc[(size_t)g];
    }
    return ret;
}

// Devices of a series (before the first pair runs): n >= 1 HIP device ids, duplicates allowed (two workers sharing one
// GPU: functional tests on a 1-GPU box).  Replaces the single device given to pwicp_series_open.
PWICP_API int pwicp_series_set_devices(pwicp_series* s, const int32_t* devices, int n) {
    if (!s || !devices || n < 1) return PWICP_E_INVALID;
    std::vector<std::unique_ptr<SeriesWorker>> nw;
    for (int g = 0; g < n; ++g) {
        if (g == 0 && !s->workers.empty() && s->workers[0]->device == devices[0]) { nw.push_back(std::move(s->workers[0])); continue; }
        nw.emplace_back(new SeriesWorker);
        nw.back()->device = devices[g];
    }
    for (auto& w : s->workers) if (w) w->close();
    s->workers.swap(nw);
    s->device = devices[0];
    return PWICP_OK;
}
PWICP_API int pwicp_series_num_devices(const pwicp_series* s) { return s ? std::max<int>(1, (int)s->workers.size()) : 0; }

// one iteration of the pair loop
PWICP_API int pwicp_series_run_pair(pwicp_series* s, int pair, pwicp_pair_record* rec) {
    if (!s || !rec || pair < 0 || pair >= s->num_pairs()) return PWICP_E_INVALID;
    const int32_t p = pair;
    const int rc = pwicp_series_run_pairs(s, &p, 1, rec);
    return rc != PWICP_OK ? rc : rec->status;
}

// File output of the series from the records of all pairs (any order; failed or missing pairs are skipped as the
// reference skips a failed step): per-pair TransMatrix files, TransMatrices.txt, TransParameters.txt (R.cpp:111-180),
// then the composition to the reference epoch and the accuracy report (R.cpp:197-211).
PWICP_API int pwicp_series_write_results(pwicp_series* s, const pwicp_pair_record* recs, int n_recs) {
    if (!s || (!recs && n_recs > 0)) return PWICP_E_INVALID;
    const int n = s->num_pairs();
    std::vector<const pwicp_pair_record*> byPair((size_t)n, nullptr);
    for (int k = 0; k < n_recs; ++k)
        if (recs[k].pair >= 0 && recs[k].pair < n && recs[k].status == PWICP_OK) byPair[(size_t)recs[k].pair] = &recs[k];
    const std::string& outputFolder = s->outputFolder;
    const std::string fTM = outputFolder + "TransMatrices.txt", fTP = outputFolder + "TransParameters.txt";
    std::ofstream oTM(fTM.c_str()), oTP(fTP.c_str());
    if (!oTM || !oTP) { std::cerr << "Error: Unable to open output file(s).\n"; return PWICP_E_INTERNAL; }
    oTP << trans_parameters_header() << std::endl;
    int done = 0;
    for (int p = 0; p < n; ++p) {
        const pwicp_pair_record* r = byPair[(size_t)p];
        if (!r) continue;
        const long stamp = s->times[(size_t)(s->startEpoch + p + 1)];
        const std::string prefix = outputFolder + std::to_string(stamp) +
                                   (s->pairMode == 0 ? "_Direct2Ref_" : s->pairMode > 0 ? "_Fixed_" : "_Adaptive_");
        if (!write_transmatrix_file(prefix + "TransMatrix.txt", r->T, r->VCM)) continue;
        float ang[3], para[6];
        matrix2angle(r->T, ang);                                                  // R.cpp:464-480
        para[0] = (float)(ang[0] * ARC_TO_GON); para[1] = (float)(ang[1] * ARC_TO_GON); para[2] = (float)(ang[2] * ARC_TO_GON);
        para[3] = r->T[3]; para[4] = r->T[7]; para[5] = r->T[11];
        append_transmatrices(oTM, stamp, r->T, r->VCM);
        append_transparameters(oTP, stamp, para, r->VCM);
        ++done;
    }
    oTM.close();
    oTP.close();
    if (done != n) {
        std::cerr << "Warning: " << n - done << " pair(s) failed; composition to the reference epoch skipped.\n";
        return done > 0 ? PWICP_OK : PWICP_E_INTERNAL;
    }
    if (!trans_to_reference(fTM, s->pairMode, s->regPairs, n, outputFolder + "TransMatrices_toRef.txt", outputFolder + "TransParameters_toRef.txt"))
        return PWICP_E_INTERNAL;
    // accuracy report only if the ground-truth file of the synthetic data set is present (the reference hard-codes
    // this path and exits when it is missing, R.cpp:207-211, 1189-1192)
    abs_error_report(outputFolder + "TransMatrices_toRef.txt", "data/data_synthetic/defined_transformations.txt", s->epochNum, s->startEpoch,
                     outputFolder + "TransPara_AbsError.txt");
    return PWICP_OK;
}

PWICP_API bool PiecewiseICP_4D_call(const char* confile, int startEpoch, int epochNum, int pairMode, float overlapThd) {
    if (!confile) return false;
    // one rank of a multi-process launch (one process per GPU, RCCL gather of the records): opt-in through the environment
    if (const char* e = getenv("PWICP_RCCL")) {
        const char* ws = getenv("WORLD_SIZE");
        if (atoi(e) != 0 && ws && atoi(ws) > 1) {
            const char* rk = getenv("RANK");
            std::string idf = getenv("PWICP_RCCL_ID_FILE") ? getenv("PWICP_RCCL_ID_FILE")
                                                           : std::string("/tmp/pwicp_rccl_") + (getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0") + ".id";
            return pwicp_series_run_distributed(confile, startEpoch, epochNum, pairMode, overlapThd, rk ? atoi(rk) : 0, atoi(ws), env_device(),
                                                idf.c_str());
        }
    }
    pwicp_series* s = nullptr;
    const std::vector<int32_t> devs = env_devices();            // every visible GPU unless the environment says otherwise
    if (pwicp_series_open(confile, startEpoch, epochNum, pairMode, overlapThd, devs[0], nullptr, 0, &s) != PWICP_OK) return false;
    if (devs.size() > 1 && pwicp_series_set_devices(s, devs.data(), (int)devs.size()) != PWICP_OK) { pwicp_series_close(s); return false; }
    const int n = pwicp_series_num_pairs(s);
    std::vector<pwicp_pair_record> recs((size_t)std::max(n, 1));
    std::vector<int32_t> all((size_t)std::max(n, 1));
    for (int p = 0; p < n; ++p) all[(size_t)p] = p;
    const bool device_ok = pwicp_series_run_pairs(s, all.data(), n, recs.data()) != PWICP_E_NO_DEVICE;
    const bool ok = device_ok && pwicp_series_write_results(s, recs.data(), n) == PWICP_OK;
    pwicp_series_close(s);
    return ok;
}

}  // extern "C"
