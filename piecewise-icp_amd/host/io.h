// Host-side file formats and small helpers of the reference's entry points (io.cpp).
#pragma once

#include <array>
#include <ostream>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

namespace pwhost {

// PWICP_TRACE=1: wall time of the stages of an entry point on stderr
struct StageTimer {
    const bool on = std::getenv("PWICP_TRACE") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[pwicp entry point] %-34s %9.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};


constexpr double ARC_TO_GON = 63.6619772368;       // include/CommonFunc.h:40

struct ConfigPara {                                // include/CommonFunc.h:48-61
    std::string FolderFilePath1, FolderFilePath2;
    bool isSetResSVsize = false;
    float PCres1 = 0, PCres2 = 0, SVsize1 = 0, SVsize2 = 0;
    bool isSetDTinit = false;
    float DTinit = 0, DTmin = 0;
    bool isVisual = false;
};

bool read_config(const std::string& path, ConfigPara* c);
int extract_all_files(const std::string& folder, std::vector<std::string>* names, std::vector<long>* times);
bool load_pcd(const std::string& path, std::vector<float>* xyz4);
bool save_pcd_binary(const std::string& path, const float* xyz4, int n);
void mat4_mul(const float* A, const float* B, float* C);
void matrix2angle(const float* T16, float* ang3);
bool write_transmatrix_file(const std::string& path, const float* T16, const double* VCM36);
void append_transmatrices(std::ostream& o, long stamp, const float* T16, const double* VCM36);
const char* trans_parameters_header();
void append_transparameters(std::ostream& o, long stamp, const float* para6, const double* VCM36);
bool read_transmatrices(const std::string& path, int n, std::vector<int>* stamps, std::vector<std::array<float, 16>>* Ts,
                        std::vector<std::array<double, 36>>* Vs);

// preprocess.cpp
// PCA normals + supervoxel fusion + boundary refinement from a k-NN graph (n rows of k indices, the point itself first)
int segment_from_knn(const float* cloud_xyz4, int n, const int32_t* nb, int k, float sv_resolution, int32_t* labels,
                     int* n_supervoxels);
int voxel_grid(const float* in4, int n, float leaf, float* out4);
int sor_filter(const float* in4, int n, int mean_k, double std_mul, float* out4);
float pc_resolution(const float* c4, int n);

}  // namespace pwhost
