// File formats of the reference's entry points (host): the 11-line positional configuration file, PCD v0.7 clouds,
// and the result text files.
//
// Reference: readConfigFile src/CommonFunc.cpp:11-136, extractAllFilesFromFolder / getFiles /
// extractTimeFromFileName CommonFunc.cpp:182-236 (Win32 _findfirst there, POSIX dirent here),
// pcl::io::loadPCDFile / savePCDFileBinary as used at src/Registration.cpp:87, 252-253, 394, the writers at
// Registration.cpp:341-388 / 492-539 (TransMatrix.txt), 152-180 (TransMatrices.txt, TransParameters.txt),
// matrix2angle CommonFunc.cpp:385-407.
#include "io.h"

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <sstream>

namespace pwhost {

// ---- configuration ------------------------------------------------------------------------------------------------
static bool value_after_colon(const std::string& line, size_t skip, std::string* out) {
    const size_t c = line.find(':');
    if (c == std::string::npos || c + skip > line.size()) return false;
    *out = line.substr(c + skip);
    while (!out->empty() && (out->back() == '\r' || out->back() == '\n')) out->pop_back();
    return true;
}

bool read_config(const std::string& path, ConfigPara* c) {
    std::ifstream in(path);
    if (!in || path.empty()) { std::cerr << "Cannot open configuration file! Aborting.\n"; return false; }
    std::string line, v;
    try {
        // 1-2: paths (text after ": "), C.cpp:21-35
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 2, &v)) c->FolderFilePath1 = v;
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 2, &v)) c->FolderFilePath2 = v;
        // 3-11: numbers (text after ":"), C.cpp:37-131
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 1, &v)) c->isSetResSVsize = std::stoi(v) != 0;
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 1, &v)) c->PCres1 = std::stof(v);
        if (c->PCres1 <= 0) { std::cerr << "PCres1 out of limits! \n"; return false; }
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 1, &v)) c->PCres2 = std::stof(v);
        if (c->PCres2 <= 0) { std::cerr << "PCres2 out of limits! \n"; return false; }
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 1, &v)) c->SVsize1 = std::stof(v);
        if (c->SVsize1 < c->PCres1 || c->SVsize1 > 40 * c->PCres1) { std::cerr << "SVsize1 out of limits! \n"; return false; }
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 1, &v)) c->SVsize2 = std::stof(v);
        if (c->SVsize2 < c->PCres2 || c->SVsize2 > 40 * c->PCres2) { std::cerr << "SVsize2 out of limits! \n"; return false; }
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 1, &v)) c->isSetDTinit = std::stoi(v) != 0;
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 1, &v)) c->DTinit = std::stof(v);
        if (c->DTinit <= 0) { std::cerr << "DTinit out of limits! \n"; return false; }
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 1, &v)) c->DTmin = std::stof(v);
        if (c->DTinit < c->DTmin) { std::cerr << "DTmin out of limits! \n"; return false; }
        if (std::getline(in, line) && !line.empty() && value_after_colon(line, 1, &v)) c->isVisual = std::stoi(v) != 0;
    } catch (const std::exception&) {
        std::cerr << "Malformed configuration file: " << path << "\n";
        return false;
    }
    return true;
}

// ---- folder scan ---------------------------------------------------------------------------------------------------
static void list_files_rec(const std::string& dir, std::vector<std::string>* out) {
    DIR* d = opendir(dir.c_str());
    if (!d) return;
    while (dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        const std::string full = dir + "/" + name;
        struct stat st;
        if (stat(full.c_str(), &st) != 0) continue;
        if (S_ISDIR(st.st_mode)) list_files_rec(full, out);
        else out->push_back(full);
    }
    closedir(d);
}

int extract_all_files(const std::string& folder, std::vector<std::string>* names, std::vector<long>* times) {
    names->clear();
    times->clear();
    std::vector<std::string> all;
    list_files_rec(folder, &all);
    std::vector<std::pair<std::string, long>> v;
    for (const std::string& f : all) {
        const size_t p = f.find("Epoch_");                 // C.cpp:191, 231-236: 3 digits after "Epoch_"
        if (p == std::string::npos || p + 9 > f.size()) continue;
        try { v.emplace_back(f, std::stol(f.substr(p + 6, 3))); } catch (...) { continue; }
    }
    std::sort(v.begin(), v.end());                          // std::map order of the reference (by name) ...
    std::stable_sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.second < b.second; });   // ... then by time
    for (auto& e : v) { names->push_back(e.first); times->push_back(e.second); }
    return (int)names->size();
}

// ---- PCD -------------------------------------------------------------------------------------------------------------
bool load_pcd(const std::string& path, std::vector<float>* xyz4) {
    xyz4->clear();
    std::ifstream in(path, std::ios::binary);
    if (!in) return false;
    std::vector<std::string> fields, types;
    std::vector<int> sizes, counts;
    long npts = -1, width = 0, height = 1;
    std::string mode, line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string key;
        ss >> key;
        std::string tok;
        if (key == "FIELDS") while (ss >> tok) fields.push_back(tok);
        else if (key == "SIZE") while (ss >> tok) sizes.push_back(std::stoi(tok));
        else if (key == "TYPE") while (ss >> tok) types.push_back(tok);
        else if (key == "COUNT") while (ss >> tok) counts.push_back(std::stoi(tok));
        else if (key == "WIDTH") ss >> width;
        else if (key == "HEIGHT") ss >> height;
        else if (key == "POINTS") ss >> npts;
        else if (key == "DATA") { ss >> mode; break; }
    }
    if (npts < 0) npts = width * height;
    if (counts.empty()) counts.assign(fields.size(), 1);
    if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size() || npts < 0) return false;
    int off[3] = {-1, -1, -1}, col[3] = {-1, -1, -1}, rec = 0, ncol = 0;
    for (size_t f = 0; f < fields.size(); ++f) {
        for (int d = 0; d < 3; ++d)
            if (fields[f] == (d == 0 ? "x" : d == 1 ? "y" : "z")) {
                if (sizes[f] != 4 || (types[f] != "F" && types[f] != "f")) return false;
                off[d] = rec; col[d] = ncol;
            }
        rec += sizes[f] * counts[f];
        ncol += counts[f];
    }
    if (off[0] < 0 || off[1] < 0 || off[2] < 0) return false;
    xyz4->resize((size_t)npts * 4);
    if (mode == "binary") {
        std::vector<char> buf((size_t)npts * rec);
        in.read(buf.data(), (std::streamsize)buf.size());
        if ((size_t)in.gcount() != buf.size()) return false;
        for (long i = 0; i < npts; ++i) {
            float* o = xyz4->data() + 4 * (size_t)i;
            for (int d = 0; d < 3; ++d) std::memcpy(o + d, buf.data() + (size_t)i * rec + off[d], 4);
            o[3] = 1.0f;
        }
    } else if (mode == "ascii") {
        std::vector<double> row((size_t)ncol);
        std::string tok;
        for (long i = 0; i < npts; ++i) {
            for (int c = 0; c < ncol; ++c) {                 // via strtod: "nan" / "inf" are legal values in PCL's ascii files
                if (!(in >> tok)) return false;
                char* end = nullptr;
                row[(size_t)c] = std::strtod(tok.c_str(), &end);
                if (end == tok.c_str()) return false;
            }
            float* o = xyz4->data() + 4 * (size_t)i;
            for (int d = 0; d < 3; ++d) o[d] = (float)row[(size_t)col[d]];
            o[3] = 1.0f;
        }
    } else {
        return false;                                       // binary_compressed is not produced by the reference
    }
    // PCL keeps non-finite points of a file and makes every consumer on this path skip them (VoxelGrid, getMinMax3D,
    // KdTreeFLANN::setInputCloud and calPCresolution test pcl_isfinite when !is_dense); dropping them here is equivalent
    // for the registration (only RegisteredSourceCloud.pcd of the pair entry point loses those points)
    {
        size_t w = 0;
        const size_t np = xyz4->size() / 4;
        for (size_t i = 0; i < np; ++i) {
            const float* q = xyz4->data() + 4 * i;
            if (std::isfinite(q[0]) && std::isfinite(q[1]) && std::isfinite(q[2])) {
                if (w != i) std::memcpy(xyz4->data() + 4 * w, q, 16);
                ++w;
            }
        }
        xyz4->resize(4 * w);
    }
    return true;
}

bool save_pcd_binary(const std::string& path, const float* xyz4, int n) {
    std::ofstream out(path, std::ios::binary);
    if (!out) return false;
    out << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\n"
        << "WIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA binary\n";
    std::vector<float> packed((size_t)n * 3);
    for (int i = 0; i < n; ++i) std::memcpy(&packed[3 * (size_t)i], xyz4 + 4 * (size_t)i, 12);
    out.write(reinterpret_cast<const char*>(packed.data()), (std::streamsize)(packed.size() * sizeof(float)));
    return (bool)out;
}

// ---- small matrix helpers (Eigen::Matrix4f semantics, row-major storage) --------------------------------------------------
void mat4_mul(const float* A, const float* B, float* C) {
    float R[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = A[4 * i] * B[j];
            s = s + A[4 * i + 1] * B[4 + j];
            s = s + A[4 * i + 2] * B[8 + j];
            s = s + A[4 * i + 3] * B[12 + j];
            R[4 * i + j] = s;
        }
    std::memcpy(C, R, sizeof(R));
}

void matrix2angle(const float* T, float* ang) {            // C.cpp:385-407
    double ax, ay, az;
    if (T[8] == 1 || T[8] == -1) {
        az = 0;
        const double dlta = (double)std::atan2(T[1], T[2]);
        if (T[8] == -1) { ay = M_PI / 2; ax = az + dlta; }
        else { ay = -M_PI / 2; ax = -az + dlta; }
    } else {
        ay = (double)(-std::asin(T[8]));                    // float overload, as in the reference
        ax = std::atan2((double)T[9] / std::cos(ay), (double)T[10] / std::cos(ay));
        az = std::atan2((double)T[4] / std::cos(ay), (double)T[0] / std::cos(ay));
    }
    ang[0] = (float)ax; ang[1] = (float)ay; ang[2] = (float)az;
}

// ---- result files -------------------------------------------------------------------------------------------------------
bool write_transmatrix_file(const std::string& path, const float* T, const double* VCM) {   // R.cpp:341-388 / 492-539
    std::ofstream o(path.c_str());
    if (!o) { std::cerr << "Cannot open TransMatrix.txt for writing!\n\n"; return false; }
    float ang[3];
    matrix2angle(T, ang);
    const float tr[3] = {T[3], T[7], T[11]};
    o << "4x4 Transformation Matrix:\n";
    o << std::fixed << std::setprecision(12);
    for (int i = 0; i < 4; ++i) {
        for (int j = 0; j < 4; ++j) o << T[4 * i + j] << " ";
        o << "\n";
    }
    o << std::endl;
    o << "Rotation Angles (unit: gon):\n" << std::fixed << std::setprecision(10)
      << "Rx = " << ang[0] * ARC_TO_GON << "\n" << "Ry = " << ang[1] * ARC_TO_GON << "\n" << "Rz = " << ang[2] * ARC_TO_GON << "\n";
    o << "Translation (unit: m):\n" << "tx = " << tr[0] << "\n" << "ty = " << tr[1] << "\n" << "tz = " << tr[2] << "\n";
    o << std::endl;
    o << "6x6 Variance-Covariance Matrix of transformation parameters:\n";
    o << std::fixed << std::setprecision(12);
    for (int i = 0; i < 6; ++i) {
        for (int j = 0; j < 6; ++j) o << VCM[6 * i + j] << " ";
        o << "\n";
    }
    o << std::endl;
    o << "Standard Deviations of estimated transformation parameters:\n";
    o << std::fixed << std::setprecision(10)
      << "Std_Rx = " << 1000 * ARC_TO_GON * std::sqrt(VCM[0]) << " mgon\n"
      << "Std_Ry = " << 1000 * ARC_TO_GON * std::sqrt(VCM[7]) << " mgon\n"
      << "Std_Rz = " << 1000 * ARC_TO_GON * std::sqrt(VCM[14]) << " mgon\n"
      << "Std_tx = " << 1000 * std::sqrt(VCM[21]) << " mm\n"
      << "Std_ty = " << 1000 * std::sqrt(VCM[28]) << " mm\n"
      << "Std_tz = " << 1000 * std::sqrt(VCM[35]) << " mm\n";
    return (bool)o;
}

void append_transmatrices(std::ostream& o, long stamp, const float* T, const double* VCM) {   // R.cpp:152-167
    o << std::fixed << std::setprecision(12);
    o << stamp << "\n";
    for (int r = 0; r < 4; ++r) {
        for (int c = 0; c < 4; ++c) o << T[4 * r + c] << " ";
        o << "\n";
    }
    for (int r = 0; r < 6; ++r) {
        for (int c = 0; c < 6; ++c) o << VCM[6 * r + c] << " ";
        o << "\n";
    }
}

const char* trans_parameters_header() {
    return "Epoch  Rx[gon]  Ry[gon]  Rz[gon]  tx[m]  ty[m]  tz[m]  Std_Rx[mgon]  Std_Ry[mgon]  Std_Rz[mgon]  "
           "Std_tx[mm]  Std_ty[mm]  Std_tz[mm]";
}

void append_transparameters(std::ostream& o, long stamp, const float* para6, const double* VCM) {   // R.cpp:170-180
    o << std::fixed << std::setprecision(10);
    o << stamp << " ";
    for (int p = 0; p < 6; ++p) o << para6[p] << " ";
    o << 1000 * std::sqrt(VCM[0]) * ARC_TO_GON << " " << 1000 * std::sqrt(VCM[7]) * ARC_TO_GON << " "
      << 1000 * std::sqrt(VCM[14]) * ARC_TO_GON << " " << 1000 * std::sqrt(VCM[21]) << " " << 1000 * std::sqrt(VCM[28]) << " "
      << 1000 * std::sqrt(VCM[35]) << "\n";
}

bool read_transmatrices(const std::string& path, int n, std::vector<int>* stamps, std::vector<std::array<float, 16>>* Ts,
                        std::vector<std::array<double, 36>>* Vs) {     // R.cpp:983-1011
    std::ifstream in(path);
    if (!in) return false;
    for (int i = 0; i < n; ++i) {
        int stamp;
        std::array<float, 16> T;
        std::array<double, 36> V;
        if (!(in >> stamp)) return false;
        for (int k = 0; k < 16; ++k) if (!(in >> T[(size_t)k])) return false;
        for (int k = 0; k < 36; ++k) if (!(in >> V[(size_t)k])) return false;
        stamps->push_back(stamp); Ts->push_back(T); Vs->push_back(V);
    }
    return true;
}

}  // namespace pwhost
