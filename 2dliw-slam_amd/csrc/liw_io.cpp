// liw_io.cpp — TUM trajectory and `record` timing-table outputs (host C++).
// Formats restate reference src/trajectory/trajectory.cpp:59-67,549-559 and src/utilies/record.h:19-93 byte for byte;
// numbers are rendered with snprintf ("%.10f" == std::fixed << setprecision(10); "%g" == the default ostream format).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/liw_io.h"
#include "liw_dual.hpp"

void liw_normalize_rotation_host(double* R9);   // liw_capi.hip

namespace {
// Eigen::Quaterniond(Matrix3d) (same branches as lie.h / params.cpp round trip): out = x y z w
void rotmat_to_quat(const liw::M3<double>& M, double* q) {
    double t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (M(2, 1) - M(1, 2)) * t; q[1] = (M(0, 2) - M(2, 0)) * t; q[2] = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (M(k, j) - M(j, k)) * t; q[j] = (M(j, i) + M(i, j)) * t; q[k] = (M(k, i) + M(i, k)) * t;
    }
}
liw::Iso<double> wheel_extrinsic(const liw_params& prm) {
    double R[9], t[3];
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = prm.T_imu_to_wheel[i * 4 + j]; t[i] = prm.T_imu_to_wheel[i * 4 + 3]; }
    if (prm.normalize_extrinsics) liw_normalize_rotation_host(R);
    return liw::cast_iso<double>(R, t);
}
void app(std::string& s, const char* fmt, double v) { char b[64]; snprintf(b, sizeof b, fmt, v); s += b; }
void app_u(std::string& s, unsigned long long v) { char b[32]; snprintf(b, sizeof b, "%llu", v); s += b; }
}  // namespace

struct liw_tum_writer { FILE* f; liw_params prm; double last_time; };
struct liw_record {
    std::vector<std::chrono::steady_clock::time_point> open;
    std::map<std::string, std::vector<uint64_t>> time_recorder, others_recorder;
};

extern "C" {

void liw_tum_pose(const liw_params* prm, const double* p, const double* q, double* out7) {
    const liw::Iso<double> T = liw::mul(liw::make_tf(liw::cast_v3<double>(p), liw::cast_v3<double>(q)), wheel_extrinsic(*prm));
    out7[0] = T.t.x; out7[1] = T.t.y; out7[2] = T.t.z;
    rotmat_to_quat(T.R, out7 + 3);
}
int liw_tum_format_line(double time, const double* v, char* buf, int cap) {
    return snprintf(buf, cap > 0 ? (size_t)cap : 0, "%.10f %.10f %.10f %.10f %.10f %.10f %.10f %.10f\n", time, v[0], v[1], v[2], v[3], v[4], v[5], v[6]);
}
liw_tum_writer* liw_tum_open(const char* path, const liw_params* prm) {
    if (!path || !prm) return nullptr;
    FILE* f = fopen(path, "w");
    if (!f) return nullptr;
    fputs("#Time px py pz qx qy qz qw\n", f);
    liw_tum_writer* w = new liw_tum_writer{f, *prm, -1e300};
    return w;
}
int liw_tum_append(liw_tum_writer* w, double time, const double* p, const double* q) {
    if (!w) return LIW_EINVAL;
    double v[7];
    char line[320];
    liw_tum_pose(&w->prm, p, q, v);
    const int n = liw_tum_format_line(time, v, line, sizeof line);
    fwrite(line, 1, (size_t)n, w->f);
    fflush(w->f);                                   // std::endl in the reference
    const int rc = w->last_time >= time ? LIW_ESTATE : LIW_OK;
    w->last_time = time;
    return rc;
}
void liw_tum_close(liw_tum_writer* w) { if (w) { fclose(w->f); delete w; } }

liw_record* liw_record_create(void) { return new liw_record(); }
void liw_record_destroy(liw_record* r) { delete r; }
void liw_record_begin(liw_record* r) { if (r) r->open.push_back(std::chrono::steady_clock::now()); }
uint64_t liw_record_end(liw_record* r, const char* name) {
    if (!r || r->open.empty() || !name) return 0;
    const uint64_t us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - r->open.back()).count();
    r->open.pop_back();
    r->time_recorder[name].push_back(us);
    return us;
}
void liw_record_add_time(liw_record* r, const char* name, uint64_t us) { if (r && name) r->time_recorder[name].push_back(us); }
void liw_record_add(liw_record* r, const char* name, uint64_t v) { if (r && name) r->others_recorder[name].push_back(v); }

static void table(std::string& s, const std::map<std::string, std::vector<uint64_t>>& rec) {
    for (const auto& kv : rec) {
        s += "| " + kv.first + " | ";
        app_u(s, kv.second.size());
        s += " | ";
        uint64_t total = 0, mx = 0, mn = 999999999;
        for (uint64_t it : kv.second) { total += it; if (it > mx) mx = it; if (it < mn) mn = it; }
        double aver = 0, variance = 0;
        if (!kv.second.empty()) aver = (double)total / (double)kv.second.size();
        for (uint64_t it : kv.second) variance += ((double)it - aver) * ((double)it - aver);
        variance /= (double)kv.second.size();
        app_u(s, mx); s += " |"; app_u(s, mn); s += " |"; app(s, "%g", aver); s += " |"; app(s, "%g", variance); s += " |\n";
    }
}
int liw_record_format(const liw_record* r, char* buf, int cap) {
    if (!r) return LIW_EINVAL;
    std::string s = "time_recorder\nsize of total record type:";
    app_u(s, r->time_recorder.size());
    s += "\n\n| type name | record size | max(us) | min(us) | aver(us) | variance(${us}^2$) |\n| --- | --- | --- | --- | --- | --- |\n";
    table(s, r->time_recorder);
    s += "\nothers_recorder\nsize of total record type:";
    app_u(s, r->others_recorder.size());
    s += "\n\n| type name | record size | max | min | aver | variance |\n| --- | --- | --- | --- | --- | --- |\n";
    table(s, r->others_recorder);
    if (buf && cap > 0) { const size_t k = std::min((size_t)cap - 1, s.size()); std::memcpy(buf, s.data(), k); buf[k] = 0; }
    return (int)s.size();
}
int liw_record_write(const liw_record* r, const char* path) {
    if (!r || !path) return LIW_EINVAL;
    const int n = liw_record_format(r, nullptr, 0);
    std::vector<char> b((size_t)n + 1);
    liw_record_format(r, b.data(), n + 1);
    FILE* f = fopen(path, "w");
    if (!f) return LIW_EINVAL;
    fwrite(b.data(), 1, (size_t)n, f);
    fclose(f);
    return LIW_OK;
}

}  // extern "C"
