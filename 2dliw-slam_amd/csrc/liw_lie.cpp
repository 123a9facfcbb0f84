// liw_lie.cpp — C exports of the shared host/device geometry header (liw_dual.hpp) for host code above the C ABI.
#include "../../include/liw_lie.h"

#include "liw_dual.hpp"

void liw_normalize_rotation_host(double* R9);   // liw_capi.hip

namespace {
inline liw::Iso<double> ld(const double* T) { return liw::cast_iso<double>(T, T + 9); }
inline void st(const liw::Iso<double>& A, double* T) { for (int k = 0; k < 9; ++k) T[k] = A.R.m[k]; T[9] = A.t.x; T[10] = A.t.y; T[11] = A.t.z; }
}  // namespace

extern "C" {
void liw_lie_exp_so3(const double* a, double* R9) { const liw::M3<double> R = liw::exp_so3(liw::cast_v3<double>(a)); for (int k = 0; k < 9; ++k) R9[k] = R.m[k]; }
void liw_lie_log_SO3(const double* R9, double* a) { const liw::V3<double> v = liw::log_SO3(liw::cast_m3<double>(R9)); a[0] = v.x; a[1] = v.y; a[2] = v.z; }
void liw_lie_make_tf(const double* p, const double* a, double* T) { st(liw::make_tf(liw::cast_v3<double>(p), liw::cast_v3<double>(a)), T); }
void liw_lie_log_SE3(const double* T, double* p, double* a) { p[0] = T[9]; p[1] = T[10]; p[2] = T[11]; liw_lie_log_SO3(T, a); }
void liw_lie_mul(const double* A, const double* B, double* C) { st(liw::mul(ld(A), ld(B)), C); }
void liw_lie_inverse(const double* A, double* B) { st(liw::inverse(ld(A)), B); }
void liw_lie_apply(const double* T, const double* x, double* y) {
    const liw::Iso<double> A = ld(T);
    const liw::V3<double> r = liw::mul(A.R, liw::cast_v3<double>(x));
    y[0] = r.x + A.t.x; y[1] = r.y + A.t.y; y[2] = r.z + A.t.z;
}
void liw_lie_from_matrix16(const double* M, int normalize, double* T) {
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) T[i * 3 + j] = M[i * 4 + j]; T[9 + i] = M[i * 4 + 3]; }
    if (normalize) liw_normalize_rotation_host(T);
}
}
