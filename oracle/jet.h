// ORACLE — TEST INFRASTRUCTURE ONLY.  Never linked into, imported by or executed from the
// product path (2dliw-slam_amd/); only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may touch anything under oracle/.
//
// PARITY UNPINNED: the reference (/root/reference) has no tests or golden vectors and cannot
// be compiled in this image (no Ceres, no Eigen, no ROS).  This file restates the published
// forward-mode dual number ("Jet") that Ceres Solver's AutoDiffCostFunction evaluates the
// reference's functors with (reference call sites: src/utilies/common.h:198-239,
// src/factor/laser_factor.h:96, src/factor/imu_factor.h:94, src/factor/wheel_factor.h:78,
// src/factor/ground_factor.h:51,85).  Ceres is a third-party dependency that is NOT vendored
// in the reference (CMakeLists.txt:18, docker/Dockerfile:46 => distro libceres-dev 1.14.x).
//
// A Jet<N> is a value `a` plus N partial derivatives `v[]`; every operator below applies the
// textbook chain rule exactly as ceres/jet.h documents it (f/g uses the "divide once, reuse
// the quotient" form; sqrt/asin/atan2 derivative formulas are the closed forms), so a
// derivative at a non-smooth point (sqrt at 0, asin at +-1) comes out NaN/Inf like it does
// in the reference.
#pragma once
#include <cmath>

namespace oracle {

template <int N>
struct Jet {
    double a;
    double v[N];
    Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
    Jet(double x) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT(implicit)
    Jet(double x, int k) : a(x) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
    Jet& operator+=(const Jet& y) { *this = *this + y; return *this; }
    Jet& operator-=(const Jet& y) { *this = *this - y; return *this; }
    Jet& operator*=(const Jet& y) { *this = *this * y; return *this; }
    Jet& operator/=(const Jet& y) { *this = *this / y; return *this; }
};

template <int N> inline Jet<N> operator+(const Jet<N>& f) { return f; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) {
    Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h;
}
template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h;
}
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h;
}
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h; h.a = f.a * g.a;
    for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a;
    return h;
}
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
    // h = f/g ; dh = (df - h dg)/g
    const double g_a_inverse = 1.0 / g.a;
    const double f_a_by_g_a = f.a * g_a_inverse;
    Jet<N> h; h.a = f_a_by_g_a;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
    return h;
}
// mixed scalar forms
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> h = f; h.a += s; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> h = f; h.a -= s; return h; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) {
    Jet<N> h; h.a = s - f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h;
}
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) {
    Jet<N> h; h.a = f.a * s; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s; return h;
}
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) {
    const double s_inverse = 1.0 / s;
    Jet<N> h; h.a = f.a * s_inverse; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * s_inverse; return h;
}
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) {
    const double minus_s_g_a_inverse2 = -s / (g.a * g.a);
    Jet<N> h; h.a = s / g.a; for (int i = 0; i < N; ++i) h.v[i] = g.v[i] * minus_s_g_a_inverse2; return h;
}

// comparisons look at the value part only (this is what makes the reference's `if`s on Jets
// data-dependent but derivative-free)
#define ORACLE_JET_CMP(op)                                                                      \
    template <int N> inline bool operator op(const Jet<N>& f, const Jet<N>& g) { return f.a op g.a; } \
    template <int N> inline bool operator op(const Jet<N>& f, double g) { return f.a op g; }     \
    template <int N> inline bool operator op(double f, const Jet<N>& g) { return f op g.a; }
ORACLE_JET_CMP(<) ORACLE_JET_CMP(<=) ORACLE_JET_CMP(>) ORACLE_JET_CMP(>=) ORACLE_JET_CMP(==) ORACLE_JET_CMP(!=)
#undef ORACLE_JET_CMP

template <int N> inline Jet<N> sqrt(const Jet<N>& f) {
    const double tmp = std::sqrt(f.a);
    const double two_a_inverse = 1.0 / (2.0 * tmp);
    Jet<N> h; h.a = tmp; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * two_a_inverse; return h;
}
template <int N> inline Jet<N> sin(const Jet<N>& f) {
    const double c = std::cos(f.a);
    Jet<N> h; h.a = std::sin(f.a); for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h;
}
template <int N> inline Jet<N> cos(const Jet<N>& f) {
    const double ms = -std::sin(f.a);
    Jet<N> h; h.a = std::cos(f.a); for (int i = 0; i < N; ++i) h.v[i] = ms * f.v[i]; return h;
}
template <int N> inline Jet<N> asin(const Jet<N>& f) {
    const double tmp = 1.0 / std::sqrt(1.0 - f.a * f.a);
    Jet<N> h; h.a = std::asin(f.a); for (int i = 0; i < N; ++i) h.v[i] = tmp * f.v[i]; return h;
}
template <int N> inline Jet<N> atan2(const Jet<N>& g, const Jet<N>& f) {
    // atan2(g, f): d = (f dg - g df) / (f^2 + g^2)
    const double tmp = 1.0 / (f.a * f.a + g.a * g.a);
    Jet<N> h; h.a = std::atan2(g.a, f.a);
    for (int i = 0; i < N; ++i) h.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
    return h;
}
template <int N> inline Jet<N> floor(const Jet<N>& f) { return Jet<N>(std::floor(f.a)); }

// double overloads so templated code can call oracle::sqrt(T) etc. for T = double
inline double sqrt(double x) { return std::sqrt(x); }
inline double sin(double x) { return std::sin(x); }
inline double cos(double x) { return std::cos(x); }
inline double asin(double x) { return std::asin(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }
inline double floor(double x) { return std::floor(x); }

inline double value_of(double x) { return x; }
template <int N> inline double value_of(const Jet<N>& x) { return x.a; }

}  // namespace oracle
