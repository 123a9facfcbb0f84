// liw_dual.hpp — device-side forward-mode scalar for gfx950 and the SO3/SE3 helpers written on it.
//
// MI355X mapping of the reference's Ceres Jets: instead of one thread carrying a Jet<N> (value + N partials),
// a LANE carries (value, ONE directional derivative).  The lanes of a wavefront that work on the same factor
// hold the same value part and different derivative directions, so one pass of the functor over a group of
// N lanes yields the full residual Jacobian column-per-lane (12 lanes: laser/wheel, 30: IMU, 6: ground).
// T = double gives the residual-only evaluation.
//
// Geometry follows the reference's templated helpers (what they compute, branch for branch):
//   lie::exp_so3 / log_SO3 / normalize_so3 / make_tf      src/utilies/common.h:119-181
// which in turn call ceres::AngleAxisToQuaternion / QuaternionToAngleAxis (ceres/rotation.h, not vendored)
// and Eigen's Quaternion(Matrix3) / toRotationMatrix / normalize (Eigen 3.3, not vendored).
#pragma once
#include <hip/hip_runtime.h>

namespace liw {

struct LJ {
    double v, d;
    __host__ __device__ LJ() : v(0.0), d(0.0) {}
    __host__ __device__ LJ(double c) : v(c), d(0.0) {}  // NOLINT(implicit): constants have zero derivative
    __host__ __device__ LJ(double v_, double d_) : v(v_), d(d_) {}
};
__host__ __device__ inline LJ operator+(const LJ& a, const LJ& b) { return LJ(a.v + b.v, a.d + b.d); }
__host__ __device__ inline LJ operator-(const LJ& a, const LJ& b) { return LJ(a.v - b.v, a.d - b.d); }
__host__ __device__ inline LJ operator-(const LJ& a) { return LJ(-a.v, -a.d); }
__host__ __device__ inline LJ operator*(const LJ& a, const LJ& b) { return LJ(a.v * b.v, a.v * b.d + a.d * b.v); }
__host__ __device__ inline LJ operator/(const LJ& f, const LJ& g) {
    const double gi = 1.0 / g.v;
    const double q = f.v * gi;
    return LJ(q, (f.d - q * g.d) * gi);
}
__host__ __device__ inline bool operator>(const LJ& a, const LJ& b) { return a.v > b.v; }
__host__ __device__ inline bool operator<(const LJ& a, const LJ& b) { return a.v < b.v; }

__host__ __device__ inline LJ dsqrt(const LJ& f) { const double t = sqrt(f.v); return LJ(t, f.d * (1.0 / (2.0 * t))); }
__host__ __device__ inline LJ dsin(const LJ& f) { return LJ(sin(f.v), cos(f.v) * f.d); }
__host__ __device__ inline LJ dcos(const LJ& f) { return LJ(cos(f.v), -sin(f.v) * f.d); }
__host__ __device__ inline LJ dasin(const LJ& f) { return LJ(asin(f.v), f.d * (1.0 / sqrt(1.0 - f.v * f.v))); }
__host__ __device__ inline LJ datan2(const LJ& g, const LJ& f) {
    const double t = 1.0 / (f.v * f.v + g.v * g.v);
    return LJ(atan2(g.v, f.v), t * (-g.v * f.d + f.v * g.d));
}
__host__ __device__ inline LJ dfloor(const LJ& f) { return LJ(floor(f.v), 0.0); }
__host__ __device__ inline double dsqrt(double x) { return sqrt(x); }
__host__ __device__ inline double dsin(double x) { return sin(x); }
__host__ __device__ inline double dcos(double x) { return cos(x); }
__host__ __device__ inline double dasin(double x) { return asin(x); }
__host__ __device__ inline double datan2(double y, double x) { return atan2(y, x); }
__host__ __device__ inline double dfloor(double x) { return floor(x); }
__host__ __device__ inline double val(double x) { return x; }
__host__ __device__ inline double val(const LJ& x) { return x.v; }

// N derivative directions per lane.  The IMU / wheel / ground roles carry three (the 9 non-linear directions of an IMU or wheel block
// on 3 lanes, the 6 of a ground block on 2): the value part — every sqrt / sin / cos / atan2 / division of the chain — is then
// evaluated once per 3 directions instead of once per direction, a wave holds 21 blocks instead of 6, and the three derivative
// chains of a lane are independent instructions that hide each other's fp64 latency.
template <int N> struct LJN {
    double v, d[N];
    __host__ __device__ LJN() : v(0.0) { for (int k = 0; k < N; ++k) d[k] = 0.0; }
    __host__ __device__ LJN(double c) : v(c) { for (int k = 0; k < N; ++k) d[k] = 0.0; }  // NOLINT(implicit)
};
template <int N> __host__ __device__ inline LJN<N> operator+(const LJN<N>& a, const LJN<N>& b) { LJN<N> r; r.v = a.v + b.v; for (int k = 0; k < N; ++k) r.d[k] = a.d[k] + b.d[k]; return r; }
template <int N> __host__ __device__ inline LJN<N> operator-(const LJN<N>& a, const LJN<N>& b) { LJN<N> r; r.v = a.v - b.v; for (int k = 0; k < N; ++k) r.d[k] = a.d[k] - b.d[k]; return r; }
template <int N> __host__ __device__ inline LJN<N> operator-(const LJN<N>& a) { LJN<N> r; r.v = -a.v; for (int k = 0; k < N; ++k) r.d[k] = -a.d[k]; return r; }
template <int N> __host__ __device__ inline LJN<N> operator*(const LJN<N>& a, const LJN<N>& b) {
    LJN<N> r; r.v = a.v * b.v;
    for (int k = 0; k < N; ++k) r.d[k] = a.v * b.d[k] + a.d[k] * b.v;
    return r;
}
template <int N> __host__ __device__ inline LJN<N> operator/(const LJN<N>& f, const LJN<N>& g) {
    const double gi = 1.0 / g.v;
    LJN<N> r; r.v = f.v * gi;
    for (int k = 0; k < N; ++k) r.d[k] = (f.d[k] - r.v * g.d[k]) * gi;
    return r;
}
template <int N> __host__ __device__ inline LJN<N> operator*(const LJN<N>& a, double c) { LJN<N> r; r.v = a.v * c; for (int k = 0; k < N; ++k) r.d[k] = a.d[k] * c; return r; }
template <int N> __host__ __device__ inline LJN<N> operator*(double c, const LJN<N>& a) { return a * c; }
template <int N> __host__ __device__ inline bool operator>(const LJN<N>& a, const LJN<N>& b) { return a.v > b.v; }
template <int N> __host__ __device__ inline bool operator<(const LJN<N>& a, const LJN<N>& b) { return a.v < b.v; }
template <int N> __host__ __device__ inline LJN<N> chain(double value, double slope, const LJN<N>& f) {   // g(f): value g(f.v), slope g'(f.v)
    LJN<N> r; r.v = value;
    for (int k = 0; k < N; ++k) r.d[k] = slope * f.d[k];
    return r;
}
template <int N> __host__ __device__ inline LJN<N> dsqrt(const LJN<N>& f) { const double t = sqrt(f.v); return chain(t, 1.0 / (2.0 * t), f); }
template <int N> __host__ __device__ inline LJN<N> dsin(const LJN<N>& f) { return chain(sin(f.v), cos(f.v), f); }
template <int N> __host__ __device__ inline LJN<N> dcos(const LJN<N>& f) { return chain(cos(f.v), -sin(f.v), f); }
template <int N> __host__ __device__ inline LJN<N> dasin(const LJN<N>& f) { return chain(asin(f.v), 1.0 / sqrt(1.0 - f.v * f.v), f); }
template <int N> __host__ __device__ inline LJN<N> datan2(const LJN<N>& g, const LJN<N>& f) {
    const double t = 1.0 / (f.v * f.v + g.v * g.v);
    LJN<N> r; r.v = atan2(g.v, f.v);
    for (int k = 0; k < N; ++k) r.d[k] = t * (-g.v * f.d[k] + f.v * g.d[k]);
    return r;
}
template <int N> __host__ __device__ inline LJN<N> dfloor(const LJN<N>& f) { return LJN<N>(floor(f.v)); }
template <int N> __host__ __device__ inline double val(const LJN<N>& x) { return x.v; }
// seed: value x, derivative 1 in slot k when `on` (the lane that owns this group of directions), else a constant
template <int N> __host__ __device__ inline LJN<N> seed(double x, int k, bool on) { LJN<N> r(x); if (on) r.d[k] = 1.0; return r; }

template <class T> struct V3 {
    T x, y, z;
    __host__ __device__ V3() : x(0.0), y(0.0), z(0.0) {}
    __host__ __device__ V3(const T& a, const T& b, const T& c) : x(a), y(b), z(c) {}
};
template <class T> __host__ __device__ inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return V3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <class T> __host__ __device__ inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return V3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <class T> __host__ __device__ inline V3<T> operator-(const V3<T>& a) { return V3<T>(-a.x, -a.y, -a.z); }
template <class T> __host__ __device__ inline V3<T> operator*(const V3<T>& a, const T& s) { return V3<T>(a.x * s, a.y * s, a.z * s); }
template <class T> __host__ __device__ inline V3<T> operator/(const V3<T>& a, const T& s) { return V3<T>(a.x / s, a.y / s, a.z / s); }
template <class T> __host__ __device__ inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> __host__ __device__ inline V3<T> cross(const V3<T>& a, const V3<T>& b) {
    return V3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <class T> __host__ __device__ inline T norm(const V3<T>& a) { return dsqrt(dot(a, a)); }
template <class T> __host__ __device__ inline V3<T> normalized(const V3<T>& a) {
    T z = dot(a, a);
    if (z > T(0.0)) return a / dsqrt(z);
    return a;
}

template <class T> __host__ __device__ inline V3<T> cast_v3(const double* p) { return V3<T>(T(p[0]), T(p[1]), T(p[2])); }

template <class T> struct M3 {
    T m[9];  // row-major
    __host__ __device__ T& operator()(int i, int j) { return m[i * 3 + j]; }
    __host__ __device__ const T& operator()(int i, int j) const { return m[i * 3 + j]; }
};
template <class T> __host__ __device__ inline M3<T> mul(const M3<T>& a, const M3<T>& b) {
    M3<T> r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
    return r;
}
template <class T> __host__ __device__ inline V3<T> mul(const M3<T>& a, const V3<T>& v) {
    return V3<T>(a(0, 0) * v.x + a(0, 1) * v.y + a(0, 2) * v.z, a(1, 0) * v.x + a(1, 1) * v.y + a(1, 2) * v.z,
                 a(2, 0) * v.x + a(2, 1) * v.y + a(2, 2) * v.z);
}
template <class T> __host__ __device__ inline V3<T> mulT(const M3<T>& a, const V3<T>& v) {  // a^T v
    return V3<T>(a(0, 0) * v.x + a(1, 0) * v.y + a(2, 0) * v.z, a(0, 1) * v.x + a(1, 1) * v.y + a(2, 1) * v.z,
                 a(0, 2) * v.x + a(1, 2) * v.y + a(2, 2) * v.z);
}
template <class T> __host__ __device__ inline M3<T> transpose(const M3<T>& a) {
    M3<T> r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r(i, j) = a(j, i);
    return r;
}
template <class T> __host__ __device__ inline M3<T> cast_m3(const double* rm9) {
    M3<T> r;
#pragma unroll
    for (int k = 0; k < 9; ++k) r.m[k] = T(rm9[k]);
    return r;
}

// rotation vector -> rotation matrix : toRotationMatrix(AngleAxisToQuaternion(a))   (common.h:137-146)
template <class T> __host__ __device__ inline M3<T> exp_so3(const V3<T>& a) {
    T qw, qx, qy, qz;
    const T theta_squared = a.x * a.x + a.y * a.y + a.z * a.z;
    if (theta_squared > T(0.0)) {
        const T theta = dsqrt(theta_squared);
        const T half_theta = theta * T(0.5);
        const T k = dsin(half_theta) / theta;
        qw = dcos(half_theta);
        qx = a.x * k; qy = a.y * k; qz = a.z * k;
    } else {
        const T k(0.5);
        qw = T(1.0);
        qx = a.x * k; qy = a.y * k; qz = a.z * k;
    }
    const T tx = T(2.0) * qx, ty = T(2.0) * qy, tz = T(2.0) * qz;
    const T twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const T txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const T tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    M3<T> r;
    r(0, 0) = T(1.0) - (tyy + tzz); r(0, 1) = txy - twz;            r(0, 2) = txz + twy;
    r(1, 0) = txy + twz;            r(1, 1) = T(1.0) - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy;            r(2, 1) = tyz + twx;            r(2, 2) = T(1.0) - (txx + tyy);
    return r;
}
// common.h:121-135
template <class T> __host__ __device__ inline V3<T> normalize_so3(const V3<T>& so3) {
    T angle = norm(so3);
    const T two_pi(6.283185307179586476925286766559);
    const T pi(3.141592653589793238462643383279);
    if (angle > pi) {
        T normalize_angle = angle - two_pi * dfloor((angle + pi) / two_pi);
        return (so3 / angle) * normalize_angle;
    }
    return so3;
}
// rotation matrix -> rotation vector : Quaternion(R).normalize() -> QuaternionToAngleAxis -> normalize_so3  (common.h:148-163)
template <class T> __host__ __device__ inline V3<T> log_SO3(const M3<T>& mat) {
    T c[4];  // x y z w
    T t = mat(0, 0) + mat(1, 1) + mat(2, 2);
    if (t > T(0.0)) {
        t = dsqrt(t + T(1.0));
        c[3] = T(0.5) * t;
        t = T(0.5) / t;
        c[0] = (mat(2, 1) - mat(1, 2)) * t;
        c[1] = (mat(0, 2) - mat(2, 0)) * t;
        c[2] = (mat(1, 0) - mat(0, 1)) * t;
    } else {
        int i = 0;
        if (mat(1, 1) > mat(0, 0)) i = 1;
        if (mat(2, 2) > (i == 1 ? mat(1, 1) : mat(0, 0))) i = 2;
        // the permuted entries are picked with selects on compile-time indices, so nothing is indexed dynamically
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        // scalar copies first: selects between array elements would otherwise be folded into a dynamically indexed load
        const T e0 = mat.m[0], e1 = mat.m[1], e2 = mat.m[2], e3 = mat.m[3], e4 = mat.m[4], e5 = mat.m[5], e6 = mat.m[6], e7 = mat.m[7], e8 = mat.m[8];
        const T d0 = e0, d1 = e4, d2 = e8;
        const T mii = i == 0 ? d0 : (i == 1 ? d1 : d2);
        const T mjj = j == 0 ? d0 : (j == 1 ? d1 : d2);
        const T mkk = k == 0 ? d0 : (k == 1 ? d1 : d2);
        // (k,j),(j,k): i=0 -> (2,1),(1,2); i=1 -> (0,2),(2,0); i=2 -> (1,0),(0,1)
        const T mkj = i == 0 ? e7 : (i == 1 ? e2 : e3);
        const T mjk = i == 0 ? e5 : (i == 1 ? e6 : e1);
        // (j,i),(i,j): i=0 -> (1,0),(0,1); i=1 -> (2,1),(1,2); i=2 -> (0,2),(2,0)
        const T mji = i == 0 ? e3 : (i == 1 ? e7 : e2);
        const T mij = i == 0 ? e1 : (i == 1 ? e5 : e6);
        // (k,i),(i,k): i=0 -> (2,0),(0,2); i=1 -> (0,1),(1,0); i=2 -> (1,2),(2,1)
        const T mki = i == 0 ? e6 : (i == 1 ? e1 : e5);
        const T mik = i == 0 ? e2 : (i == 1 ? e3 : e7);
        t = dsqrt(mii - mjj - mkk + T(1.0));
        const T ci = T(0.5) * t;
        t = T(0.5) / t;
        const T cw = (mkj - mjk) * t;
        const T cj = (mji + mij) * t;
        const T ck = (mki + mik) * t;
        c[3] = cw;
        c[0] = (i == 0) ? ci : ((j == 0) ? cj : ck);
        c[1] = (i == 1) ? ci : ((j == 1) ? cj : ck);
        c[2] = (i == 2) ? ci : ((j == 2) ? cj : ck);
    }
    // Quaternion::normalize()
    {
        T z = c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + c[3] * c[3];
        if (z > T(0.0)) {
            T s = dsqrt(z);
            c[0] = c[0] / s; c[1] = c[1] / s; c[2] = c[2] / s; c[3] = c[3] / s;
        }
    }
    // QuaternionToAngleAxis
    V3<T> aa;
    const T sin_squared_theta = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    if (sin_squared_theta > T(0.0)) {
        const T sin_theta = dsqrt(sin_squared_theta);
        const T cos_theta = c[3];
        const T two_theta = T(2.0) * ((cos_theta < T(0.0)) ? datan2(-sin_theta, -cos_theta) : datan2(sin_theta, cos_theta));
        const T k = two_theta / sin_theta;
        aa = V3<T>(c[0] * k, c[1] * k, c[2] * k);
    } else {
        const T k(2.0);
        aa = V3<T>(c[0] * k, c[1] * k, c[2] * k);
    }
    return normalize_so3(aa);
}

// Isometry as (R, t); products as Eigen's Transform<Isometry> does them.
template <class T> struct Iso { M3<T> R; V3<T> t; };
template <class T> __host__ __device__ inline Iso<T> mul(const Iso<T>& a, const Iso<T>& b) {
    Iso<T> r; r.R = mul(a.R, b.R); r.t = mul(a.R, b.t) + a.t; return r;
}
template <class T> __host__ __device__ inline Iso<T> inverse(const Iso<T>& a) {
    Iso<T> r; r.R = transpose(a.R); r.t = -mul(r.R, a.t); return r;
}
template <class T> __host__ __device__ inline Iso<T> make_tf(const V3<T>& p, const V3<T>& so3) {
    Iso<T> r; r.R = exp_so3(so3); r.t = p; return r;
}
template <class T> __host__ __device__ inline Iso<T> cast_iso(const double* R9, const double* t3) {
    Iso<T> r; r.R = cast_m3<T>(R9); r.t = V3<T>(T(t3[0]), T(t3[1]), T(t3[2])); return r;
}

}  // namespace liw
