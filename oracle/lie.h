// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Restates, for T = double or Jet<N>:
//   * the fixed-size Eigen pieces the reference's functors use (Vector3, Matrix3, Isometry
//     product / inverse / point transform, Quaternion(Matrix3), Quaternion::toRotationMatrix,
//     normalized()) — Eigen 3.3.x is a third-party dependency not present in /root/reference
//     (CMakeLists.txt:25 hard-codes /usr/include/eigen3; ros:noetic => 3.3.7);
//   * ceres::AngleAxisToQuaternion / QuaternionToAngleAxis (ceres/rotation.h, Ceres 1.14.x,
//     not vendored) as called from src/utilies/common.h:143,157;
//   * the reference's own lie:: helpers, src/utilies/common.h:119-196, and
//     e_laser::dis_from_line, src/utilies/common.h:86-95.
#pragma once
#include "jet.h"

namespace oracle {

template <typename T> struct Vec3 {
    T x[3];
    Vec3() { x[0] = T(0.0); x[1] = T(0.0); x[2] = T(0.0); }
    Vec3(const T& a, const T& b, const T& c) { x[0] = a; x[1] = b; x[2] = c; }
    T& operator()(int i) { return x[i]; }
    const T& operator()(int i) const { return x[i]; }
};
template <typename T> inline Vec3<T> operator+(const Vec3<T>& a, const Vec3<T>& b) { return {a(0) + b(0), a(1) + b(1), a(2) + b(2)}; }
template <typename T> inline Vec3<T> operator-(const Vec3<T>& a, const Vec3<T>& b) { return {a(0) - b(0), a(1) - b(1), a(2) - b(2)}; }
template <typename T> inline Vec3<T> operator-(const Vec3<T>& a) { return {-a(0), -a(1), -a(2)}; }
template <typename T> inline Vec3<T> operator*(const Vec3<T>& a, const T& s) { return {a(0) * s, a(1) * s, a(2) * s}; }
template <typename T> inline Vec3<T> operator*(const T& s, const Vec3<T>& a) { return {s * a(0), s * a(1), s * a(2)}; }
template <typename T> inline Vec3<T> operator/(const Vec3<T>& a, const T& s) { return {a(0) / s, a(1) / s, a(2) / s}; }
template <typename T> inline T dot(const Vec3<T>& a, const Vec3<T>& b) { return a(0) * b(0) + a(1) * b(1) + a(2) * b(2); }
template <typename T> inline Vec3<T> cross(const Vec3<T>& a, const Vec3<T>& b) {
    return {a(1) * b(2) - a(2) * b(1), a(2) * b(0) - a(0) * b(2), a(0) * b(1) - a(1) * b(0)};
}
template <typename T> inline T squared_norm(const Vec3<T>& a) { return dot(a, a); }
// Eigen: norm() = sqrt(squaredNorm())
template <typename T> inline T norm(const Vec3<T>& a) { return sqrt(squared_norm(a)); }
// Eigen 3.3 MatrixBase::normalized(): z = squaredNorm(); z > 0 ? x / sqrt(z) : x
template <typename T> inline Vec3<T> normalized(const Vec3<T>& a) {
    T z = squared_norm(a);
    if (z > T(0.0)) return a / sqrt(z);
    return a;
}
template <typename T, typename S> inline Vec3<T> cast3(const Vec3<S>& a) { return {T(a(0)), T(a(1)), T(a(2))}; }

template <typename T> struct Mat3 {
    T m[3][3];
    Mat3() { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m[i][j] = T(0.0); }
    T& operator()(int i, int j) { return m[i][j]; }
    const T& operator()(int i, int j) const { return m[i][j]; }
    static Mat3 identity() { Mat3 r; r(0, 0) = T(1.0); r(1, 1) = T(1.0); r(2, 2) = T(1.0); return r; }
    Mat3 transpose() const { Mat3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = m[j][i]; return r; }
    Vec3<T> col(int j) const { return {m[0][j], m[1][j], m[2][j]}; }
};
template <typename T> inline Mat3<T> operator*(const Mat3<T>& a, const Mat3<T>& b) {
    Mat3<T> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r(i, j) = a(i, 0) * b(0, j) + a(i, 1) * b(1, j) + a(i, 2) * b(2, j);
    return r;
}
template <typename T> inline Vec3<T> operator*(const Mat3<T>& a, const Vec3<T>& v) {
    return {a(0, 0) * v(0) + a(0, 1) * v(1) + a(0, 2) * v(2), a(1, 0) * v(0) + a(1, 1) * v(1) + a(1, 2) * v(2),
            a(2, 0) * v(0) + a(2, 1) * v(1) + a(2, 2) * v(2)};
}
template <typename T> inline Mat3<T> operator-(const Mat3<T>& a) {
    Mat3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = -a(i, j); return r;
}
template <typename T, typename S> inline Mat3<T> cast33(const Mat3<S>& a) {
    Mat3<T> r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = T(a(i, j)); return r;
}
// convert::cross_matrix, src/utilies/common.h:16-29
template <typename T> inline Mat3<T> cross_matrix(const Vec3<T>& v) {
    Mat3<T> r;
    r(0, 1) = -v(2); r(1, 0) = v(2);
    r(0, 2) = v(1);  r(2, 0) = -v(1);
    r(1, 2) = -v(0); r(2, 1) = v(0);
    return r;
}

// Eigen::Transform<T,3,Isometry>: linear() R, translation() t
template <typename T> struct Iso3 {
    Mat3<T> R;
    Vec3<T> t;
    Iso3() : R(Mat3<T>::identity()), t() {}
    Iso3(const Mat3<T>& R_, const Vec3<T>& t_) : R(R_), t(t_) {}
    // Transform<Isometry>::inverse(): (R^T, -R^T t)
    Iso3 inverse() const { Mat3<T> Rt = R.transpose(); return Iso3(Rt, -(Rt * t)); }
};
// Transform * Transform (affine compact): linear = L*R ; translation = L.linear*R.translation + L.translation
template <typename T> inline Iso3<T> operator*(const Iso3<T>& a, const Iso3<T>& b) { return Iso3<T>(a.R * b.R, a.R * b.t + a.t); }
template <typename T> inline Vec3<T> operator*(const Iso3<T>& a, const Vec3<T>& p) { return a.R * p + a.t; }
template <typename T, typename S> inline Iso3<T> cast_iso(const Iso3<S>& a) { return Iso3<T>(cast33<T>(a.R), cast3<T>(a.t)); }

// Quaternion stored (w,x,y,z)
template <typename T> struct Quat { T w, x, y, z; };

// Eigen QuaternionBase::toRotationMatrix()
template <typename T> inline Mat3<T> quat_to_rotmat(const Quat<T>& q) {
    const T tx = T(2.0) * q.x, ty = T(2.0) * q.y, tz = T(2.0) * q.z;
    const T twx = tx * q.w, twy = ty * q.w, twz = tz * q.w;
    const T txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
    const T tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
    Mat3<T> r;
    r(0, 0) = T(1.0) - (tyy + tzz); r(0, 1) = txy - twz;            r(0, 2) = txz + twy;
    r(1, 0) = txy + twz;            r(1, 1) = T(1.0) - (txx + tzz); r(1, 2) = tyz - twx;
    r(2, 0) = txz - twy;            r(2, 1) = tyz + twx;            r(2, 2) = T(1.0) - (txx + tyy);
    return r;
}
// Eigen quaternionbase_assign_impl<Matrix3>::run (Quaternion(const Matrix3&))
template <typename T> inline Quat<T> rotmat_to_quat(const Mat3<T>& mat) {
    T c[4];  // x y z w, Eigen coeffs() order
    T t = mat(0, 0) + mat(1, 1) + mat(2, 2);
    if (t > T(0.0)) {
        t = sqrt(t + T(1.0));
        c[3] = T(0.5) * t;
        t = T(0.5) / t;
        c[0] = (mat(2, 1) - mat(1, 2)) * t;
        c[1] = (mat(0, 2) - mat(2, 0)) * t;
        c[2] = (mat(1, 0) - mat(0, 1)) * t;
    } else {
        int i = 0;
        if (mat(1, 1) > mat(0, 0)) i = 1;
        if (mat(2, 2) > mat(i, i)) i = 2;
        int j = (i + 1) % 3;
        int k = (j + 1) % 3;
        t = sqrt(mat(i, i) - mat(j, j) - mat(k, k) + T(1.0));
        c[i] = T(0.5) * t;
        t = T(0.5) / t;
        c[3] = (mat(k, j) - mat(j, k)) * t;
        c[j] = (mat(j, i) + mat(i, j)) * t;
        c[k] = (mat(k, i) + mat(i, k)) * t;
    }
    return Quat<T>{c[3], c[0], c[1], c[2]};
}
// Quaternion::normalize(): coeffs().normalize() (z>0 guarded division by sqrt(squaredNorm))
template <typename T> inline void quat_normalize(Quat<T>& q) {
    // Eigen sums the 4 coefficients in storage order x,y,z,w
    T z = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
    if (z > T(0.0)) {
        T s = sqrt(z);
        q.x = q.x / s; q.y = q.y / s; q.z = q.z / s; q.w = q.w / s;
    }
}

// ceres::AngleAxisToQuaternion (ceres/rotation.h)
template <typename T> inline void AngleAxisToQuaternion(const T* angle_axis, T* quaternion) {
    const T& a0 = angle_axis[0];
    const T& a1 = angle_axis[1];
    const T& a2 = angle_axis[2];
    const T theta_squared = a0 * a0 + a1 * a1 + a2 * a2;
    if (theta_squared > T(0.0)) {
        const T theta = sqrt(theta_squared);
        const T half_theta = theta * T(0.5);
        const T k = sin(half_theta) / theta;
        quaternion[0] = cos(half_theta);
        quaternion[1] = a0 * k;
        quaternion[2] = a1 * k;
        quaternion[3] = a2 * k;
    } else {
        // first-order Taylor so derivatives survive at the origin
        const T k(0.5);
        quaternion[0] = T(1.0);
        quaternion[1] = a0 * k;
        quaternion[2] = a1 * k;
        quaternion[3] = a2 * k;
    }
}
// ceres::QuaternionToAngleAxis (ceres/rotation.h)
template <typename T> inline void QuaternionToAngleAxis(const T* quaternion, T* angle_axis) {
    const T& q1 = quaternion[1];
    const T& q2 = quaternion[2];
    const T& q3 = quaternion[3];
    const T sin_squared_theta = q1 * q1 + q2 * q2 + q3 * q3;
    if (sin_squared_theta > T(0.0)) {
        const T sin_theta = sqrt(sin_squared_theta);
        const T& cos_theta = quaternion[0];
        const T two_theta =
            T(2.0) * ((cos_theta < T(0.0)) ? atan2(-sin_theta, -cos_theta) : atan2(sin_theta, cos_theta));
        const T k = two_theta / sin_theta;
        angle_axis[0] = q1 * k;
        angle_axis[1] = q2 * k;
        angle_axis[2] = q3 * k;
    } else {
        const T k(2.0);
        angle_axis[0] = q1 * k;
        angle_axis[1] = q2 * k;
        angle_axis[2] = q3 * k;
    }
}

namespace lie {
// src/utilies/common.h:121-135
template <typename T> inline void normalize_so3(Vec3<T>& so3) {
    T angle = norm(so3);
    T normalize_angle = angle;
    T two_pi(2.0 * M_PI);
    T pi(M_PI);
    if (angle > pi)
        normalize_angle -= two_pi * floor((angle + pi) / two_pi);
    else
        return;
    so3 = so3 / angle;
    so3 = so3 * normalize_angle;
}
// src/utilies/common.h:137-146
template <typename T> inline Mat3<T> exp_so3(const Vec3<T>& so3) {
    T angleAxis_[3] = {so3(0), so3(1), so3(2)};
    T q_[4];
    AngleAxisToQuaternion(angleAxis_, q_);
    Quat<T> q{q_[0], q_[1], q_[2], q_[3]};
    return quat_to_rotmat(q);
}
// src/utilies/common.h:148-163
template <typename T> inline Vec3<T> log_SO3(const Mat3<T>& SO3) {
    Quat<T> q = rotmat_to_quat(SO3);
    quat_normalize(q);
    T angleAxis_[3];
    T q_[4]{q.w, q.x, q.y, q.z};
    QuaternionToAngleAxis(q_, angleAxis_);
    Vec3<T> ret(angleAxis_[0], angleAxis_[1], angleAxis_[2]);
    normalize_so3<T>(ret);
    return ret;
}
// src/utilies/common.h:165-171
template <typename T> inline void log_SE3(const Iso3<T>& SE3, Vec3<T>& p, Vec3<T>& so3) {
    p = SE3.t;
    so3 = log_SO3<T>(SE3.R);
}
// src/utilies/common.h:173-181
template <typename T> inline Iso3<T> make_tf(const Vec3<T>& p, const Vec3<T>& so3) { return Iso3<T>(exp_so3(so3), p); }
// src/utilies/common.h:183-189 (used by the parameter loader, src/utilies/params.cpp:44-54)
template <typename T> inline void normalize_tf(Iso3<T>& SE3) { SE3.R = quat_to_rotmat(rotmat_to_quat(SE3.R)); }
}  // namespace lie

namespace e_laser {
// src/utilies/common.h:86-95 — unsigned distance, `line` normalised twice (kept)
template <typename T> inline T dis_from_line(const Vec3<T>& p, const Vec3<T>& p1, const Vec3<T>& p2) {
    Vec3<T> line = normalized(p2 - p1);
    Vec3<T> p2p = p - p2;
    return norm(p2p - dot(normalized(line), p2p) * line);
}
}  // namespace e_laser

}  // namespace oracle
