// liw_preint.cpp — host pre-integrators of libliw_window.so (per-message, sequential by nature).
//
// Native replacements of the reference's accumulators
//   imu_preintegraption          src/factor/imu_preintegraption.h:105-208   (+ imu_noise :8-44)
//   wheel_odom_preintegration    src/factor/wheel_odom_preintegration.h:44-152 (+ wheel_noise :6-23)
// Same public operations (reset / add measure / update_only_t / get result).  Unlike the reference, which
// forms dense 15x15 F and 15x12 G and multiplies them out (three dense 15^3 products per IMU sample), the
// propagation here uses the block structure of F = I + dt*[..] directly: only the alpha<-beta, beta<-{gamma,ba},
// gamma<-{gamma,bw} couplings are non-trivial, so J <- F J and P <- F P F^T + (G dt) Q (G dt)^T cost a few
// 3x15 row updates.  Reference quirks are kept (SURVEY Appendix C 5,6): F(gamma,gamma) is built from
// hat_gyro - last_ba, the previous sample drives the whole Euler step, P0 = 1e-5 I.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "../../include/liw_window.h"
#include "liw_dual.hpp"

using liw::M3;
using liw::V3;

namespace {

typedef double Mat15[15][15];

inline M3<double> skew(const V3<double>& v) {
    M3<double> r;
    for (int k = 0; k < 9; ++k) r.m[k] = 0.0;
    r(0, 1) = -v.z; r(1, 0) = v.z; r(0, 2) = v.y; r(2, 0) = -v.y; r(1, 2) = -v.x; r(2, 1) = v.x;
    return r;
}

// symmetric positive definite inverse + Cholesky for the 15x15 / 3x3 information square roots.
// sqrt_inverse_P = LLT(P^-1).matrixL().transpose()  (imu_preintegraption.h:149): upper U with U^T U = P^-1.
void lu_inverse(int n, const double* A, double* Ainv) {
    double lu[15 * 15];
    int perm[15];
    std::memcpy(lu, A, sizeof(double) * n * n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = std::fabs(lu[k * n + k]);
        for (int i = k + 1; i < n; ++i) if (std::fabs(lu[i * n + k]) > best) { best = std::fabs(lu[i * n + k]); piv = i; }
        if (piv != k) { for (int j = 0; j < n; ++j) std::swap(lu[k * n + j], lu[piv * n + j]); std::swap(perm[k], perm[piv]); }
        const double inv = 1.0 / lu[k * n + k];
        for (int i = k + 1; i < n; ++i) {
            const double f = lu[i * n + k] * inv;
            lu[i * n + k] = f;
            for (int j = k + 1; j < n; ++j) lu[i * n + j] -= f * lu[k * n + j];
        }
    }
    for (int i = 0; i < n * n; ++i) Ainv[i] = 0.0;
    for (int i = 0; i < n; ++i) Ainv[i * n + perm[i]] = 1.0;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < i; ++k) { const double f = lu[i * n + k]; for (int j = 0; j < n; ++j) Ainv[i * n + j] -= f * Ainv[k * n + j]; }
    for (int i = n - 1; i >= 0; --i) {
        for (int k = i + 1; k < n; ++k) { const double f = lu[i * n + k]; for (int j = 0; j < n; ++j) Ainv[i * n + j] -= f * Ainv[k * n + j]; }
        const double inv = 1.0 / lu[i * n + i];
        for (int j = 0; j < n; ++j) Ainv[i * n + j] *= inv;
    }
}
void chol_upper_of(int n, const double* A, double* U) {   // A = L L^T, U = L^T
    double L[15 * 15];
    for (int i = 0; i < n * n; ++i) L[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = A[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        const double d = std::sqrt(s);
        L[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double t = A[i * n + j];
            for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = t / d;
        }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) U[i * n + j] = L[j * n + i];
}

}  // namespace

struct liw_imu_preint {
    double q_na[3], q_nw[3], q_nba[3], q_nbw[3];   // diag of Q (sigma^2)
    double last_add_imu_time;
    double Dt;
    double X[15];
    Mat15 J, P;
    double last_acc[3], last_gyro[3];

    void update(double dt) {
        const V3<double> a_unb(last_acc[0] - X[9], last_acc[1] - X[10], last_acc[2] - X[11]);
        const V3<double> w_unb(last_gyro[0] - X[12], last_gyro[1] - X[13], last_gyro[2] - X[14]);
        const V3<double> gamma(X[6], X[7], X[8]);
        const M3<double> Rz = liw::exp_so3(gamma);
        const V3<double> Ra = liw::mul(Rz, a_unb);
        // state (imu_preintegraption.h:183-185)
        const double b0[3] = {X[3], X[4], X[5]};
        const double ra[3] = {Ra.x, Ra.y, Ra.z};
        for (int k = 0; k < 3; ++k) {
            X[k] = X[k] + b0[k] * dt + 0.5 * ra[k] * dt * dt;
            X[3 + k] = b0[k] + ra[k] * dt;
        }
        const V3<double> g2 = liw::log_SO3(liw::mul(Rz, liw::exp_so3(V3<double>(w_unb.x * dt, w_unb.y * dt, w_unb.z * dt))));
        X[6] = g2.x; X[7] = g2.y; X[8] = g2.z;
        // F = I + dt * Fc, Fc blocks (:189-193):  (alpha,beta)=I  (beta,gamma)=-Rz [a]x  (beta,ba)=-Rz
        //                                         (gamma,gamma)=-[hat_gyro - last_ba]x (sic)  (gamma,bw)=-I
        const M3<double> Rax = liw::mul(Rz, skew(a_unb));
        const M3<double> Wx = skew(V3<double>(last_gyro[0] - X[9], last_gyro[1] - X[10], last_gyro[2] - X[11]));
        double Fbg[3][3], Fbb[3][3], Fgg[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                Fbg[i][j] = -Rax(i, j) * dt;
                Fbb[i][j] = -Rz(i, j) * dt;
                Fgg[i][j] = (i == j ? 1.0 : 0.0) - Wx(i, j) * dt;
            }
        auto left_mul_F = [&](Mat15& M) {   // M <- F M using old rows
            double nb[3][15], ng[3][15], na[3][15];
            for (int i = 0; i < 3; ++i)
                for (int c = 0; c < 15; ++c) {
                    na[i][c] = M[i][c] + dt * M[3 + i][c];
                    double sb = M[3 + i][c], sg = -dt * M[12 + i][c];
                    for (int k = 0; k < 3; ++k) {
                        sb += Fbg[i][k] * M[6 + k][c] + Fbb[i][k] * M[9 + k][c];
                        sg += Fgg[i][k] * M[6 + k][c];
                    }
                    nb[i][c] = sb; ng[i][c] = sg;
                }
            for (int i = 0; i < 3; ++i)
                for (int c = 0; c < 15; ++c) { M[i][c] = na[i][c]; M[3 + i][c] = nb[i][c]; M[6 + i][c] = ng[i][c]; }
        };
        auto right_mul_Ft = [&](Mat15& M) {  // M <- M F^T using old columns
            for (int r = 0; r < 15; ++r) {
                double na[3], nb[3], ng[3];
                for (int i = 0; i < 3; ++i) {
                    na[i] = M[r][i] + dt * M[r][3 + i];
                    double sb = M[r][3 + i], sg = -dt * M[r][12 + i];
                    for (int k = 0; k < 3; ++k) {
                        sb += Fbg[i][k] * M[r][6 + k] + Fbb[i][k] * M[r][9 + k];
                        sg += Fgg[i][k] * M[r][6 + k];
                    }
                    nb[i] = sb; ng[i] = sg;
                }
                for (int i = 0; i < 3; ++i) { M[r][i] = na[i]; M[r][3 + i] = nb[i]; M[r][6 + i] = ng[i]; }
            }
        };
        left_mul_F(J);
        left_mul_F(P);
        right_mul_Ft(P);
        // (G dt) Q (G dt)^T : G blocks (:196-199): (beta,na)=-Rz (gamma,nw)=-I (ba,nba)=I (bw,nbw)=I
        const double dt2 = dt * dt;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0.0;
                for (int k = 0; k < 3; ++k) s += Rz(i, k) * q_na[k] * Rz(j, k);
                P[3 + i][3 + j] += s * dt2;
            }
        for (int k = 0; k < 3; ++k) {
            P[6 + k][6 + k] += q_nw[k] * dt2;
            P[9 + k][9 + k] += q_nba[k] * dt2;
            P[12 + k][12 + k] += q_nbw[k] * dt2;
        }
        Dt += dt;
    }
};

struct liw_wheel_preint {
    double wheel_cov[3];
    liw::Iso<double> delta_Tij, last_pose;
    double last_update_time, last_add_time, Dt;
    double v[3], omega[3];

    void update_by_v(double dt) {
        if (dt <= 0 || dt >= 10) return;   // wheel_odom_preintegration.h:143-147
        Dt += dt;
        liw::Iso<double> dT = liw::make_tf(V3<double>(v[0] * dt, v[1] * dt, v[2] * dt), V3<double>(omega[0] * dt, omega[1] * dt, omega[2] * dt));
        delta_Tij = liw::mul(delta_Tij, dT);
    }
};

static liw::Iso<double> iso_identity() {
    liw::Iso<double> T;
    for (int k = 0; k < 9; ++k) T.R.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
    T.t = V3<double>(0.0, 0.0, 0.0);
    return T;
}

extern "C" {

liw_imu_preint* liw_imu_preint_create(const liw_params* prm) {
    if (!prm) return nullptr;
    liw_imu_preint* p = new liw_imu_preint();
    for (int k = 0; k < 3; ++k) {
        p->q_na[k] = prm->imu_noise_acc_sigma[k] * prm->imu_noise_acc_sigma[k];
        p->q_nw[k] = prm->imu_noise_gyro_sigma[k] * prm->imu_noise_gyro_sigma[k];
        p->q_nba[k] = prm->imu_bias_acc_sigma[k] * prm->imu_bias_acc_sigma[k];
        p->q_nbw[k] = prm->imu_bias_gyro_sigma[k] * prm->imu_bias_gyro_sigma[k];
    }
    const double z[3] = {0, 0, 0};
    liw_imu_preint_reset(p, -1.0, z, z);
    return p;
}
void liw_imu_preint_destroy(liw_imu_preint* p) { delete p; }
void liw_imu_preint_reset(liw_imu_preint* p, double time, const double* ba, const double* bw) {
    for (int i = 0; i < 15; ++i) {
        p->X[i] = 0.0;
        for (int j = 0; j < 15; ++j) { p->J[i][j] = i == j ? 1.0 : 0.0; p->P[i][j] = i == j ? 0.00001 : 0.0; }
    }
    for (int k = 0; k < 3; ++k) { p->X[9 + k] = ba[k]; p->X[12 + k] = bw[k]; }
    p->last_add_imu_time = time;
    p->Dt = 0.0;
}
int liw_imu_preint_add(liw_imu_preint* p, double t, const double* acc, const double* gyro) {
    int integrated = 0;
    if (p->last_add_imu_time != -1) { p->update(t - p->last_add_imu_time); integrated = 1; }
    for (int k = 0; k < 3; ++k) { p->last_acc[k] = acc[k]; p->last_gyro[k] = gyro[k]; }
    p->last_add_imu_time = t;
    return integrated;
}
void liw_imu_preint_update_only_t(liw_imu_preint* p, double time) {
    if (p->last_add_imu_time == -1) return;
    p->update(time - p->last_add_imu_time);
    p->last_add_imu_time = time;
}
double liw_imu_preint_Dt(const liw_imu_preint* p) { return p->Dt; }
void liw_imu_preint_result(const liw_imu_preint* p, double* X15, double* J225, double* sq225, double* Dt) {
    double Pm[225], Pinv[225];
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) Pm[i * 15 + j] = p->P[i][j];
    lu_inverse(15, Pm, Pinv);
    chol_upper_of(15, Pinv, sq225);
    std::memcpy(X15, p->X, sizeof(p->X));
    for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) J225[i * 15 + j] = p->J[i][j];
    if (Dt) *Dt = p->Dt;
}

liw_wheel_preint* liw_wheel_preint_create(const liw_params* prm) {
    if (!prm) return nullptr;
    liw_wheel_preint* p = new liw_wheel_preint();
    for (int k = 0; k < 3; ++k) { p->wheel_cov[k] = prm->wheel_sigma[k] * prm->wheel_sigma[k]; p->v[k] = p->omega[k] = 0.0; }
    p->last_add_time = 0.0;
    p->last_pose = iso_identity();
    liw_wheel_preint_reset(p, -1.0);
    return p;
}
void liw_wheel_preint_destroy(liw_wheel_preint* p) { delete p; }
void liw_wheel_preint_reset(liw_wheel_preint* p, double time) {
    p->last_update_time = time;
    p->delta_Tij = iso_identity();
    p->Dt = 0.0;
}
int liw_wheel_preint_add(liw_wheel_preint* p, double t, const double* R9, const double* t3) {
    liw::Iso<double> pose = liw::cast_iso<double>(R9, t3);
    if (p->last_update_time < 0) {   // wheel_odom_preintegration.h:65-75
        p->last_pose = pose;
        p->last_add_time = t;
        p->last_update_time = t;
        p->delta_Tij = iso_identity();
        for (int k = 0; k < 3; ++k) p->v[k] = p->omega[k] = 0.0;
        return 0;
    }
    const double dt = t - p->last_add_time;
    liw::Iso<double> rel = liw::mul(liw::inverse(p->last_pose), pose);
    const V3<double> dth = liw::log_SO3(rel.R);
    if (dt < 0.05) return 0;
    p->v[0] = rel.t.x / dt; p->v[1] = rel.t.y / dt; p->v[2] = rel.t.z / dt;
    p->omega[0] = dth.x / dt; p->omega[1] = dth.y / dt; p->omega[2] = dth.z / dt;
    p->update_by_v(t - p->last_update_time);
    p->last_pose = pose;
    p->last_add_time = t;
    p->last_update_time = t;
    return 1;
}
void liw_wheel_preint_update_only_t(liw_wheel_preint* p, double time) {
    if (p->last_update_time < 0) return;
    p->update_by_v(time - p->last_update_time);
    p->last_update_time = time;
}
void liw_wheel_preint_result(const liw_wheel_preint* p, double* T12, double* sq9, double* Dt) {
    const V3<double> dq = liw::log_SO3(p->delta_Tij.R);
    const double len_norm = std::max(p->delta_Tij.t.x * p->delta_Tij.t.x + p->delta_Tij.t.y * p->delta_Tij.t.y + p->delta_Tij.t.z * p->delta_Tij.t.z, 0.005 * 0.005);
    const double yaw_norm = std::max(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z, 0.005 * 0.005);
    const double kd[3] = {len_norm, len_norm, yaw_norm};
    // cov is diagonal: LLT(cov^-1)^T = diag(1/sqrt(cov))
    for (int k = 0; k < 9; ++k) sq9[k] = 0.0;
    for (int k = 0; k < 3; ++k) sq9[k * 4] = std::sqrt(1.0 / (p->wheel_cov[k] * kd[k]));
    for (int k = 0; k < 9; ++k) T12[k] = p->delta_Tij.R.m[k];
    T12[9] = p->delta_Tij.t.x; T12[10] = p->delta_Tij.t.y; T12[11] = p->delta_Tij.t.z;
    if (Dt) *Dt = p->Dt;
}

}  // extern "C"
