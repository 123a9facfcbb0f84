// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Restatement of the reference's pre-integrators:
//   imu_noise / imu_preintegraption         src/factor/imu_preintegraption.h:8-44, :105-208
//   wheel_noise / wheel_odom_preintegration src/factor/wheel_odom_preintegration.h:6-23, :44-152
// Dense 15x15 arithmetic exactly as written there (F, G built in full and multiplied in full),
// including the reference quirks SURVEY Appendix C lists: F(gamma,gamma) uses `hat_gyro -
// last_ba` (:192); the whole step uses the PREVIOUS sample (:179-185); P0 = 1e-5 I (:117).
#pragma once
#include "dense.h"
#include "factors.h"

namespace oracle {

struct imu_sample { double time_stamp; Vec3<double> acc, gyro; };
struct wheel_sample { double time_stamp; Iso3<double> pose; };

class imu_preintegraption {
public:
    double Dt;
    explicit imu_preintegraption(const params* prm_) : prm(prm_) {
        double z[3] = {0, 0, 0};
        reset_imu_measure(-1, z, z);
    }
    void reset_imu_measure(double time, const double* acc_bias, const double* gyr_bias) {
        J = DMat(15, 15); P = DMat(15, 15);
        for (int i = 0; i < 15; ++i) { J(i, i) = 1.0; P(i, i) = 0.00001; X[i] = 0.0; }
        for (int k = 0; k < 3; ++k) { X[ba_index + k] = acc_bias[k]; X[bw_index + k] = gyr_bias[k]; }
        last_add_imu_time = time;
        Dt = 0;
    }
    bool add_imu_measure(const imu_sample& data) {
        if (last_add_imu_time == -1) {
            last_info = data;
            last_add_imu_time = data.time_stamp;
            return false;
        }
        double dt = data.time_stamp - last_add_imu_time;
        update(dt);
        last_info = data;
        last_add_imu_time = data.time_stamp;
        return true;
    }
    void update_only_t(double time) {
        if (last_add_imu_time == -1) return;
        double dt = time - last_add_imu_time;
        update(dt);
        last_add_imu_time = time;
    }
    imu_preint_result get_preintegraption_result() const {
        // sqrt_inverse_P = LLT(P^-1).matrixL().transpose()
        DMat Pinv, L;
        lu_inverse(P, Pinv);
        llt_lower(Pinv, L);
        imu_preint_result r;
        for (int i = 0; i < 15; ++i) {
            r.X[i] = X[i];
            for (int j = 0; j < 15; ++j) { r.J[i][j] = J(i, j); r.sqrt_inverse_P[i][j] = L(j, i); }
        }
        r.Dt = Dt;
        return r;
    }

private:
    const params* prm;
    double last_add_imu_time;
    DMat J, P;
    double X[15];
    imu_sample last_info;

    void update(double dt) {
        Vec3<double> last_alpha(X[0], X[1], X[2]), last_beta(X[3], X[4], X[5]), last_gamma(X[6], X[7], X[8]);
        Vec3<double> last_ba(X[9], X[10], X[11]), last_bw(X[12], X[13], X[14]);
        Mat3<double> last_Rz = lie::exp_so3<double>(last_gamma);
        Vec3<double> hat_acc = last_info.acc, hat_gyro = last_info.gyro;

        Vec3<double> a_unb = hat_acc - last_ba;
        Vec3<double> Ra = last_Rz * a_unb;
        Vec3<double> n_alpha = last_alpha + last_beta * dt + (0.5 * Ra) * dt * dt;   // 0.5 * last_Rz * (..) * dt * dt
        Vec3<double> n_beta = last_beta + Ra * dt;
        Vec3<double> n_gamma = lie::log_SO3<double>(lie::exp_so3<double>(last_gamma) * lie::exp_so3<double>((hat_gyro - last_bw) * dt));
        for (int k = 0; k < 3; ++k) { X[k] = n_alpha(k); X[3 + k] = n_beta(k); X[6 + k] = n_gamma(k); }

        DMat F(15, 15);
        for (int k = 0; k < 3; ++k) F(alpha_index + k, beta_index + k) = 1.0;
        Mat3<double> Rax = (-last_Rz) * cross_matrix<double>(a_unb);
        Mat3<double> mR = -last_Rz;
        Mat3<double> mWx = -cross_matrix<double>(hat_gyro - last_ba);  // sic: last_ba, reference :192
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                F(beta_index + i, gamma_index + j) = Rax(i, j);
                F(beta_index + i, ba_index + j) = mR(i, j);
                F(gamma_index + i, gamma_index + j) = mWx(i, j);
            }
        for (int k = 0; k < 3; ++k) F(gamma_index + k, bw_index + k) = -1.0;

        DMat G(15, 12);
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) G(beta_index + i, 0 + j) = mR(i, j);
        for (int k = 0; k < 3; ++k) {
            G(gamma_index + k, 3 + k) = -1.0;
            G(ba_index + k, 6 + k) = 1.0;
            G(bw_index + k, 9 + k) = 1.0;
        }
        // F = I + F dt ; J = F J ; P = F P F^T + (G dt) Q (G dt)^T
        for (int i = 0; i < 15; ++i)
            for (int j = 0; j < 15; ++j) F(i, j) = (i == j ? 1.0 : 0.0) + F(i, j) * dt;
        J = matmul(F, J);
        DMat Ft(15, 15);
        for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) Ft(i, j) = F(j, i);
        DMat Gd(15, 12), Gdt(12, 15), Q(12, 12);
        for (int i = 0; i < 15; ++i) for (int j = 0; j < 12; ++j) { Gd(i, j) = G(i, j) * dt; Gdt(j, i) = Gd(i, j); }
        for (int k = 0; k < 3; ++k) {   // imu_noise::Q, :30-43 — na, nw, nba, nbw
            Q(0 + k, 0 + k) = prm->imu_noise_acc_sigma[k] * prm->imu_noise_acc_sigma[k];
            Q(3 + k, 3 + k) = prm->imu_noise_gyro_sigma[k] * prm->imu_noise_gyro_sigma[k];
            Q(6 + k, 6 + k) = prm->imu_bias_acc_sigma[k] * prm->imu_bias_acc_sigma[k];
            Q(9 + k, 9 + k) = prm->imu_bias_gyro_sigma[k] * prm->imu_bias_gyro_sigma[k];
        }
        DMat FPFt = matmul(matmul(F, P), Ft);
        DMat GQGt = matmul(matmul(Gd, Q), Gdt);
        for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) P(i, j) = FPFt(i, j) + GQGt(i, j);
        Dt += dt;
    }
};

class wheel_odom_preintegration {
public:
    explicit wheel_odom_preintegration(const params* prm_) : prm(prm_) { reset_wheel_odom_measure(-1); }
    void reset_wheel_odom_measure(double time) {
        last_update_time = time;
        delta_Tij = Iso3<double>();
        Dt = 0;
    }
    bool add_wheel_odom_measure(const wheel_sample& data) {
        if (last_update_time < 0) {
            last_add_wheel_odom_pose = data.pose;
            last_add_wheel_odom_time = data.time_stamp;
            last_update_time = data.time_stamp;
            delta_Tij = Iso3<double>();
            v = Vec3<double>(); omega = Vec3<double>();
            return false;
        }
        double dt = data.time_stamp - last_add_wheel_odom_time;
        Vec3<double> delta_p, delta_theta;
        lie::log_SE3<double>(last_add_wheel_odom_pose.inverse() * data.pose, delta_p, delta_theta);
        if (dt < 0.05) return false;
        v = delta_p / dt;
        omega = delta_theta / dt;
        double update_dt = data.time_stamp - last_update_time;
        update_by_v(update_dt);
        last_add_wheel_odom_pose = data.pose;
        last_add_wheel_odom_time = data.time_stamp;
        last_update_time = data.time_stamp;
        return true;
    }
    void update_only_t(double time) {
        if (last_update_time < 0) return;
        double update_dt = time - last_update_time;
        update_by_v(update_dt);
        last_update_time = time;
    }
    wheel_odom_preint_result get_preintegraption_result() const {
        Vec3<double> delta_p, delta_q;
        lie::log_SE3<double>(delta_Tij, delta_p, delta_q);
        double len_norm = std::max(squared_norm(delta_p), 0.005 * 0.005);
        double delta_yaw_norm = std::max(squared_norm(delta_q), 0.005 * 0.005);
        double kd[3] = {len_norm, len_norm, delta_yaw_norm};
        DMat cov(3, 3), inv, L;
        for (int k = 0; k < 3; ++k) cov(k, k) = prm->wheel_sigma[k] * prm->wheel_sigma[k] * kd[k];
        lu_inverse(cov, inv);
        llt_lower(inv, L);
        wheel_odom_preint_result r;
        r.delta_Tij = delta_Tij;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.sqrt_inverse_P[i][j] = L(j, i);
        r.Dt = Dt;
        return r;
    }

private:
    const params* prm;
    Iso3<double> delta_Tij;
    double last_update_time, last_add_wheel_odom_time = 0;
    Iso3<double> last_add_wheel_odom_pose;
    Vec3<double> omega, v;
    double Dt;
    void update_by_v(double dt) {
        if (dt <= 0 || dt >= 10) return;
        Dt = Dt + dt;
        Iso3<double> delta_T = lie::make_tf<double>(v * dt, omega * dt);
        delta_Tij = delta_Tij * delta_T;
    }
};

}  // namespace oracle
