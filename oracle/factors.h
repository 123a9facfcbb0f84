// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Restatement of the reference's residual functors, one struct per functor, same names, same
// parameter-block order, same arithmetic structure:
//   laser_factor            src/factor/laser_factor.h:26-100
//   imu_factor              src/factor/imu_factor.h:7-100
//   wheel_odom_factor       src/factor/wheel_factor.h:6-82
//   ground_factor_p / _q    src/factor/ground_factor.h:25-89
//   marginalization_factor  src/factor/marginalization_factor.h:6-77   (no camera points)
// and of the autodiff harness auto_diff::compute_res_and_jacobi (src/utilies/common.h:198-217):
// one Jet<sum of block sizes> pass, Jacobians w.r.t. the AMBIENT parameters, row-major blocks.
#pragma once
#include <algorithm>
#include <vector>

#include "lie.h"

namespace oracle {

// The subset of param::manager (src/utilies/params.h) the path reads (SURVEY §2 "Parameters").
struct params {
    Iso3<double> T_imu_to_wheel;
    Iso3<double> T_imu_to_laser;
    double g = 9.8;
    double line_to_line_sigma = 0.001;
    double manifold_p_sigma = 0.01;
    double manifold_q_sigma = 0.0005;
    double imu_noise_acc_sigma[3] = {0.0163, 0.0163, 0.0163};
    double imu_bias_acc_sigma[3] = {0.00499, 0.00499, 0.00499};
    double imu_noise_gyro_sigma[3] = {0.003208, 0.003208, 0.003208};
    double imu_bias_gyro_sigma[3] = {0.000499, 0.000499, 0.000499};
    double wheel_sigma[3] = {0.5, 99999.0, 999.99};
    bool fast_mode = false;
};

// magic_number_X, src/factor/factor_common.h:7-33
constexpr int alpha_index = 0, beta_index = 3, gamma_index = 6, ba_index = 9, bw_index = 12, all_status_len = 15;

struct imu_preint_result {      // src/factor/imu_preintegraption.h:45-67
    double X[15];               // alpha beta gamma ba bw
    double J[15][15];           // J[r][c]
    double sqrt_inverse_P[15][15];
    double Dt;
};
struct wheel_odom_preint_result {  // src/factor/wheel_odom_preintegration.h:25-42
    Iso3<double> delta_Tij;
    double sqrt_inverse_P[3][3];
    double Dt;
};

struct laser_factor {
    const params* prm;
    Vec3<double> l1_p1, l1_p2, l2_p1, l2_p2;
    double len1, len2, sum;
    laser_factor(const params* prm_, const Vec3<double>& a, const Vec3<double>& b, const Vec3<double>& c, const Vec3<double>& d)
        : prm(prm_), l1_p1(a), l1_p2(b), l2_p1(c), l2_p2(d) {
        len1 = norm(l1_p1 - l1_p2);
        len2 = norm(l2_p1 - l2_p2);
        double tmp = std::min(len1, len2);
        sum = tmp / 2.0 / 0.02;
        sum = std::sqrt(sum);
    }
    template <typename T>
    bool operator()(const T* const p_w_i, const T* const theta_w_i, const T* const p_w_j, const T* const theta_w_j, T* res) const {
        Vec3<T> p_i(p_w_i[0], p_w_i[1], p_w_i[2]), theta_i(theta_w_i[0], theta_w_i[1], theta_w_i[2]);
        Vec3<T> p_j(p_w_j[0], p_w_j[1], p_w_j[2]), theta_j(theta_w_j[0], theta_w_j[1], theta_w_j[2]);
        Iso3<T> T_i_l = cast_iso<T>(prm->T_imu_to_laser);
        Iso3<T> T_w_i = lie::make_tf<T>(p_i, theta_i) * T_i_l;
        Iso3<T> T_w_j = lie::make_tf<T>(p_j, theta_j) * T_i_l;
        Vec3<T> l2_point1 = T_w_j * cast3<T>(l2_p1);
        Vec3<T> l2_point2 = T_w_j * cast3<T>(l2_p2);
        Vec3<T> l1_point1 = T_w_i * cast3<T>(l1_p1);
        Vec3<T> l1_point2 = T_w_i * cast3<T>(l1_p2);
        l2_point1(2) = T(0.0);
        l2_point2(2) = T(0.0);
        l1_point1(2) = T(0.0);
        l1_point2(2) = T(0.0);
        T dis1 = e_laser::dis_from_line<T>(l2_point1, l1_point1, l1_point2);
        T dis2 = e_laser::dis_from_line<T>(l2_point2, l1_point1, l1_point2);
        const double sqrt_info = 1.0 / prm->line_to_line_sigma;  // laser_noise, laser_factor.h:19-24
        T e1 = T(sqrt_info) * dis1;
        T e2 = T(sqrt_info) * dis2;
        res[0] = T(sum) * e1;
        res[1] = T(sum) * e2;
        return true;
    }
};

struct imu_factor {
    const params* prm;
    const imu_preint_result* r;
    imu_factor(const params* prm_, const imu_preint_result* r_) : prm(prm_), r(r_) {}
    template <typename T>
    bool operator()(const T* const p_w_i, const T* const theta_w_i, const T* const v_w_i, const T* const bs_w_i,
                    const T* const p_w_j, const T* const theta_w_j, const T* const v_w_j, const T* const bs_w_j, T* res) const {
        Vec3<T> pi(p_w_i[0], p_w_i[1], p_w_i[2]), vi(v_w_i[0], v_w_i[1], v_w_i[2]), thetai(theta_w_i[0], theta_w_i[1], theta_w_i[2]);
        Vec3<T> bai(bs_w_i[0], bs_w_i[1], bs_w_i[2]), bwi(bs_w_i[3], bs_w_i[4], bs_w_i[5]);
        Vec3<T> pj(p_w_j[0], p_w_j[1], p_w_j[2]), vj(v_w_j[0], v_w_j[1], v_w_j[2]), thetaj(theta_w_j[0], theta_w_j[1], theta_w_j[2]);
        Vec3<T> baj(bs_w_j[0], bs_w_j[1], bs_w_j[2]), bwj(bs_w_j[3], bs_w_j[4], bs_w_j[5]);

        T g_norm = T(prm->g);
        Vec3<T> g(T(0.0), T(0.0), T(1.0));

        auto X3 = [&](int off) { return Vec3<T>(T(r->X[off]), T(r->X[off + 1]), T(r->X[off + 2])); };
        Vec3<T> alpha = X3(alpha_index), beta = X3(beta_index), gamma = X3(gamma_index), ba = X3(ba_index), bw = X3(bw_index);
        T Dt = T(r->Dt);

        Mat3<T> bk_R_w = lie::exp_so3<T>(-thetai);

        auto Jb = [&](int ro, int co) {
            Mat3<T> m;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) m(i, j) = T(r->J[ro + i][co + j]);
            return m;
        };
        Mat3<T> alpha_J_ba = Jb(alpha_index, ba_index), alpha_J_bw = Jb(alpha_index, bw_index);
        Mat3<T> beta_J_ba = Jb(beta_index, ba_index), beta_J_bw = Jb(beta_index, bw_index);
        Mat3<T> gamma_J_bw = Jb(gamma_index, bw_index);

        alpha = alpha + alpha_J_ba * (bai - ba) + alpha_J_bw * (bwi - bw);
        beta = beta + beta_J_ba * (bai - ba) + beta_J_bw * (bwi - bw);
        gamma = gamma + gamma_J_bw * (bwi - bw);

        Vec3<T> res_alpha = alpha - bk_R_w * (pj - pi + T(0.5) * g * g_norm * Dt * Dt - vi * Dt);
        Vec3<T> res_beta = beta - bk_R_w * (vj + g * g_norm * Dt - vi);
        Vec3<T> res_gamma = lie::log_SO3<T>(lie::exp_so3<T>(-gamma) * (lie::exp_so3<T>(-thetai) * lie::exp_so3<T>(thetaj)));
        Vec3<T> res_ba = baj - bai;
        Vec3<T> res_bw = bwj - bwi;

        T raw[15];
        for (int k = 0; k < 3; ++k) {
            raw[alpha_index + k] = res_alpha(k);
            raw[beta_index + k] = res_beta(k);
            raw[gamma_index + k] = res_gamma(k);
            raw[ba_index + k] = res_ba(k);
            raw[bw_index + k] = res_bw(k);
        }
        // res_all = sqrt_info * res_all (dense 15x15, imu_factor.h:85-86)
        for (int i = 0; i < 15; ++i) {
            T s = T(r->sqrt_inverse_P[i][0]) * raw[0];
            for (int k = 1; k < 15; ++k) s = s + T(r->sqrt_inverse_P[i][k]) * raw[k];
            res[i] = s;
        }
        return true;
    }
};

struct wheel_odom_factor {
    const params* prm;
    const wheel_odom_preint_result* r;
    wheel_odom_factor(const params* prm_, const wheel_odom_preint_result* r_) : prm(prm_), r(r_) {}
    template <typename T>
    bool operator()(const T* const p_w_i, const T* const theta_w_i, const T* const p_w_j, const T* const theta_w_j, T* res) const {
        Vec3<T> pi(p_w_i[0], p_w_i[1], p_w_i[2]), thetai(theta_w_i[0], theta_w_i[1], theta_w_i[2]);
        Vec3<T> pj(p_w_j[0], p_w_j[1], p_w_j[2]), thetaj(theta_w_j[0], theta_w_j[1], theta_w_j[2]);
        Iso3<T> T_i_w = cast_iso<T>(prm->T_imu_to_wheel);
        Iso3<T> tf_i = lie::make_tf<T>(pi, thetai) * T_i_w;
        Iso3<T> tf_j = lie::make_tf<T>(pj, thetaj) * T_i_w;
        Iso3<T> w_tf_ij = tf_i.inverse() * tf_j;

        Vec3<T> p, q, op, oq;
        lie::log_SE3<T>(w_tf_ij, p, q);
        lie::log_SE3<T>(cast_iso<T>(r->delta_Tij), op, oq);

        T o_len = sqrt(op(0) * op(0) + op(1) * op(1));
        T len = sqrt(p(0) * p(0) + p(1) * p(1));

        Vec3<T> o_dir(op(0), op(1), T(0.0));
        Vec3<T> dir(p(0), p(1), T(0.0));
        T angle = T(0.0);
        if (norm(o_dir) > T(0.0001) && norm(dir) > T(0.0001)) {
            o_dir = normalized(o_dir);
            dir = normalized(dir);
            T sinn = norm(cross(o_dir, dir));
            angle = asin(sinn);
        } else {
            angle = norm(dir);
        }
        if (len < T(0.0001) || o_len < T(0.0001))
            res[0] = T(r->sqrt_inverse_P[0][0]) * len;
        else
            res[0] = T(r->sqrt_inverse_P[0][0]) * (o_len - len);
        res[1] = T(r->sqrt_inverse_P[1][1]) * angle;
        if (norm(q) < T(0.001) || norm(oq) < T(0.001))
            res[2] = T(r->sqrt_inverse_P[2][2]) * norm(q);
        else
            res[2] = T(r->sqrt_inverse_P[2][2]) * (norm(oq) - norm(q));
        return true;
    }
};

struct ground_factor_p {
    const params* prm;
    explicit ground_factor_p(const params* prm_) : prm(prm_) {}
    template <typename T>
    bool operator()(const T* const p_w_i, const T* const theta_w_i, T* res) const {
        Iso3<T> tf_w_i = lie::make_tf<T>(Vec3<T>(p_w_i[0], p_w_i[1], p_w_i[2]), Vec3<T>(theta_w_i[0], theta_w_i[1], theta_w_i[2]));
        Iso3<T> T_i_w = cast_iso<T>(prm->T_imu_to_wheel);
        Iso3<T> tf_w_o = tf_w_i * T_i_w;
        T dis_from_plane = tf_w_o.t(2);
        res[0] = T(1.0 / prm->manifold_p_sigma) * dis_from_plane;
        return true;
    }
};

struct ground_factor_q {
    const params* prm;
    explicit ground_factor_q(const params* prm_) : prm(prm_) {}
    template <typename T>
    bool operator()(const T* const p_w_i, const T* const theta_w_i, T* res) const {
        Iso3<T> tf_w_i = lie::make_tf<T>(Vec3<T>(p_w_i[0], p_w_i[1], p_w_i[2]), Vec3<T>(theta_w_i[0], theta_w_i[1], theta_w_i[2]));
        Iso3<T> T_i_w = cast_iso<T>(prm->T_imu_to_wheel);
        Iso3<T> tf_w_o = tf_w_i * T_i_w;
        Vec3<T> ABC(T(0.0), T(0.0), T(1.0));
        Vec3<T> z_axis = tf_w_o.R.col(2);
        T sinn = norm(cross(z_axis, ABC));
        T angle = asin(sinn);
        res[0] = T(1.0 / prm->manifold_q_sigma) * angle;
        return true;
    }
};

// marginalization_factor with n_world_point == 0: r = linearized_J (X - linearized_X);
// `linearized_R` is NOT added (commented out in the reference, marginalization_factor.h:50).
struct marginalization_factor {
    const double* linearized_J;  // 15x15 row-major
    const double* linearized_X;  // 15
    marginalization_factor(const double* J_, const double* X_) : linearized_J(J_), linearized_X(X_) {}
    template <typename T>
    bool operator()(const T* const p, const T* const q, const T* const v, const T* const bs, T* res) const {
        T X[15];
        for (int i = 0; i < 3; ++i) X[i] = p[i];
        for (int i = 0; i < 3; ++i) X[i + 3] = q[i];
        for (int i = 0; i < 3; ++i) X[i + 6] = v[i];
        for (int i = 0; i < 6; ++i) X[i + 9] = bs[i];
        for (int i = 0; i < 15; ++i) {
            T s = T(linearized_J[i * 15 + 0]) * (X[0] - T(linearized_X[0]));
            for (int k = 1; k < 15; ++k) s = s + T(linearized_J[i * 15 + k]) * (X[k] - T(linearized_X[k]));
            res[i] = s;
        }
        return true;
    }
};

// ---------------------------------------------------------------------------------------------
// autodiff harness: evaluates FACTOR with Jet<sum(block sizes)>; jac[b] is n_r x size(b) row-major.
// (ceres::AutoDiffCostFunction::Evaluate semantics as used by src/utilies/common.h:201-217)
namespace auto_diff {
template <int... Ns> struct sum_of;
template <> struct sum_of<> { static constexpr int value = 0; };
template <int N0, int... Ns> struct sum_of<N0, Ns...> { static constexpr int value = N0 + sum_of<Ns...>::value; };

template <typename FACTOR, int n_r, int... n_xs, size_t... Is>
inline void call_functor(const FACTOR& f, Jet<sum_of<n_xs...>::value>* const* blocks, Jet<sum_of<n_xs...>::value>* res,
                         std::index_sequence<Is...>) {
    f(blocks[Is]..., res);
}

template <typename FACTOR, int n_r, int... n_xs>
inline void compute_res_and_jacobi(const FACTOR& f, const double* const* parameters, double* res, double** jacobians) {
    constexpr int NB = sizeof...(n_xs);
    constexpr int NP = sum_of<n_xs...>::value;
    constexpr int sizes[NB] = {n_xs...};
    using J = Jet<NP>;
    J x[NP];
    J* blocks[NB];
    int off = 0;
    for (int b = 0; b < NB; ++b) {
        blocks[b] = x + off;
        for (int k = 0; k < sizes[b]; ++k) x[off + k] = J(parameters[b][k], off + k);
        off += sizes[b];
    }
    J out[n_r];
    call_functor<FACTOR, n_r, n_xs...>(f, blocks, out, std::make_index_sequence<NB>{});
    off = 0;
    for (int i = 0; i < n_r; ++i) res[i] = out[i].a;
    for (int b = 0; b < NB; ++b) {
        if (jacobians && jacobians[b])
            for (int i = 0; i < n_r; ++i)
                for (int k = 0; k < sizes[b]; ++k) jacobians[b][i * sizes[b] + k] = out[i].v[off + k];
        off += sizes[b];
    }
}
// residual-only evaluation (T = double), what Ceres does when no Jacobians are requested
template <typename FACTOR, int n_r, size_t... Is>
inline void compute_res_only_impl(const FACTOR& f, const double* const* parameters, double* res, std::index_sequence<Is...>) {
    f(parameters[Is]..., res);
}
template <typename FACTOR, int n_r, int NB>
inline void compute_res_only(const FACTOR& f, const double* const* parameters, double* res) {
    compute_res_only_impl<FACTOR, n_r>(f, parameters, res, std::make_index_sequence<NB>{});
}
}  // namespace auto_diff

}  // namespace oracle
