// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Restatement of the back-end pose-graph relinearisation:
//   edge_noise / edge_factor     src/factor/edge_factor.h:4-26, :79-126 (J(1,2) as written at :19)
//   keyframe_manager::solve      src/trajectory/keyframe_manager.cpp:722-838 (first pose of the first sequential edge constant,
//                                loop edges weighted by loop_edge_k, ground factors gated by use_ground_{p,q}_factor, Ceres defaults)
#pragma once
#include <vector>

#include "factors.h"
#include "minimizer.h"

namespace oracle {

struct pg_params {
    double loop_sigma_p[3] = {0.1, 0.1, 0.1}, loop_sigma_q[3] = {0.01, 0.01, 0.01};
    double loop_edge_k = 10.0;
    bool use_ground_p_factor = true, use_ground_q_factor = true;
};
struct edge_noise {
    double J[6][6];
    explicit edge_noise(const pg_params& P) {
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) J[i][j] = i == j ? 1.0 : 0.0;
        J[0][0] = 1.0 / P.loop_sigma_p[0];
        J[1][2] = 1.0 / P.loop_sigma_p[1];
        J[2][2] = 1.0 / P.loop_sigma_p[2];
        J[3][3] = 1.0 / P.loop_sigma_q[0];
        J[4][4] = 1.0 / P.loop_sigma_q[1];
        J[5][5] = 1.0 / P.loop_sigma_q[2];
    }
};
struct edge_factor {
    Iso3<double> tf12;
    double weight;
    const edge_noise* noise;
    edge_factor(const Iso3<double>& tf12_, double weight_, const edge_noise* n) : tf12(tf12_), weight(weight_), noise(n) {}
    template <typename T>
    bool operator()(const T* const p_w_i, const T* const theta_w_i, const T* const p_w_j, const T* const theta_w_j, T* res) const {
        Iso3<T> tf_i = lie::make_tf<T>(Vec3<T>(p_w_i[0], p_w_i[1], p_w_i[2]), Vec3<T>(theta_w_i[0], theta_w_i[1], theta_w_i[2]));
        Iso3<T> tf_j = lie::make_tf<T>(Vec3<T>(p_w_j[0], p_w_j[1], p_w_j[2]), Vec3<T>(theta_w_j[0], theta_w_j[1], theta_w_j[2]));
        Iso3<T> error = tf_j.inverse() * tf_i * cast_iso<T>(tf12);
        Vec3<T> res_p, res_theta;
        lie::log_SE3<T>(error, res_p, res_theta);
        T all[6] = {res_p(0), res_p(1), res_p(2), res_theta(0), res_theta(1), res_theta(2)};
        for (int r = 0; r < 6; ++r) {
            T s(0.0);
            for (int c = 0; c < 6; ++c) s = s + T(noise->J[r][c]) * all[c];
            res[r] = T(weight) * s;
        }
        return true;
    }
};

// poses [N][6] in/out
inline void keyframe_manager_build(miniceres::Problem& problem, const edge_noise& noise, const params* prm, const pg_params& P, int N, double* poses,
                                   const std::vector<std::pair<int, int>>& seq_idx, const std::vector<Iso3<double>>& seq_tf,
                                   const std::vector<std::pair<int, int>>& loop_idx, const std::vector<Iso3<double>>& loop_tf) {
    auto add_edge = [&](int i1, int i2, const Iso3<double>& tf, double w) {
        double* pi = poses + 6 * i1; double* qi = pi + 3; double* pj = poses + 6 * i2; double* qj = pj + 3;
        const edge_noise* nz = &noise;
        problem.AddResidualBlock(6, {pi, qi, pj, qj}, {3, 3, 3, 3}, [tf, w, nz](const double* const* x, double* res, double** jac) {
            edge_factor f(tf, w, nz);
            if (jac) auto_diff::compute_res_and_jacobi<edge_factor, 6, 3, 3, 3, 3>(f, x, res, jac);
            else auto_diff::compute_res_only<edge_factor, 6, 4>(f, x, res);
        });
        problem.SetParameterization(qi);
        problem.SetParameterization(qj);
    };
    for (size_t i = 0; i < seq_idx.size(); i++) {
        add_edge(seq_idx[i].first, seq_idx[i].second, seq_tf[i], 1);
        if (i == 0) {
            problem.SetParameterBlockConstant(poses + 6 * seq_idx[i].first + 3);
            problem.SetParameterBlockConstant(poses + 6 * seq_idx[i].first);
        }
    }
    for (size_t i = 0; i < loop_idx.size(); i++) add_edge(loop_idx[i].first, loop_idx[i].second, loop_tf[i], P.loop_edge_k);
    if (P.use_ground_p_factor)
        for (int i = 0; i < N; i++)
            problem.AddResidualBlock(1, {poses + 6 * i, poses + 6 * i + 3}, {3, 3}, [prm](const double* const* x, double* res, double** jac) {
                ground_factor_p f(prm);
                if (jac) auto_diff::compute_res_and_jacobi<ground_factor_p, 1, 3, 3>(f, x, res, jac);
                else auto_diff::compute_res_only<ground_factor_p, 1, 2>(f, x, res);
            });
    if (P.use_ground_q_factor)
        for (int i = 0; i < N; i++)
            problem.AddResidualBlock(1, {poses + 6 * i, poses + 6 * i + 3}, {3, 3}, [prm](const double* const* x, double* res, double** jac) {
                ground_factor_q f(prm);
                if (jac) auto_diff::compute_res_and_jacobi<ground_factor_q, 1, 3, 3>(f, x, res, jac);
                else auto_diff::compute_res_only<ground_factor_q, 1, 2>(f, x, res);
            });
}
inline void keyframe_manager_solve(const params* prm, const pg_params& P, int N, double* poses, const std::vector<std::pair<int, int>>& seq_idx,
                                   const std::vector<Iso3<double>>& seq_tf, const std::vector<std::pair<int, int>>& loop_idx,
                                   const std::vector<Iso3<double>>& loop_tf, int max_iters, miniceres::Summary* summary) {
    edge_noise noise(P);
    miniceres::Problem problem;
    keyframe_manager_build(problem, noise, prm, P, N, poses, seq_idx, seq_tf, loop_idx, loop_tf);
    miniceres::Options o;
    if (max_iters > 0) o.max_num_iterations = max_iters;
    miniceres::Minimizer m(problem, o);
    *summary = miniceres::Summary();
    m.Solve(summary);
}
// tangent-space normal equations at `poses` over the NON-constant blocks in insertion order (tests): H [nt][nt], g [nt], cost
inline int keyframe_manager_linearize(const params* prm, const pg_params& P, int N, double* poses, const std::vector<std::pair<int, int>>& seq_idx,
                                      const std::vector<Iso3<double>>& seq_tf, const std::vector<std::pair<int, int>>& loop_idx,
                                      const std::vector<Iso3<double>>& loop_tf, std::vector<double>& H, std::vector<double>& g, double& cost,
                                      std::vector<int>& pose_of_tangent) {
    edge_noise noise(P);
    miniceres::Problem problem;
    keyframe_manager_build(problem, noise, prm, P, N, poses, seq_idx, seq_tf, loop_idx, loop_tf);
    miniceres::Minimizer m(problem, miniceres::Options());
    m.n_amb = m.n_tan = 0;
    pose_of_tangent.clear();
    for (auto& b : problem.pblocks) {
        b.amb_off = m.n_amb; m.n_amb += b.size;
        if (!b.constant) { b.tan_off = m.n_tan; m.n_tan += b.size; for (int k = 0; k < b.size; ++k) pose_of_tangent.push_back((int)(b.user - poses) + k); }
    }
    m.x.assign(m.n_amb, 0.0);
    for (auto& b : problem.pblocks) for (int k = 0; k < b.size; ++k) m.x[b.amb_off + k] = b.user[k];
    m.active_r.clear();
    for (int ri = 0; ri < int(problem.rblocks.size()); ++ri) {
        bool any = false;
        for (int b : problem.rblocks[ri].blocks) if (!problem.pblocks[b].constant) any = true;
        if (any) m.active_r.push_back(ri);
    }
    cost = m.evaluate(m.x, true);
    H = m.H.d;
    g = m.g;
    return m.n_tan;
}

}  // namespace oracle
