// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Restatement of lvio_2d::solver (src/factor/solver.h:28-79, src/factor/solver.cpp) for the
// camera-less configuration every shipped config uses (config/office.yaml:6): the data model it
// mutates (frame_info src/trajectory/trajectory_type.h:9-75, laser_match / line
// src/trajectory/laser_type.h:13-21,76-85), do_init_solve / init_solve (solver.cpp:50-195),
// solve (:631-820), marginalization + clac_prior_J + clac_frame_J + marginalization_matrix
// (:4-40, :197-255, :257-442, :443-590).  Same names, same call order, same quirks
// (n-fold ground blocks, prior residual without linearized_R, laser blocks on
// the newest frame only in solve(), frame-0 anchoring in init, eigenvalue floor 1e-8).
#pragma once
#include <deque>
#include <memory>
#include <vector>

#include "factors.h"
#include "minimizer.h"
#include "preint.h"

namespace oracle {

struct line { Vec3<double> p1, p2; };
struct laser_match {
    std::vector<line> lines1, lines2;
    double p1[3], q1[3], p2[3], q2[3];
};
struct frame_info {
    enum frame_type { laser = 0, camera = 1, unknow = 2 };
    double p[3], q[3], v[3], bs[6];
    std::shared_ptr<imu_preint_result> imu_observation_reslut;          // i-1 ~ i
    std::shared_ptr<wheel_odom_preint_result> wheel_observation_reslut; // i-1 ~ i
    std::shared_ptr<laser_match> laser_match_ptr;
    frame_type type = unknow;
    double sqrt_H[36];
    typedef std::shared_ptr<frame_info> ptr;
};

// marginalization_matrix, src/factor/solver.cpp:4-40.  `dense_product` selects the GEMM that does
// the reference's full 2*rows*cols^2 flops (timed baseline) or the zero-skipping one (tests).
inline void marginalization_matrix(int r_len, const DMat& J, const std::vector<double>& R, DMat& Delta_H,
                                   std::vector<double>& Delta_g, DMat* H_out, std::vector<double>* g_out, bool dense_product) {
    DMat H;
    std::vector<double> g;
    if (dense_product) gemm_JtJ_dense(J, H); else gemm_JtJ(J, H);
    gemv_mJtR(J, R, g);
    const int m_len = int(g.size()) - r_len, r_index = m_len;
    DMat Hmm(m_len, m_len), Hmr(m_len, r_len), Hrm(r_len, m_len), Hrr(r_len, r_len);
    for (int i = 0; i < m_len; ++i) for (int j = 0; j < m_len; ++j) Hmm(i, j) = H(i, j);
    for (int i = 0; i < m_len; ++i) for (int j = 0; j < r_len; ++j) { Hmr(i, j) = H(i, r_index + j); Hrm(j, i) = H(r_index + j, i); }
    for (int i = 0; i < r_len; ++i) for (int j = 0; j < r_len; ++j) Hrr(i, j) = H(r_index + i, r_index + j);
    DMat Hmm_inverse;
    lu_inverse(Hmm, Hmm_inverse);
    DMat T = matmul(Hrm, Hmm_inverse);   // r x m
    DMat TH = matmul(T, Hmr);            // r x r
    Delta_H = DMat(r_len, r_len);
    for (int i = 0; i < r_len; ++i) for (int j = 0; j < r_len; ++j) Delta_H(i, j) = Hrr(i, j) - TH(i, j);
    Delta_g.assign(r_len, 0.0);
    for (int i = 0; i < r_len; ++i) {
        double s = 0.0;
        for (int k = 0; k < m_len; ++k) s += T(i, k) * g[k];
        Delta_g[i] = g[r_index + i] - s;
    }
    if (H_out) *H_out = H;
    if (g_out) *g_out = g;
}

class solver {
public:
    const params* prm;
    bool has_linearized_block = false;
    std::vector<double> linearized_X;          // 15
    DMat linearized_jacobians;                 // 15 x 15
    std::vector<double> linearized_residuals;  // 15
    // last marginalisation's dense pieces, kept for tests
    DMat J, H, Delta_H;
    std::vector<double> R, g, Delta_g;
    bool dense_product = false;
    miniceres::Options options;                // defaults = Ceres defaults; fast_mode caps iterations
    miniceres::Summary last_summary;

    explicit solver(const params* p) : prm(p) {}

    typedef std::deque<frame_info::ptr> frames;

    // ---- residual block factories shared by init_solve/solve (the reference's ::Create calls)
    void add_imu_wheel_ground(miniceres::Problem& problem, frames& fi) {
        const params* P = prm;
        for (size_t i = 1; i < fi.size(); ++i) {   // imu_factor, solver.cpp:116-126 / :701-711
            const imu_preint_result* r = fi[i]->imu_observation_reslut.get();
            problem.AddResidualBlock(15,
                {fi[i - 1]->p, fi[i - 1]->q, fi[i - 1]->v, fi[i - 1]->bs, fi[i]->p, fi[i]->q, fi[i]->v, fi[i]->bs},
                {3, 3, 3, 6, 3, 3, 3, 6},
                [P, r](const double* const* x, double* res, double** jac) {
                    imu_factor f(P, r);
                    if (jac) auto_diff::compute_res_and_jacobi<imu_factor, 15, 3, 3, 3, 6, 3, 3, 3, 6>(f, x, res, jac);
                    else auto_diff::compute_res_only<imu_factor, 15, 8>(f, x, res);
                });
            problem.SetParameterization(fi[i - 1]->q);
            problem.SetParameterization(fi[i]->q);
        }
        for (size_t i = 1; i < fi.size(); ++i) {   // wheel_factor, solver.cpp:129-138 / :714-723
            const wheel_odom_preint_result* r = fi[i]->wheel_observation_reslut.get();
            problem.AddResidualBlock(3, {fi[i - 1]->p, fi[i - 1]->q, fi[i]->p, fi[i]->q}, {3, 3, 3, 3},
                [P, r](const double* const* x, double* res, double** jac) {
                    wheel_odom_factor f(P, r);
                    if (jac) auto_diff::compute_res_and_jacobi<wheel_odom_factor, 3, 3, 3, 3, 3>(f, x, res, jac);
                    else auto_diff::compute_res_only<wheel_odom_factor, 3, 4>(f, x, res);
                });
            problem.SetParameterization(fi[i - 1]->q);
            problem.SetParameterization(fi[i]->q);
        }
        for (size_t i = 0; i < fi.size(); ++i)     // ground factor, n x n, solver.cpp:142-159 / :727-743
            for (size_t j = 0; j < fi.size(); ++j) {
                problem.AddResidualBlock(1, {fi[j]->p, fi[j]->q}, {3, 3},
                    [P](const double* const* x, double* res, double** jac) {
                        ground_factor_p f(P);
                        if (jac) auto_diff::compute_res_and_jacobi<ground_factor_p, 1, 3, 3>(f, x, res, jac);
                        else auto_diff::compute_res_only<ground_factor_p, 1, 2>(f, x, res);
                    });
                problem.AddResidualBlock(1, {fi[j]->p, fi[j]->q}, {3, 3},
                    [P](const double* const* x, double* res, double** jac) {
                        ground_factor_q f(P);
                        if (jac) auto_diff::compute_res_and_jacobi<ground_factor_q, 1, 3, 3>(f, x, res, jac);
                        else auto_diff::compute_res_only<ground_factor_q, 1, 2>(f, x, res);
                    });
                problem.SetParameterization(fi[j]->q);
            }
    }
    void add_laser(miniceres::Problem& problem, const line& a, const line& b, double* p1, double* q1, double* p2, double* q2) {
        const params* P = prm;
        laser_factor f(P, a.p1, a.p2, b.p1, b.p2);
        problem.AddResidualBlock(2, {p1, q1, p2, q2}, {3, 3, 3, 3},
            [f](const double* const* x, double* res, double** jac) {
                if (jac) auto_diff::compute_res_and_jacobi<laser_factor, 2, 3, 3, 3, 3>(f, x, res, jac);
                else auto_diff::compute_res_only<laser_factor, 2, 4>(f, x, res);
            });
    }

    // solver.cpp:50-169
    void do_init_solve(frames& fi) {
        miniceres::Problem problem;
        for (size_t i = 0; i < fi.size(); ++i)
            if (fi[i]->type == frame_info::laser && fi[i]->laser_match_ptr) {
                const int index1 = 0, index2 = int(i);
                auto& lm = *fi[i]->laser_match_ptr;
                for (size_t j = 0; j < lm.lines1.size(); ++j) {
                    add_laser(problem, lm.lines1[j], lm.lines2[j], fi[index1]->p, fi[index1]->q, fi[index2]->p, fi[index2]->q);
                    problem.SetParameterization(fi[index1]->q);
                    problem.SetParameterization(fi[index2]->q);
                }
            }
        add_imu_wheel_ground(problem, fi);
        miniceres::Options o = options;   // DENSE_SCHUR, default iteration cap (fast_mode is not consulted here)
        miniceres::Minimizer m(problem, o);
        last_summary = miniceres::Summary();
        m.Solve(&last_summary);
    }
    // solver.cpp:171-195
    void init_solve(frames& fi) {
        do_init_solve(fi);
        for (size_t i = 0; i < fi.size(); ++i)
            if (fi[i]->type == frame_info::laser && fi[i]->laser_match_ptr) {
                auto& lm = *fi[i]->laser_match_ptr;
                for (int k = 0; k < 3; ++k) {
                    lm.p1[k] = fi[0]->p[k]; lm.q1[k] = fi[0]->q[k];
                    lm.p2[k] = fi[i]->p[k]; lm.q2[k] = fi[i]->q[k];
                }
            }
    }
    // solver.cpp:631-820
    void solve(frames& fi) {
        miniceres::Problem problem;
        for (size_t i = fi.size() - 1; i < fi.size(); ++i)
            if (fi[i]->type == frame_info::laser && fi[i]->laser_match_ptr) {
                auto& lm = *fi[i]->laser_match_ptr;
                for (size_t j = 0; j < lm.lines1.size(); ++j) {
                    add_laser(problem, lm.lines1[j], lm.lines2[j], lm.p1, lm.q1, fi[i]->p, fi[i]->q);
                    problem.SetParameterBlockConstant(lm.p1);
                    problem.SetParameterBlockConstant(lm.q1);
                    problem.SetParameterization(lm.q1);
                    problem.SetParameterization(fi[i]->q);
                }
            }
        add_imu_wheel_ground(problem, fi);
        if (!prm->fast_mode && has_linearized_block) {
            auto fp = fi[fi.size() - 2];
            const double* LJ = linearized_jacobians.d.data();
            const double* LX = linearized_X.data();
            problem.AddResidualBlock(15, {fp->p, fp->q, fp->v, fp->bs}, {3, 3, 3, 6},
                [LJ, LX](const double* const* x, double* res, double** jac) {
                    marginalization_factor f(LJ, LX);
                    if (jac) auto_diff::compute_res_and_jacobi<marginalization_factor, 15, 3, 3, 3, 6>(f, x, res, jac);
                    else auto_diff::compute_res_only<marginalization_factor, 15, 4>(f, x, res);
                });
            problem.SetParameterization(fp->q);
        }
        for (size_t i = 0; i + 1 < fi.size(); ++i) {
            problem.SetParameterBlockConstant(fi[i]->p);
            problem.SetParameterBlockConstant(fi[i]->q);
            if (prm->fast_mode) problem.SetParameterBlockConstant(fi[i]->bs);
        }
        miniceres::Options o = options;
        if (prm->fast_mode) o.max_num_iterations = 10;
        miniceres::Minimizer m(problem, o);
        last_summary = miniceres::Summary();
        m.Solve(&last_summary);
        for (size_t i = fi.size() - 1; i < fi.size(); ++i)
            if (fi[i]->type == frame_info::laser && fi[i]->laser_match_ptr)
                for (int k = 0; k < 3; ++k) { fi[i]->laser_match_ptr->p2[k] = fi[i]->p[k]; fi[i]->laser_match_ptr->q2[k] = fi[i]->q[k]; }
    }

    // ---- marginalization, solver.cpp:257-442
    struct status_bolck_index { int p, q, v, bs, laser_res_index, imu_res_index, wheel_res_index, ground_res_index; };
    std::vector<status_bolck_index> all_status_block_indexs;

    void marginalization(frames& fi) {
        if (prm->fast_mode) return;
        const int n = int(fi.size());
        int all_X_size = n * 15, current_X_index = 0, current_res_index = 0;
        all_status_block_indexs.assign(n, status_bolck_index());
        if (has_linearized_block) current_res_index += int(linearized_X.size());
        for (int i = 0; i < n; ++i) {
            all_status_block_indexs[i].laser_res_index = current_res_index;
            int n_laser_match = 0;
            if (fi[i]->type == frame_info::laser && fi[i]->laser_match_ptr) n_laser_match = int(fi[i]->laser_match_ptr->lines1.size());
            current_res_index += n_laser_match * 2;
            all_status_block_indexs[i].imu_res_index = current_res_index;
            if (i > 0) current_res_index += 15;
            all_status_block_indexs[i].wheel_res_index = current_res_index;
            if (i > 0) current_res_index += 3;
            all_status_block_indexs[i].ground_res_index = current_res_index;
            current_res_index += n * 2;
        }
        for (int i = 0; i < n; ++i) {
            all_status_block_indexs[i].p = current_X_index; current_X_index += 3;
            all_status_block_indexs[i].q = current_X_index; current_X_index += 3;
            all_status_block_indexs[i].v = current_X_index; current_X_index += 3;
            all_status_block_indexs[i].bs = current_X_index; current_X_index += 6;
        }
        const int all_res_size = current_res_index;
        R.assign(all_res_size, 0.0);
        J = DMat(all_res_size, all_X_size);
        if (has_linearized_block) clac_prior_J(fi);
        for (int i = 0; i < n; ++i) clac_frame_J(fi, i);

        marginalization_matrix(15, J, R, Delta_H, Delta_g, &H, &g, dense_product);

        const double eps = 1e-8;
        std::vector<double> w;
        DMat V;
        jacobi_eigh(Delta_H, w, V);
        std::vector<double> S_sqrt(15), S_inv_sqrt(15);
        for (int i = 0; i < 15; ++i) {
            const double S = w[i] > eps ? w[i] : 0.0;
            const double S_inv = w[i] > eps ? 1.0 / w[i] : 0.0;
            S_sqrt[i] = std::sqrt(S);
            S_inv_sqrt[i] = std::sqrt(S_inv);
        }
        linearized_jacobians = DMat(15, 15);
        for (int i = 0; i < 15; ++i) for (int j = 0; j < 15; ++j) linearized_jacobians(i, j) = S_sqrt[i] * V(j, i);
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) fi.back()->sqrt_H[i * 6 + j] = linearized_jacobians(i, j);
        linearized_residuals.assign(15, 0.0);
        for (int i = 0; i < 15; ++i) {
            double s = 0.0;
            for (int k = 0; k < 15; ++k) s += V(k, i) * Delta_g[k];
            linearized_residuals[i] = -(S_inv_sqrt[i] * s);
        }
        auto fp = fi.back();
        linearized_X.assign(15, 0.0);
        for (int k = 0; k < 3; ++k) { linearized_X[k] = fp->p[k]; linearized_X[3 + k] = fp->q[k]; linearized_X[6 + k] = fp->v[k]; }
        for (int k = 0; k < 6; ++k) linearized_X[9 + k] = fp->bs[k];
        has_linearized_block = true;
    }

private:
    void put(int r, int c, int nr, int nc, const double* blk) {
        for (int i = 0; i < nr; ++i) for (int j = 0; j < nc; ++j) J(r + i, c + j) = blk[i * nc + j];
    }
    // solver.cpp:197-255
    void clac_prior_J(frames& fi) {
        const int last_prior_index = int(fi.size()) - 2;
        auto fp = fi[last_prior_index];
        marginalization_factor f(linearized_jacobians.d.data(), linearized_X.data());
        const double* x[4] = {fp->p, fp->q, fp->v, fp->bs};
        double res[15], j0[45], j1[45], j2[45], j3[90];
        double* jac[4] = {j0, j1, j2, j3};
        auto_diff::compute_res_and_jacobi<marginalization_factor, 15, 3, 3, 3, 6>(f, x, res, jac);
        auto& ix = all_status_block_indexs[last_prior_index];
        put(0, ix.p, 15, 3, j0); put(0, ix.q, 15, 3, j1); put(0, ix.v, 15, 3, j2); put(0, ix.bs, 15, 6, j3);
        for (int i = 0; i < 15; ++i) R[i] = res[i];
    }
    // solver.cpp:443-590
    void clac_frame_J(frames& fi, int index) {
        auto& ix = all_status_block_indexs[index];
        int r_index = ix.laser_res_index;
        if (fi[index]->type == frame_info::laser && fi[index]->laser_match_ptr) {
            auto& lm = *fi[index]->laser_match_ptr;
            for (size_t j = 0; j < lm.lines1.size(); ++j) {
                laser_factor f(prm, lm.lines1[j].p1, lm.lines1[j].p2, lm.lines2[j].p1, lm.lines2[j].p2);
                const double* x[4] = {lm.p1, lm.q1, fi[index]->p, fi[index]->q};
                double res[2], j0[6], j1[6], j2[6], j3[6];
                double* jac[4] = {j0, j1, j2, j3};
                auto_diff::compute_res_and_jacobi<laser_factor, 2, 3, 3, 3, 3>(f, x, res, jac);
                put(r_index, ix.p, 2, 3, j2);
                put(r_index, ix.q, 2, 3, j3);
                R[r_index] = res[0]; R[r_index + 1] = res[1];
                r_index += 2;
            }
        }
        if (index > 0) {
            auto& im = all_status_block_indexs[index - 1];
            imu_factor f(prm, fi[index]->imu_observation_reslut.get());
            const double* x[8] = {fi[index - 1]->p, fi[index - 1]->q, fi[index - 1]->v, fi[index - 1]->bs,
                                  fi[index]->p, fi[index]->q, fi[index]->v, fi[index]->bs};
            double res[15], jb[8][90];
            double* jac[8] = {jb[0], jb[1], jb[2], jb[3], jb[4], jb[5], jb[6], jb[7]};
            auto_diff::compute_res_and_jacobi<imu_factor, 15, 3, 3, 3, 6, 3, 3, 3, 6>(f, x, res, jac);
            put(r_index, im.p, 15, 3, jb[0]); put(r_index, im.q, 15, 3, jb[1]); put(r_index, im.v, 15, 3, jb[2]); put(r_index, im.bs, 15, 6, jb[3]);
            put(r_index, ix.p, 15, 3, jb[4]); put(r_index, ix.q, 15, 3, jb[5]); put(r_index, ix.v, 15, 3, jb[6]); put(r_index, ix.bs, 15, 6, jb[7]);
            for (int i = 0; i < 15; ++i) R[r_index + i] = res[i];
            r_index += 15;
        }
        if (index > 0) {
            auto& im = all_status_block_indexs[index - 1];
            wheel_odom_factor f(prm, fi[index]->wheel_observation_reslut.get());
            const double* x[4] = {fi[index - 1]->p, fi[index - 1]->q, fi[index]->p, fi[index]->q};
            double res[3], jb[4][9];
            double* jac[4] = {jb[0], jb[1], jb[2], jb[3]};
            auto_diff::compute_res_and_jacobi<wheel_odom_factor, 3, 3, 3, 3, 3>(f, x, res, jac);
            put(r_index, im.p, 3, 3, jb[0]); put(r_index, im.q, 3, 3, jb[1]);
            put(r_index, ix.p, 3, 3, jb[2]); put(r_index, ix.q, 3, 3, jb[3]);
            for (int i = 0; i < 3; ++i) R[r_index + i] = res[i];
            r_index += 3;
        }
        for (size_t j = 0; j < fi.size(); ++j) {
            auto& jx = all_status_block_indexs[j];
            const double* x[2] = {fi[j]->p, fi[j]->q};
            double res[1], j0[3], j1[3];
            double* jac[2] = {j0, j1};
            {
                ground_factor_p f(prm);
                auto_diff::compute_res_and_jacobi<ground_factor_p, 1, 3, 3>(f, x, res, jac);
                put(r_index, jx.p, 1, 3, j0); put(r_index, jx.q, 1, 3, j1);
                R[r_index] = res[0];
            }
            r_index++;
            {
                ground_factor_q f(prm);
                auto_diff::compute_res_and_jacobi<ground_factor_q, 1, 3, 3>(f, x, res, jac);
                put(r_index, jx.p, 1, 3, j0); put(r_index, jx.q, 1, 3, j1);
                R[r_index] = res[0];
            }
            r_index++;
        }
    }
};

}  // namespace oracle
