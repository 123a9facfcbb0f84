// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Restatement of the slice of Ceres Solver (1.14.x, third-party, NOT vendored in the reference;
// CMakeLists.txt:18, docker/Dockerfile:46) that `ceres::Solve` runs for the reference's two
// call sites, src/factor/solver.cpp:161-168 (DENSE_SCHUR) and :795-802 (SPARSE_SCHUR):
// Problem bookkeeping (parameter blocks keyed by pointer, constant blocks, the so3 local
// parameterisation of src/factor/factor_common.h:37-60 differentiated with Jets like
// AutoDiffLocalParameterization), removal of residual blocks whose blocks are all constant,
// and the TRUST_REGION / LEVENBERG_MARQUARDT minimiser with the default options listed in
// SURVEY.md Appendix B (Jacobi scaling computed at iteration 0; LM diagonal clamp
// [1e-6,1e32]; rho-based radius update; the step that trips function/parameter tolerance is
// NOT applied — parameters are only written back after successful steps).
// Evaluation failure (round 3): a residual or a Jacobian entry of a non-constant block that is not finite fails the evaluation
// (internal/ceres/residual_block.cc ResidualBlock::Evaluate -> IsEvaluationValid, array_utils.cc IsArrayValid); at iteration 0 and
// after a successful step that ends the solve with termination FAILURE (trust_region_minimizer.cc IterationZero /
// HandleSuccessfulStep -> EvaluateGradientAndJacobian), a failed candidate evaluation is a step of infinite cost; and a FAILURE
// termination — also max_num_consecutive_invalid_steps — hands back the parameters the solve started from (solver.cc Minimize():
// StateVectorToParameterBlocks(IsSolutionUsable() ? reduced_parameters : original_reduced_parameters)).  Restated from the published
// Ceres 1.14 sources; exercised by the exactly stationary wheel interval (tests/test_oracle_stationary.py).
// The linear solve is an exact dense Cholesky of the damped normal equations: both Schur
// variants are exact solvers of the same system, so only round-off differs.
#pragma once
#include <cmath>
#include <functional>
#include <limits>
#include <map>
#include <vector>

#include "dense.h"
#include "lie.h"

namespace oracle {
namespace miniceres {

struct Options {
    int max_num_iterations = 50;
    double initial_trust_region_radius = 1e4;
    double max_trust_region_radius = 1e16;
    double min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3;
    double min_lm_diagonal = 1e-6;
    double max_lm_diagonal = 1e32;
    double function_tolerance = 1e-6;
    double gradient_tolerance = 1e-10;
    double parameter_tolerance = 1e-8;
    bool jacobi_scaling = true;
    int max_num_consecutive_invalid_steps = 5;
    int num_threads = 1;   // Ceres Solver::Options::num_threads (Jacobian evaluation); the reference leaves it at 1 (solver.cpp:798 commented out)
};

struct IterationRecord {
    int iteration;
    double cost;             // x_cost_ after this iteration (without fixed cost)
    double candidate_cost;
    double model_cost_change;
    double relative_decrease;
    double radius;           // after the update
    bool step_is_valid, step_is_successful;
    std::vector<double> x;   // ambient state of all non-constant blocks, after this iteration
};

struct Summary {
    int num_iterations = 0;       // LM iterations executed (excluding iteration 0)
    int num_successful_steps = 0;
    int termination = 0;          // 0 none, 1 gradient tol, 2 function tol, 3 parameter tol, 4 max iters, 5 min radius, 6 failure
    double initial_cost = 0, final_cost = 0, fixed_cost = 0;
    std::vector<IterationRecord> iterations;
    std::vector<std::pair<double*, int>> layout;   // (user pointer, size) of the non-constant blocks, in x order
};

typedef std::function<void(const double* const* params, double* res, double** jac)> EvalFn;

class Problem {
public:
    struct PBlock { double* user; int size; bool constant = false; bool so3 = false; int amb_off = -1; int tan_off = -1; };
    struct RBlock { int n_res; std::vector<int> blocks; EvalFn eval; };
    std::vector<PBlock> pblocks;
    std::vector<RBlock> rblocks;
    std::map<double*, int> index;

    int block_id(double* p, int size) {
        auto it = index.find(p);
        if (it != index.end()) return it->second;
        PBlock b; b.user = p; b.size = size;
        pblocks.push_back(b);
        index[p] = int(pblocks.size()) - 1;
        return int(pblocks.size()) - 1;
    }
    void AddResidualBlock(int n_res, const std::vector<double*>& ptrs, const std::vector<int>& sizes, EvalFn fn) {
        RBlock r; r.n_res = n_res; r.eval = fn;
        for (size_t i = 0; i < ptrs.size(); ++i) r.blocks.push_back(block_id(ptrs[i], sizes[i]));
        rblocks.push_back(r);
    }
    void SetParameterBlockConstant(double* p) { pblocks[index.at(p)].constant = true; }
    void SetParameterization(double* p) { pblocks[index.at(p)].so3 = true; }
};

// so3_parameterization::operator() (src/factor/factor_common.h:41-53) + its Jet<3> Jacobian at delta = 0
inline void so3_plus(const double* x, const double* delta, double* out) {
    Vec3<double> tmp(x[0] + delta[0], x[1] + delta[1], x[2] + delta[2]);
    lie::normalize_so3<double>(tmp);
    out[0] = tmp(0); out[1] = tmp(1); out[2] = tmp(2);
}
inline void so3_plus_jacobian(const double* x, double* jac /*3x3 row-major*/) {
    typedef Jet<3> J3;
    Vec3<J3> tmp(J3(x[0]) + J3(0.0, 0), J3(x[1]) + J3(0.0, 1), J3(x[2]) + J3(0.0, 2));
    lie::normalize_so3<J3>(tmp);
    for (int i = 0; i < 3; ++i) for (int k = 0; k < 3; ++k) jac[i * 3 + k] = tmp(i).v[k];
}

class Minimizer {
public:
    Problem& pr;
    Options opt;
    int n_amb = 0, n_tan = 0;
    std::vector<int> active_r;      // residual blocks that stay in the reduced program
    std::vector<double> x, cand;    // ambient values of ALL blocks (constant ones included), indexed by amb_off
    DMat H;                         // J^T J in tangent space (unscaled)
    std::vector<double> g;          // J^T r
    std::vector<double> scale;

    Minimizer(Problem& p, const Options& o) : pr(p), opt(o) { threads = o.num_threads; }

    void plus(const std::vector<double>& xin, const std::vector<double>& delta, std::vector<double>& xout) const {
        xout = xin;
        for (auto& b : pr.pblocks) {
            if (b.constant) continue;
            if (b.so3) so3_plus(&xin[b.amb_off], &delta[b.tan_off], &xout[b.amb_off]);
            else for (int k = 0; k < b.size; ++k) xout[b.amb_off + k] = xin[b.amb_off + k] + delta[b.tan_off + k];
        }
    }
    // cost (and optionally H, g) at ambient point xv.  eval_ok = false when a residual (or, with_jac, a Jacobian entry of a
    // non-constant block) is not finite: Ceres' ResidualBlock::Evaluate -> IsEvaluationValid rejects the evaluation then
    // (internal/ceres/residual_block.cc, array_utils.cc IsArrayValid).
    bool eval_ok = true;
    // threads > 1 (oracle_set_threads; the all-cores CPU leg of bench.py): the residual blocks are evaluated by an OpenMP team — what
    // Ceres does with Solver::Options::num_threads, which the reference leaves at 1 (src/factor/solver.cpp:798 is commented out) — and
    // their contributions are then added to H, g and the cost by ONE thread in the same order as the serial path: identical bits.
    int threads = 1;
    struct BlockEval { std::vector<double> res, jbuf; bool ok = true; };
    // one residual block at xv: residuals, (with_jac) Jacobian blocks in the tangent space of their parameter blocks, validity
    void eval_block(int ri, const std::vector<double>& xv, bool with_jac, BlockEval& be) const {
        const auto& rb = pr.rblocks[ri];
        const int nb = int(rb.blocks.size());
        const double* pp[16];
        double* jp[16];
        for (int b = 0; b < nb; ++b) pp[b] = &xv[pr.pblocks[rb.blocks[b]].amb_off];
        be.ok = true;
        be.res.assign(rb.n_res, 0.0);
        if (!with_jac) {
            rb.eval(pp, be.res.data(), nullptr);
        } else {
            int tot = 0;
            for (int b = 0; b < nb; ++b) tot += pr.pblocks[rb.blocks[b]].size;
            be.jbuf.assign(size_t(rb.n_res) * tot, 0.0);
            int o = 0;
            for (int b = 0; b < nb; ++b) { jp[b] = be.jbuf.data() + size_t(rb.n_res) * o; o += pr.pblocks[rb.blocks[b]].size; }
            rb.eval(pp, be.res.data(), jp);
            for (int b = 0; b < nb; ++b) {
                const auto& pb = pr.pblocks[rb.blocks[b]];
                if (pb.constant) continue;
                for (int e = 0; e < rb.n_res * pb.size; ++e) if (!std::isfinite(jp[b][e])) be.ok = false;
            }
            // local parameterisation: J <- J * dPlus/ddelta
            for (int b = 0; b < nb; ++b) {
                const auto& pb = pr.pblocks[rb.blocks[b]];
                if (pb.constant || !pb.so3) continue;
                double P[9];
                so3_plus_jacobian(pp[b], P);
                for (int i = 0; i < rb.n_res; ++i) {
                    double* row = jp[b] + i * 3;
                    double t0 = row[0] * P[0] + row[1] * P[3] + row[2] * P[6];
                    double t1 = row[0] * P[1] + row[1] * P[4] + row[2] * P[7];
                    double t2 = row[0] * P[2] + row[1] * P[5] + row[2] * P[8];
                    row[0] = t0; row[1] = t1; row[2] = t2;
                }
            }
        }
        for (int i = 0; i < rb.n_res; ++i) if (!std::isfinite(be.res[i])) be.ok = false;
    }
    // the block's share of H = J^T J, g = J^T r and sum r^2
    void accumulate_block(int ri, bool with_jac, BlockEval& be, double& cost) {
        const auto& rb = pr.rblocks[ri];
        const int nb = int(rb.blocks.size());
        if (with_jac) {
            double* jp[16];
            int o = 0;
            for (int b = 0; b < nb; ++b) { jp[b] = be.jbuf.data() + size_t(rb.n_res) * o; o += pr.pblocks[rb.blocks[b]].size; }
            for (int a = 0; a < nb; ++a) {
                auto& pa = pr.pblocks[rb.blocks[a]];
                if (pa.constant) continue;
                for (int ka = 0; ka < pa.size; ++ka) {
                    double gs = 0.0;
                    for (int i = 0; i < rb.n_res; ++i) gs += jp[a][i * pa.size + ka] * be.res[i];
                    g[pa.tan_off + ka] += gs;
                }
                for (int b = 0; b < nb; ++b) {
                    auto& pb = pr.pblocks[rb.blocks[b]];
                    if (pb.constant) continue;
                    for (int ka = 0; ka < pa.size; ++ka)
                        for (int kb = 0; kb < pb.size; ++kb) {
                            double s = 0.0;
                            for (int i = 0; i < rb.n_res; ++i) s += jp[a][i * pa.size + ka] * jp[b][i * pb.size + kb];
                            H(pa.tan_off + ka, pb.tan_off + kb) += s;
                        }
                }
            }
        }
        for (int i = 0; i < rb.n_res; ++i) cost += be.res[i] * be.res[i];
        if (!be.ok) eval_ok = false;
    }
    double evaluate(const std::vector<double>& xv, bool with_jac) {
        double cost = 0.0;
        eval_ok = true;
        if (with_jac) { H = DMat(n_tan, n_tan); g.assign(n_tan, 0.0); }
        const int nr = int(active_r.size());
        if (threads > 1 && nr > 1) {
            std::vector<BlockEval> all(nr);
#if defined(_OPENMP)
#pragma omp parallel for schedule(dynamic, 8) num_threads(threads)
#endif
            for (int k = 0; k < nr; ++k) eval_block(active_r[k], xv, with_jac, all[k]);
            for (int k = 0; k < nr; ++k) accumulate_block(active_r[k], with_jac, all[k], cost);
        } else {
            BlockEval be;
            for (int ri : active_r) {
                eval_block(ri, xv, with_jac, be);
                accumulate_block(ri, with_jac, be, cost);
            }
        }
        return 0.5 * cost;
    }

    void snapshot(IterationRecord& rec) const {
        rec.x.clear();
        for (auto& b : pr.pblocks) if (!b.constant) for (int k = 0; k < b.size; ++k) rec.x.push_back(x[b.amb_off + k]);
    }
    void write_back() const {
        for (auto& b : pr.pblocks) if (!b.constant) for (int k = 0; k < b.size; ++k) b.user[k] = x[b.amb_off + k];
    }

    void Solve(Summary* sum) {
        // ---- program reduction
        n_amb = n_tan = 0;
        for (auto& b : pr.pblocks) {
            b.amb_off = n_amb; n_amb += b.size;
            if (!b.constant) { b.tan_off = n_tan; n_tan += b.size; }
        }
        x.assign(n_amb, 0.0);
        for (auto& b : pr.pblocks) for (int k = 0; k < b.size; ++k) x[b.amb_off + k] = b.user[k];
        active_r.clear();
        std::vector<int> fixed_r;
        for (int ri = 0; ri < int(pr.rblocks.size()); ++ri) {
            bool any = false;
            for (int b : pr.rblocks[ri].blocks) if (!pr.pblocks[b].constant) any = true;
            (any ? active_r : fixed_r).push_back(ri);
        }
        {   // fixed cost (reported only)
            std::vector<int> keep = active_r;
            active_r = fixed_r;
            sum->fixed_cost = evaluate(x, false);
            active_r = keep;
        }
        sum->layout.clear();
        for (auto& b : pr.pblocks) if (!b.constant) sum->layout.push_back(std::make_pair(b.user, b.size));
        if (n_tan == 0) { sum->termination = 1; return; }

        // ---- iteration 0
        double radius = opt.initial_trust_region_radius;
        double decrease_factor = 2.0;
        bool reuse_diagonal = false;
        std::vector<double> diagonal(n_tan, 0.0);
        double x_cost = evaluate(x, true);
        sum->initial_cost = x_cost;
        // TrustRegionMinimizer::IterationZero -> EvaluateGradientAndJacobian fails: "Residual and Jacobian evaluation failed.",
        // termination FAILURE, no iteration recorded; Solver then hands the ORIGINAL parameters back (see restore_user below)
        if (!eval_ok) { sum->termination = 6; sum->final_cost = x_cost; sum->num_iterations = 0; return; }
        const std::vector<double> x_initial = x;
        // solver.cc Minimize(): StateVectorToParameterBlocks(IsSolutionUsable() ? reduced_parameters : original_reduced_parameters):
        // a FAILURE termination (evaluation failure after a successful step, or max_num_consecutive_invalid_steps) is not "usable"
        // and restores the parameters the solve started from, whatever progress was written back before
        auto restore_user = [&]() { x = x_initial; write_back(); };
        scale.assign(n_tan, 1.0);
        if (opt.jacobi_scaling)
            for (int i = 0; i < n_tan; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(H(i, i)));
        double minimum_cost = x_cost;
        double x_norm = 0.0;
        for (auto& b : pr.pblocks) if (!b.constant) for (int k = 0; k < b.size; ++k) x_norm += x[b.amb_off + k] * x[b.amb_off + k];
        x_norm = std::sqrt(x_norm);
        auto gradient_max_norm = [&]() {
            std::vector<double> ng(n_tan), xp;
            for (int i = 0; i < n_tan; ++i) ng[i] = -g[i];
            plus(x, ng, xp);
            double m = 0.0;
            for (auto& b : pr.pblocks) if (!b.constant) for (int k = 0; k < b.size; ++k) m = std::max(m, std::fabs(x[b.amb_off + k] - xp[b.amb_off + k]));
            return m;
        };
        double gmax = gradient_max_norm();
        {
            IterationRecord rec{};
            rec.iteration = 0; rec.cost = x_cost; rec.radius = radius; rec.step_is_valid = true; rec.step_is_successful = true;
            snapshot(rec);
            sum->iterations.push_back(rec);
        }
        sum->final_cost = x_cost;
        if (gmax <= opt.gradient_tolerance) { sum->termination = 1; return; }

        int num_consecutive_invalid_steps = 0;
        int iteration = 0;
        bool last_successful = true;
        while (true) {
            // FinalizeIterationAndCheckIfMinimizerCanContinue
            if (iteration >= opt.max_num_iterations) { sum->termination = 4; break; }
            if (iteration > 0 && last_successful && gmax <= opt.gradient_tolerance) { sum->termination = 1; break; }
            if (!(radius > opt.min_trust_region_radius)) { sum->termination = 5; break; }
            ++iteration;
            IterationRecord rec{};
            rec.iteration = iteration;

            // ---- ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep)
            DMat A(n_tan, n_tan);
            std::vector<double> gs(n_tan);
            for (int i = 0; i < n_tan; ++i) {
                gs[i] = g[i] * scale[i];
                for (int j = 0; j < n_tan; ++j) A(i, j) = H(i, j) * scale[i] * scale[j];
            }
            if (!reuse_diagonal)
                for (int i = 0; i < n_tan; ++i) diagonal[i] = std::min(std::max(A(i, i), opt.min_lm_diagonal), opt.max_lm_diagonal);
            DMat Ad = A;
            for (int i = 0; i < n_tan; ++i) {
                const double lm = std::sqrt(diagonal[i] / radius);
                Ad(i, i) += lm * lm;
            }
            reuse_diagonal = true;
            DMat Lc;
            std::vector<double> step = gs;
            bool solved = llt_lower(Ad, Lc);
            if (solved) {
                llt_solve(Lc, step);
                for (int i = 0; i < n_tan; ++i) { if (!std::isfinite(step[i])) solved = false; step[i] = -step[i]; }
            }
            double model_cost_change = 0.0;
            bool valid = false;
            if (solved) {
                // -(J s)'(r + J s/2) = -(s'g_s + s'A s/2)
                double sg = 0.0, sAs = 0.0;
                for (int i = 0; i < n_tan; ++i) {
                    sg += step[i] * gs[i];
                    double t = 0.0;
                    for (int j = 0; j < n_tan; ++j) t += A(i, j) * step[j];
                    sAs += step[i] * t;
                }
                model_cost_change = -(sg + 0.5 * sAs);
                valid = model_cost_change > 0.0;
            }
            rec.model_cost_change = model_cost_change;
            rec.step_is_valid = valid;
            if (!valid) {
                // HandleInvalidStep
                if (++num_consecutive_invalid_steps >= opt.max_num_consecutive_invalid_steps) { sum->termination = 6; --iteration; restore_user(); x_cost = sum->initial_cost; break; }
                radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;   // StepIsInvalid -> StepRejected(0)
                rec.cost = x_cost; rec.radius = radius; rec.step_is_successful = false;
                snapshot(rec);
                sum->iterations.push_back(rec);
                last_successful = false;
                continue;
            }
            num_consecutive_invalid_steps = 0;
            std::vector<double> delta(n_tan);
            for (int i = 0; i < n_tan; ++i) delta[i] = step[i] * scale[i];

            // ---- ComputeCandidatePointAndEvaluateCost
            plus(x, delta, cand);
            double candidate_cost = evaluate(cand, false);
            rec.candidate_cost = candidate_cost;
            // "Step failed to evaluate. Treating it as a step with infinite cost" (a non-finite residual fails the evaluation)
            if (!eval_ok || !std::isfinite(candidate_cost)) candidate_cost = std::numeric_limits<double>::max();

            // ---- ParameterToleranceReached
            double step_norm = 0.0;
            for (auto& b : pr.pblocks) if (!b.constant) for (int k = 0; k < b.size; ++k) {
                const double d = x[b.amb_off + k] - cand[b.amb_off + k];
                step_norm += d * d;
            }
            step_norm = std::sqrt(step_norm);
            if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance)) {
                rec.cost = x_cost; rec.radius = radius; rec.step_is_successful = false;
                snapshot(rec); sum->iterations.push_back(rec);
                sum->termination = 3; break;
            }
            // ---- FunctionToleranceReached
            if (std::fabs(x_cost - candidate_cost) <= opt.function_tolerance * x_cost) {
                rec.cost = x_cost; rec.radius = radius; rec.step_is_successful = false;
                snapshot(rec); sum->iterations.push_back(rec);
                sum->termination = 2; break;
            }
            // ---- IsStepSuccessful
            const double relative_decrease = (x_cost - candidate_cost) / model_cost_change;
            rec.relative_decrease = relative_decrease;
            if (relative_decrease > opt.min_relative_decrease) {
                // HandleSuccessfulStep
                x = cand;
                x_norm = 0.0;
                for (auto& b : pr.pblocks) if (!b.constant) for (int k = 0; k < b.size; ++k) x_norm += x[b.amb_off + k] * x[b.amb_off + k];
                x_norm = std::sqrt(x_norm);
                x_cost = evaluate(x, true);
                // HandleSuccessfulStep -> EvaluateGradientAndJacobian fails: FAILURE, returned before the iteration is recorded
                if (!eval_ok) { sum->termination = 6; --iteration; restore_user(); x_cost = sum->initial_cost; break; }
                gmax = gradient_max_norm();
                radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
                radius = std::min(opt.max_trust_region_radius, radius);
                decrease_factor = 2.0;
                reuse_diagonal = false;
                last_successful = true;
                ++sum->num_successful_steps;
                if (x_cost < minimum_cost) { minimum_cost = x_cost; write_back(); }
            } else {
                radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
                last_successful = false;
            }
            rec.cost = x_cost; rec.radius = radius; rec.step_is_successful = last_successful;
            snapshot(rec);
            sum->iterations.push_back(rec);
        }
        sum->num_iterations = iteration;
        sum->final_cost = x_cost;
    }
};

}  // namespace miniceres
}  // namespace oracle
