/* liw_posegraph.h — C ABI of the back-end pose-graph relinearisation.
 *
 * Stands in for keyframe_manager::solve (reference src/trajectory/keyframe_manager.cpp:722-838): Levenberg-Marquardt over
 * the key-frame poses with sequential edges (weight 1), loop edges (weight loop_edge_k), optional ground factors, the
 * first pose of the first sequential edge held constant, so3 local parameterisation on every rotation vector.
 *   edge_factor / edge_noise   reference src/factor/edge_factor.h:79-126, :4-26 (incl. the J(1,2) entry of :19)
 *   ground_factor_p / _q       reference src/factor/ground_factor.h:27-82 (one block each per key frame, :790-811)
 * On the MI355X: one 16-lane group per edge evaluates residual + Jacobian (dual numbers, direction per lane), the normal
 * equations are assembled as a dense fp64 matrix in HBM (6 unknowns per key frame) and factorised by a blocked right-
 * looking Cholesky whose trailing update runs on the fp64 matrix cores; the trust-region bookkeeping runs on the host.
 * Loop DETECTION (keyframe_manager.cpp:642-712, :945-1183) is not part of this library.
 */
#ifndef LIW_POSEGRAPH_H
#define LIW_POSEGRAPH_H
#include "liw_window.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct liw_pg_params {
    double loop_sigma_p[3], loop_sigma_q[3];   /* config/office.yaml:110-111 */
    double loop_edge_k;                        /* :106 */
    int use_ground_p_factor, use_ground_q_factor;   /* :114-115 */
} liw_pg_params;

/* poses [N][6] = p, q (rotation vector) of the key frames, in/out (host).  seq_idx / loop_idx [.][2] = index1, index2;
 * seq_tf12 / loop_tf12 [.][12] = R (row-major 9) then t of the measured tf12.  max_iters <= 0: Ceres default 50.
 * Returns LIW_OK and fills `summary` (iterations, successful steps, termination, initial / final cost). */
int liw_posegraph_solve(liw_ctx* ctx, const liw_pg_params* pg, int N, double* poses, int n_seq, const int* seq_idx, const double* seq_tf12,
                        int n_loop, const int* loop_idx, const double* loop_tf12, int max_iters, liw_summary* summary);
/* the tangent-space normal equations at `poses` (tests): H [6N][6N] dense symmetric, g [6N] = J^T r, cost = 1/2 |r|^2
 * (any output may be NULL); the constant key frame has an identity block and a zero gradient */
int liw_posegraph_linearize(liw_ctx* ctx, const liw_pg_params* pg, int N, const double* poses, int n_seq, const int* seq_idx, const double* seq_tf12,
                            int n_loop, const int* loop_idx, const double* loop_tf12, double* H, double* g, double* cost);
/* the dense SPD solver underneath (blocked Cholesky + triangular solves on the device), for tests and other callers:
 * A [n][n] row-major host (only the lower triangle is read), b [n] -> x [n].  Returns LIW_ESTATE if A is not positive definite. */
int liw_dense_spd_solve(liw_ctx* ctx, int n, const double* A, const double* b, double* x);

#ifdef __cplusplus
}
#endif
#endif
