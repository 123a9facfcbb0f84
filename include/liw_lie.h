/* liw_lie.h — the SE3 / SO3 helpers of reference src/utilies/common.h:119-181 (lie::exp_so3, log_SO3, make_tf, log_SE3)
 * on plain arrays, exported so that host code above the C ABI (include/lvio_2d_trajectory.hpp) composes poses with exactly
 * the arithmetic the library and its kernels use.  A transform is T12 = R (9, row-major) then t (3). */
#ifndef LIW_LIE_H
#define LIW_LIE_H
#ifdef __cplusplus
extern "C" {
#endif
void liw_lie_exp_so3(const double* so3, double* R9);
void liw_lie_log_SO3(const double* R9, double* so3);
void liw_lie_make_tf(const double* p3, const double* so3, double* T12);
void liw_lie_log_SE3(const double* T12, double* p3, double* so3);
void liw_lie_mul(const double* A12, const double* B12, double* C12);      /* C = A * B */
void liw_lie_inverse(const double* A12, double* B12);
void liw_lie_apply(const double* T12, const double* x3, double* y3);       /* y = T * x */
/* extrinsic of liw_params (4x4 row-major) -> T12, re-orthonormalised when normalize != 0 (params.cpp:44-54) */
void liw_lie_from_matrix16(const double* M16, int normalize, double* T12);
#ifdef __cplusplus
}
#endif
#endif
