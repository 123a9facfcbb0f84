// k_lm_common.hpp — device helpers shared by the LM step kernels (k_lm.hip, k_lm_quad.hip): Ceres' trust-region constants, DPP wave /
// row reductions, the so3 local parameterisation (reference src/factor/factor_common.h:37-60), the constant-block test of the TRACK
// topology (solver.cpp:787-794).
#pragma once
#include "liw_kernels.hpp"

namespace liw {

constexpr double kMinDiag = 1e-6, kMaxDiag = 1e32, kMinRelDec = 1e-3, kFuncTol = 1e-6, kGradTol = 1e-10, kParamTol = 1e-8;
constexpr double kMaxRadius = 1e16, kMinRadius = 1e-32, kInitRadius = 1e4;
constexpr double kPi = 3.141592653589793238462643383279, kTwoPi = 6.283185307179586476925286766559;

// Cross-lane reductions on DPP row operations (v_mov_b32_dpp on both halves of the double): a row of 16 lanes in 4 steps, the four
// rows joined by row_bcast:15 / :31, result broadcast from lane 63 — 175 cycles measured (tools/ubench/dpp.hip) against 460 for six
// ds_bpermute butterflies; the step kernels are single waves whose run time is the sum of such latencies.
template <int CTRL, int ROWMASK = 0xF>
__device__ __forceinline__ double dpp64(double v, double old = 0.0) {   // lanes without a source / outside ROWMASK: `old`
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(__double2loint(old), lo, CTRL, ROWMASK, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(__double2hiint(old), hi, CTRL, ROWMASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum(double v) {   // every lane: the sum over its row of 16 lanes
    v += dpp64<0xB1>(v); v += dpp64<0x4E>(v); v += dpp64<0x141>(v); v += dpp64<0x140>(v);   // quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
    return v;
}
__device__ __forceinline__ double rdlane(double v, int l);
__device__ __forceinline__ double wave_sum(double v) {
    v = row_sum(v);
    v += dpp64<0x142, 0xA>(v);   // row_bcast:15 into rows 1 and 3
    v += dpp64<0x143, 0xC>(v);   // row_bcast:31 into rows 2 and 3
    return rdlane(v, 63);
}
__device__ __forceinline__ double wave_max(double v) {
    v = fmax(v, dpp64<0xB1>(v, v)); v = fmax(v, dpp64<0x4E>(v, v)); v = fmax(v, dpp64<0x141>(v, v)); v = fmax(v, dpp64<0x140>(v, v));
    v = fmax(v, dpp64<0x142, 0xA>(v, v));
    v = fmax(v, dpp64<0x143, 0xC>(v, v));
    return rdlane(v, 63);
}

__device__ __forceinline__ double rdlane(double v, int l) {   // v_readlane_b32 x2: broadcast lane l's value (l uniform)
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// so3 Plus and its Jacobian at delta = 0 (factor_common.h:41-53 through AutoDiffLocalParameterization)
__device__ __forceinline__ void so3_plus(const double* x, const double* d, double* out) {
    const double a0 = x[0] + d[0], a1 = x[1] + d[1], a2 = x[2] + d[2];
    // normalize_so3 returns its argument unchanged unless |a| > pi: clearly below that (|a|^2 < 9.8 < pi^2 = 9.8696) the
    // square root of the norm is skipped; the callers pass wave-uniform values, so the branch does not diverge
    if (a0 * a0 + a1 * a1 + a2 * a2 < 9.8) { out[0] = a0; out[1] = a1; out[2] = a2; return; }
    V3<double> r = normalize_so3(V3<double>(a0, a1, a2));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
__device__ __forceinline__ bool so3_plus_jac(const double* x, double* P9) {
    const double a = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (!(a > kPi)) return false;  // identity
    const double k = floor((a + kPi) / kTwoPi);
    const double c = kTwoPi * k / a;
    const double u[3] = {x[0] / a, x[1] / a, x[2] / a};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) P9[i * 3 + j] = (i == j ? 1.0 : 0.0) - c * ((i == j ? 1.0 : 0.0) - u[i] * u[j]);
    return true;
}

__device__ __forceinline__ bool var_is_const(int mode, int fast, int n, int i, int v) {
    if (mode != LIW_MODE_TRACK) return false;
    if (i >= n - 1) return false;
    return v < 6 || (fast && v >= 9);
}

// Is this window outside what the quad kernel handles in this launch?  Pure function of memory at launch start, evaluated identically
// at the start of k_lm_step_quad (which marks the windows it takes in LmState::pad_ for the one-wave kernel launched behind it): a rotation vector with |theta|^2 > 9.6 (pi^2 = 9.87) among the current
// states or, when a candidate is pending, the candidate states.
__device__ __forceinline__ bool quad_slow_lane(const double* xw, const double* xc, int have_cand, int n, int first, int stride) {
    bool slow = false;
    for (int i = first; i < n; i += stride) {
        const double* q = xw + (size_t)i * 15 + 3;
        slow = slow || !(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] <= 9.6);
        if (have_cand) {
            const double* qc = xc + (size_t)i * 15 + 3;
            slow = slow || !(qc[0] * qc[0] + qc[1] * qc[1] + qc[2] * qc[2] <= 9.6);
        }
    }
    return slow;
}
}  // namespace liw
