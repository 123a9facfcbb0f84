// k_linearize.hip — factor residual/Jacobian evaluation + J^T J block partial sums (gfx950, fp64).
//
// One linearisation = three role kernels over every residual block of every window of a batch (or ONE kernel for small
// batches), one wavefront (64-thread work-group) per work item; the roles write disjoint partial slots and run concurrently:
//   k_frame_tf  : per (window, frame): rows 0,1 of make_tf(p,theta) * T_imu_to_laser and d/dtheta_k (dual numbers,
//                 direction-per-lane) — every exp_so3 of the laser path is hoisted out of the blocks (k_lin_all: done by
//                 the laser waves themselves).
//   k_lin_laser : G (window, owning frame) groups per wave; a LANE is one laser_factor block (reference
//                 src/factor/laser_factor.h:45-89, two point-to-line rows), 64 blocks per pass, coalesced reads of the
//                 component-major end-point arrays, closed-form Jacobian, register accumulation of the pair products
//                 and ONE butterfly reduction per group.
//   k_lin_imu   : six imu_factor blocks per wave (src/factor/imu_factor.h:13-89): 9 dual directions + value lane per
//                 block, closed forms for the linear columns, whitening and Y^T Y on the fp64 matrix cores.
//   k_lin_small : wheel_odom_factor (src/factor/wheel_factor.h:12-73, six blocks per wave, 9 dual directions) and
//                 ground_factor_p/q (src/factor/ground_factor.h:27-82, eight frames per wave); the n-fold duplication
//                 of solver.cpp:142-159 is applied as an integer weight n.
// Every role reduces G = Y^T Y with Y = [J | r] deterministically (no atomics).  G blocks go to the partial-sum slots
// described in liw_kernels.hpp; k_lm.hip assembles them.  Jacobians here are w.r.t. the AMBIENT parameters, exactly like
// auto_diff::compute_res_and_jacobi (src/utilies/common.h:201-217); the so3 local parameterisation is applied at assembly.
#include "liw_kernels.hpp"

namespace liw {

// ------------------------------------------------------------------------------------------- laser
// Frame transform record (FTF doubles): rows 0,1 of  make_tf(p,theta) * T_imu_to_laser  and d/dtheta_k, k = 0..2:
//   [0..5] M[2][3]   [6..7] t[2]   [8+6k .. 8+6k+5] dM_k[2][3]   [26+2k .. 26+2k+1] dt_k[2]
// Two records per (window, frame): slot 0 at the frame's own pose, slot 1 at the constant laser_match pose (p1,q1) the
// tracking / marginalisation topologies tie the frame to.  4 lanes per record: 3 derivative directions + the value.
__device__ __forceinline__ void frame_tf_record(const DevParams& P, const double* pose6, int dir, double* o) {
    V3<LJ> p = cast_v3<LJ>(pose6);
    V3<LJ> th(LJ(pose6[3], dir == 0 ? 1.0 : 0.0), LJ(pose6[4], dir == 1 ? 1.0 : 0.0), LJ(pose6[5], dir == 2 ? 1.0 : 0.0));
    Iso<LJ> Twl = mul(make_tf(p, th), cast_iso<LJ>(P.Ril, P.til));
    const LJ tt[2] = {Twl.t.x, Twl.t.y};
    if (dir == 3) {
        for (int r = 0; r < 2; ++r) { for (int c = 0; c < 3; ++c) o[r * 3 + c] = Twl.R(r, c).v; o[6 + r] = tt[r].v; }
    } else {
        for (int r = 0; r < 2; ++r) { for (int c = 0; c < 3; ++c) o[8 + 6 * dir + r * 3 + c] = Twl.R(r, c).d; o[26 + 2 * dir + r] = tt[r].d; }
    }
}
// large batches: every record once, ahead of the laser kernel (k_lin_all computes its records in the laser waves instead)
__global__ void k_frame_tf(int B, int n, const double* x, const double* match_pose, double* ftf, DevParams P, const LmState* lm) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int dir = t & 3, rec = t >> 2;
    if (rec >= B * n * 2) return;
    const int slot = rec & 1, fi = rec >> 1;
    if (lm && lm[fi / n].done) return;
    frame_tf_record(P, slot ? (match_pose + (size_t)fi * 12) : (x + (size_t)fi * 15), dir, ftf + (size_t)rec * FTF);
}

// Wave-wide sums of V per-lane values (V = 32 or 16) in V-1 pair exchanges + log2(64/V) plain steps: each step pairs
// value j with value j + V/2 across the lane bit BIT, halving the values a lane still owns.  Afterwards every lane
// holds the total of value  lane >> log2(64/V).
template <int V, int BIT>
__device__ __forceinline__ void bfly_step(double* v, int lane) {
    if constexpr (V > 1) {
        const bool up = (lane & BIT) != 0;
#pragma unroll
        for (int j = 0; j < V / 2; ++j) {
            const double send = up ? v[j] : v[j + V / 2];
            const double keep = up ? v[j + V / 2] : v[j];
            v[j] = keep + __shfl_xor(send, BIT, 64);
        }
        if constexpr (BIT > 1) bfly_step<V / 2, BIT / 2>(v, lane);
    } else {
        v[0] += __shfl_xor(v[0], BIT, 64);
        if constexpr (BIT > 1) bfly_step<1, BIT / 2>(v, lane);
    }
}

#ifdef LIW_CLK
__device__ long long g_clk_lin[512];
// stamps of one wave in the middle of the grid (so that it runs under load), every memory operation drained first
#define LSTAMP(id) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (blockIdx.x == gridDim.x / 2 && lane == 0 && (id) < 512) g_clk_lin[(id)] = clock64(); } while (0)
#else
#define LSTAMP(id) do { } while (0)
#endif
constexpr int LASER_GMAX = 8;   // (window, frame) groups one wave may own

// Laser group kernel.  Columns of a block's two Jacobian rows: the translation columns of the two poses differ only
// in sign (d s/d p_b = - d s/d p_a), so the unique columns are  BOTH: [a_x a_y a_th0..2 b_th0..2 r] (9, 45 pairs),
// one free pose: [b_x b_y b_th0..2 r] (6, 21 pairs).  Every lane accumulates its blocks' pair products in
// registers over all passes; one butterfly per group reduces them across the wave.
template <bool BOTH>
__device__ void laser_wave_local(const LinArgs& A, const DevParams& P, int G, int vblock) {
#define LASER_LOCAL_TF 1
#include "k_lin_laser_body.inc"
#undef LASER_LOCAL_TF
}
template <bool BOTH>
__global__ __launch_bounds__(64, 2) void k_lin_laser(LinArgs A, DevParams P, int G) {
    const int vblock = (int)blockIdx.x;
#define LASER_LOCAL_TF 0
#include "k_lin_laser_body.inc"
#undef LASER_LOCAL_TF
}

// ------------------------------------------------------------------------------------------- imu
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int IMU_XS = 31, IMU_X = 15 * IMU_XS;   // LDS tile of one block: [J_raw | r_raw], 15 rows x 31 columns
constexpr int IMU_PER_WAVE = 6;   // 10 lanes per block: 9 derivative directions (theta_i, theta_j, bw_i) + the value lane

// IMU role.  Only the rotation vectors and the gyro bias enter the residual non-linearly, so the dual-number pass
// carries 9 directions per block (6 blocks per wave); the remaining 21 Jacobian columns are closed forms of R_i^T,
// Dt and the pre-integration Jacobian blocks.  The whitening  Y = sqrt_info [J_raw | r_raw]  (imu_factor.h:85-86,
// dense 15x15) and the normal-equation block  G = Y^T Y  run on the matrix cores: 8 + 12 v_mfma_f64_16x16x4_f64 per
// block, the accumulator layout of Y being directly the operand layout of Y^T Y.
__device__ void imu_group(const LinArgs& A, const DevParams& P, int b, int item, int sel, double* lds) {
    const int lane = threadIdx.x & 63, grp = lane / 10, d = lane % 10;
    const int n = A.n, k0 = IMU_PER_WAVE * item;
    const int k = k0 + grp;
    const bool on = grp < IMU_PER_WAVE && k < n - 1;
    // [15][31] per block: [J_raw wrt x_i (15) | r_raw | J_raw wrt x_j (15)] — the residual sits in column 15 so that the three
    // 16x16 tiles of G = Y^T Y are exactly the blocks ii (+ gradient_i, cost), ij (+ gradient_j) and jj
    constexpr int XR = 15, XJ = 16;
    double* Xg = lds + (grp < IMU_PER_WAVE ? grp : 0) * IMU_X;
    // whitening-matrix operands of every block of this wave, fetched up front (one memory round trip, hidden behind
    // the dual-number pass):  sop[g][c] = A[i = lane & 15][k = (lane >> 4) + 4c] = sqrt_info_g[i][k]
    double sop[IMU_PER_WAVE][4];
    {
        const int ml = lane & 15, mk = lane >> 4;
#pragma unroll
        for (int g = 0; g < IMU_PER_WAVE; ++g) {
            const bool gon = k0 + g < n - 1;
            const double* S = A.imu_sqrtP + ((size_t)b * (n - 1) + (gon ? k0 + g : 0)) * 225;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int kk = mk + 4 * c;
                const bool in = gon && ml < 15 && kk < 15;
                const double v = S[in ? ml * 15 + kk : 0];
                sop[g][c] = in ? v : 0.0;
            }
        }
    }
    LSTAMP(300);
    for (int e = lane; e < IMU_PER_WAVE * IMU_X; e += 64) lds[e] = 0.0;
    __syncthreads();
    LSTAMP(301);
    const size_t fk = (size_t)b * (n - 1) + (on ? k : 0);
    if (on) {
        const double* si_ = A.x + ((size_t)b * n + k) * 15;
        const double* sj_ = si_ + 15;
        const double* Jp = A.imu_J + fk * 225;
        const double Dt = A.imu_Dt[fk];
        // imu_factor::operator() (imu_factor.h:41-83) evaluated stage by stage; every stage writes its rows of
        // [J_raw | r_raw] to LDS at once so its temporaries die (keeps the kernel at 2 waves per SIMD)
        const double* X0 = A.imu_X + fk * 15;
        const int col = d < 3 ? 3 + d : (d < 6 ? XJ + 3 + (d - 3) : (d < 9 ? 12 + (d - 6) : XR));
        auto put3 = [&](int row0, const V3<LJ>& v) {
            Xg[(row0 + 0) * IMU_XS + col] = d < 9 ? v.x.d : v.x.v;
            Xg[(row0 + 1) * IMU_XS + col] = d < 9 ? v.y.d : v.y.v;
            Xg[(row0 + 2) * IMU_XS + col] = d < 9 ? v.z.d : v.z.v;
        };
        auto blk = [&](int ro, int co) {
            M3<LJ> m;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) m(r, c) = LJ(Jp[(ro + r) * 15 + co + c]);
            return m;
        };
        const V3<LJ> thetai(LJ(si_[3], d == 0 ? 1.0 : 0.0), LJ(si_[4], d == 1 ? 1.0 : 0.0), LJ(si_[5], d == 2 ? 1.0 : 0.0));
        const V3<LJ> thetaj(LJ(sj_[3], d == 3 ? 1.0 : 0.0), LJ(sj_[4], d == 4 ? 1.0 : 0.0), LJ(sj_[5], d == 5 ? 1.0 : 0.0));
        const V3<LJ> bwi(LJ(si_[12], d == 6 ? 1.0 : 0.0), LJ(si_[13], d == 7 ? 1.0 : 0.0), LJ(si_[14], d == 8 ? 1.0 : 0.0));
        const V3<LJ> dbw = bwi - cast_v3<LJ>(X0 + 12);
        const V3<LJ> dba = cast_v3<LJ>(si_ + 9) - cast_v3<LJ>(X0 + 9);
        const LJ g_norm(P.g), DtJ(Dt);
        const V3<LJ> gdir(LJ(0.0), LJ(0.0), LJ(1.0));
        const M3<LJ> Rt = exp_so3(-thetai);                                        // bk_R_w
        {   // alpha, beta rows
            const V3<LJ> pi = cast_v3<LJ>(si_), vi = cast_v3<LJ>(si_ + 6), pj = cast_v3<LJ>(sj_), vj = cast_v3<LJ>(sj_ + 6);
            const V3<LJ> alpha = cast_v3<LJ>(X0) + mul(blk(0, 9), dba) + mul(blk(0, 12), dbw);
            put3(0, alpha - mul(Rt, pj - pi + ((gdir * LJ(0.5)) * g_norm) * DtJ * DtJ - vi * DtJ));
            const V3<LJ> beta = cast_v3<LJ>(X0 + 3) + mul(blk(3, 9), dba) + mul(blk(3, 12), dbw);
            put3(3, beta - mul(Rt, vj + (gdir * g_norm) * DtJ - vi));
            put3(9, cast_v3<LJ>(sj_ + 9) - cast_v3<LJ>(si_ + 9));                  // res_ba
            put3(12, cast_v3<LJ>(sj_ + 12) - bwi);                                  // res_bw
        }
        LSTAMP(302);
        // closed-form columns (value parts are uniform over the block's lanes; each lane writes one column group)
        if (d == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Xg[r * IMU_XS + c] = Rt(r, c).v;                      // d r_alpha / d p_i
        } else if (d == 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) { Xg[r * IMU_XS + 6 + c] = Rt(r, c).v * Dt; Xg[(3 + r) * IMU_XS + 6 + c] = Rt(r, c).v; }   // d/d v_i
        } else if (d == 2) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    Xg[r * IMU_XS + 9 + c] = Jp[r * 15 + 9 + c];               // alpha_J_ba
                    Xg[(3 + r) * IMU_XS + 9 + c] = Jp[(3 + r) * 15 + 9 + c];   // beta_J_ba
                    Xg[(9 + r) * IMU_XS + 9 + c] = r == c ? -1.0 : 0.0;        // d r_ba / d ba_i
                }
        } else if (d == 3) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Xg[r * IMU_XS + XJ + c] = -Rt(r, c).v;                // d r_alpha / d p_j
        } else if (d == 4) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) Xg[(3 + r) * IMU_XS + XJ + 6 + c] = -Rt(r, c).v;      // d r_beta / d v_j
        } else if (d == 5) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { Xg[(9 + r) * IMU_XS + XJ + 9 + r] = 1.0; Xg[(12 + r) * IMU_XS + XJ + 12 + r] = 1.0; }   // d r_ba/d ba_j, d r_bw/d bw_j
        }
        LSTAMP(303);
        __builtin_amdgcn_sched_barrier(0);
        {   // gamma rows: log( exp(-gamma^) R_i^T R_j )
            const M3<LJ> RiRj = mul(Rt, exp_so3(thetaj));
            __builtin_amdgcn_sched_barrier(0);
            const V3<LJ> gamma = cast_v3<LJ>(X0 + 6) + mul(blk(6, 12), dbw);
            const M3<LJ> E = mul(exp_so3(-gamma), RiRj);
            __builtin_amdgcn_sched_barrier(0);
            put3(6, log_SO3(E));
        }
    }
    __syncthreads();
    LSTAMP(304);
    // ---- matrix-core part, one block at a time (the whole wave cooperates)
    const int ml = lane & 15, mk = lane >> 4;
#pragma unroll
    for (int g = 0; g < IMU_PER_WAVE; ++g) {
        const int kg = k0 + g;
        if (kg >= n - 1) break;
        const size_t fg = (size_t)b * (n - 1) + kg;
        const double* X = lds + g * IMU_X;
        d4 y0 = {0.0, 0.0, 0.0, 0.0}, y1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int kk = mk + 4 * c;
            const double a = sop[g][c];                                        // A[i = ml][k = kk] = sqrt_info
            // row 15 and column 31 of the 16x32 operand are zero padding: not stored (15 x 31 doubles per block keep 7 waves per CU in LDS)
            const double x0v = kk < 15 ? X[kk * IMU_XS + ml] : 0.0;
            const double x1v = (kk < 15 && ml < 15) ? X[kk * IMU_XS + 16 + ml] : 0.0;
            y0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x0v, y0, 0, 0, 0);
            y1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, x1v, y1, 0, 0, 0);
        }
        // y_t[r] = Y[mk + 4r][ml + 16t]  ==  operand chunk r of Y^T Y
        d4 g00 = {0.0, 0.0, 0.0, 0.0}, g01 = g00, g11 = g00;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            g00 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[c], y0[c], g00, 0, 0, 0);
            g01 = __builtin_amdgcn_mfma_f64_16x16x4f64(y0[c], y1[c], g01, 0, 0, 0);
            g11 = __builtin_amdgcn_mfma_f64_16x16x4f64(y1[c], y1[c], g11, 0, 0, 0);
        }
        LSTAMP(310 + 2 * g);
        double* out = A.PI[sel] + fg * PIS;
        // tile (0,0) = [ii | gradient_i ; . | cost], tile (0,1) = [ij ; gradient_j], tile (1,1) = jj: one masked store per tile row group
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mk + 4 * r;
            if (row < 15 && ml < 15) {
                out[PI_II + row * 15 + ml] = g00[r];
                out[PI_IJ + row * 15 + ml] = g01[r];
                out[PI_JJ + row * 15 + ml] = g11[r];
            }
            if (row < 15 && ml == 15) out[PI_G + row] = g00[r];
            if (r == 3) {
                if (row == 15 && ml < 15) out[PI_G + 15 + ml] = g01[r];
                if (row == 15 && ml == 15) out[PI_C] = g00[r];
            }
            if (A.dbg_imu_res && ml == 15 && row < 15) A.dbg_imu_res[fg * 15 + row] = y0[r];
            if (A.dbg_imu_jac && row < 15 && ml < 15) {
                A.dbg_imu_jac[(fg * 15 + row) * 30 + ml] = y0[r];
                A.dbg_imu_jac[(fg * 15 + row) * 30 + 15 + ml] = y1[r];
            }
        }
        LSTAMP(311 + 2 * g);
    }
}

// ------------------------------------------------------------------------------------------- wheel
// wheel_odom_factor::operator(), src/factor/wheel_factor.h:12-73
template <class T>
__device__ __forceinline__ void wheel_res(const DevParams& P, const double* T12, const double* sq9, const T* pi_, const T* qi_, const T* pj_, const T* qj_, T* res,
                                          const T* dp, M3<T>* R_w_wheel_i) {
    V3<T> pi(pi_[0], pi_[1], pi_[2]), thetai(qi_[0], qi_[1], qi_[2]), pj(pj_[0], pj_[1], pj_[2]), thetaj(qj_[0], qj_[1], qj_[2]);
    Iso<T> T_i_w = cast_iso<T>(P.Riw, P.tiw);
    Iso<T> tf_i = mul(make_tf(pi, thetai), T_i_w);
    Iso<T> tf_j = mul(make_tf(pj, thetaj), T_i_w);
    Iso<T> w_tf_ij = mul(inverse(tf_i), tf_j);
    // dp: perturbation of the relative translation (the positions enter only through R_wi^T (p_j - p_i), see wheel_hex)
    V3<T> p = w_tf_ij.t + V3<T>(dp[0], dp[1], dp[2]), q = log_SO3(w_tf_ij.R);
    *R_w_wheel_i = tf_i.R;
    // log_SE3 of the constant odometry increment: no parameter enters, so it is evaluated on plain doubles
    const V3<double> oqd = log_SO3(cast_m3<double>(T12));
    const V3<T> op = cast_v3<T>(T12 + 9);
    const V3<T> oq = V3<T>(T(oqd.x), T(oqd.y), T(oqd.z));
    T o_len = dsqrt(op.x * op.x + op.y * op.y);
    T len = dsqrt(p.x * p.x + p.y * p.y);
    V3<T> o_dir(op.x, op.y, T(0.0)), dir(p.x, p.y, T(0.0));
    T angle(0.0);
    if (norm(o_dir) > T(0.0001) && norm(dir) > T(0.0001)) {
        o_dir = normalized(o_dir);
        dir = normalized(dir);
        T sinn = norm(cross(o_dir, dir));
        angle = dasin(sinn);
    } else {
        angle = norm(dir);
    }
    if (len < T(0.0001) || o_len < T(0.0001)) res[0] = T(sq9[0]) * len;
    else res[0] = T(sq9[0]) * (o_len - len);
    res[1] = T(sq9[4]) * angle;
    if (norm(q) < T(0.001) || norm(oq) < T(0.001)) res[2] = T(sq9[8]) * norm(q);
    else res[2] = T(sq9[8]) * (norm(oq) - norm(q));
}

// Six blocks per wave, ten lanes each: 6 dual directions for theta_i, theta_j, 3 for the RELATIVE translation, the value.
// The positions enter the residual only through p = R_wi^T (p_j - p_i) + c(theta), R_wi = R_i R_imu_to_wheel, so
// d res / d p_j = (d res / d p) R_wi^T and d res / d p_i = -(d res / d p_j): three directions instead of six.
constexpr int WHEEL_PER_WAVE = 6;
__device__ void wheel_hex(const LinArgs& A, const DevParams& P, int b, int item, int sel, double* lds) {
    const int lane = threadIdx.x & 63, sub = lane / 10, dir = lane % 10;
    const int n = A.n, k = WHEEL_PER_WAVE * item + sub;
    const bool on = sub < WHEEL_PER_WAVE && k < n - 1;
    double* Y = lds + (sub < WHEEL_PER_WAVE ? sub : 0) * 64;   // [3][13] then Dp [3][3] at 40, R_wi [3][3] at 49
    const size_t fk = (size_t)b * (n - 1) + (on ? k : 0);
    if (on) {
        const double* si_ = A.x + ((size_t)b * n + k) * 15;
        const double* sj_ = si_ + 15;
        LJ pi[3], qi[3], pj[3], qj[3], dp[3], res[3];
        M3<LJ> Rwi;
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            pi[e] = LJ(si_[e]);
            qi[e] = LJ(si_[3 + e], dir == e ? 1.0 : 0.0);
            pj[e] = LJ(sj_[e]);
            qj[e] = LJ(sj_[3 + e], dir == 3 + e ? 1.0 : 0.0);
            dp[e] = LJ(0.0, dir == 6 + e ? 1.0 : 0.0);
        }
        wheel_res<LJ>(P, A.wheel_T + fk * 12, A.wheel_sqrtP + fk * 9, pi, qi, pj, qj, res, dp, &Rwi);
        if (dir < 3) { for (int r = 0; r < 3; ++r) Y[r * 13 + 3 + dir] = res[r].d; }
        else if (dir < 6) { for (int r = 0; r < 3; ++r) Y[r * 13 + 6 + dir] = res[r].d; }
        else if (dir < 9) { for (int r = 0; r < 3; ++r) Y[40 + r * 3 + (dir - 6)] = res[r].d; }
        else {
            for (int r = 0; r < 3; ++r) Y[r * 13 + 12] = res[r].v;
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Y[49 + r * 3 + c] = Rwi(r, c).v;
        }
    }
    __syncthreads();
    if (on && dir < 9) {   // position columns: Y[r][p_j c] = sum_k Dp[r][k] R_wi[c][k],  Y[r][p_i c] = -Y[r][p_j c]
        const int r = dir / 3, c = dir % 3;
        const double v = Y[40 + r * 3] * Y[49 + c * 3] + Y[40 + r * 3 + 1] * Y[49 + c * 3 + 1] + Y[40 + r * 3 + 2] * Y[49 + c * 3 + 2];
        Y[r * 13 + 6 + c] = v;
        Y[r * 13 + c] = -v;
    }
    __syncthreads();
    if (on && dir == 9 && A.dbg_wheel_res)
        for (int r = 0; r < 3; ++r) A.dbg_wheel_res[fk * 3 + r] = Y[r * 13 + 12];
    if (on && A.dbg_wheel_jac)
        for (int e = dir; e < 36; e += 10) A.dbg_wheel_jac[fk * 36 + e] = Y[(e / 12) * 13 + e % 12];
    if (on) {
        double* out = A.PW[sel] + fk * PWS;
        for (int e = dir; e < 91; e += 10) {
            int c1 = 0, rem = e;
            while (rem >= 13 - c1) { rem -= 13 - c1; ++c1; }
            const int c2 = c1 + rem;
            const double s = Y[c1] * Y[c2] + Y[13 + c1] * Y[13 + c2] + Y[26 + c1] * Y[26 + c2];
            out[c1 * 13 + c2] = s;
            out[c2 * 13 + c1] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------- ground
// ground_factor_p / ground_factor_q, src/factor/ground_factor.h:27-48, :59-82
template <class T>
__device__ __forceinline__ void ground_res(const DevParams& P, const T* p_, const T* q_, T* res) {
    Iso<T> tf_w_o = mul(make_tf(V3<T>(p_[0], p_[1], p_[2]), V3<T>(q_[0], q_[1], q_[2])), cast_iso<T>(P.Riw, P.tiw));
    res[0] = T(P.ground_p_info) * tf_w_o.t.z;
    V3<T> ABC(T(0.0), T(0.0), T(1.0));
    V3<T> z_axis(tf_w_o.R(0, 2), tf_w_o.R(1, 2), tf_w_o.R(2, 2));
    T sinn = norm(cross(z_axis, ABC));
    res[1] = T(P.ground_q_info) * dasin(sinn);
}

__device__ void ground_oct(const LinArgs& A, const DevParams& P, int b, int item, int sel, double* lds) {
    const int lane = threadIdx.x & 63, sub = lane >> 3, dir = lane & 7;
    const int n = A.n, i = 8 * item + sub;
    const bool on = i < n;
    double* Y = lds + sub * 16;   // [2][7]
    const size_t fi = (size_t)b * n + (on ? i : 0);
    double y[2] = {0.0, 0.0};
    if (on) {
        const double* s_ = A.x + fi * 15;
        LJ p[3], q[3], res[2];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            p[e] = LJ(s_[e], dir == e ? 1.0 : 0.0);
            q[e] = LJ(s_[3 + e], dir == 3 + e ? 1.0 : 0.0);
        }
        ground_res<LJ>(P, p, q, res);
#pragma unroll
        for (int r = 0; r < 2; ++r) y[r] = dir < 6 ? res[r].d : (dir == 6 ? res[r].v : 0.0);
    }
    if (dir < 7) { Y[dir] = y[0]; Y[7 + dir] = y[1]; }
    if (on && dir == 6 && A.dbg_ground_res) { A.dbg_ground_res[fi * 2] = y[0]; A.dbg_ground_res[fi * 2 + 1] = y[1]; }
    if (on && dir < 6 && A.dbg_ground_jac) { A.dbg_ground_jac[(fi * 2) * 6 + dir] = y[0]; A.dbg_ground_jac[(fi * 2 + 1) * 6 + dir] = y[1]; }
    __syncthreads();
    if (on) {
        double* out = A.PG[sel] + fi * PGS;
        const double mult = (double)n;   // the block set is added once per outer frame index (solver.cpp:142-159)
        for (int e = dir; e < 28; e += 8) {
            int c1 = 0, rem = e;
            while (rem >= 7 - c1) { rem -= 7 - c1; ++c1; }
            const int c2 = c1 + rem;
            const double s = mult * (Y[c1] * Y[c2] + Y[7 + c1] * Y[7 + c2]);
            out[c1 * 7 + c2] = s;
            out[c2 * 7 + c1] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------- dispatch
__device__ void imu_role(const LinArgs& A, const DevParams& P, int vblock, double* lds) {
    const int n = A.n, items = (n - 1 + IMU_PER_WAVE - 1) / IMU_PER_WAVE;
    const int b = vblock / items, item = vblock % items;
    if (b >= A.B) return;
    if (A.lm && A.lm[b].done) return;
    const int sel = A.lm ? (A.candidate ? 1 - A.lm[b].cur : A.lm[b].cur) : 0;
    imu_group(A, P, b, item, sel, lds);
}
__device__ void small_role(const LinArgs& A, const DevParams& P, int vblock, double* lds) {
    const int n = A.n, n_wheel = (n - 1 + WHEEL_PER_WAVE - 1) / WHEEL_PER_WAVE, items = n_wheel + (n + 7) / 8;
    const int b = vblock / items, item = vblock % items;
    if (b >= A.B) return;
    if (A.lm && A.lm[b].done) return;
    const int sel = A.lm ? (A.candidate ? 1 - A.lm[b].cur : A.lm[b].cur) : 0;
    if (item < n_wheel) wheel_hex(A, P, b, item, sel, lds);
    else ground_oct(A, P, b, item - n_wheel, sel, lds);
}
__global__ __launch_bounds__(64, 2) void k_lin_imu(LinArgs A, DevParams P) {
    __shared__ double lds[IMU_PER_WAVE * IMU_X];
    imu_role(A, P, (int)blockIdx.x, lds);
}
__global__ __launch_bounds__(64) void k_lin_small(LinArgs A, DevParams P) {
    __shared__ double lds[WHEEL_PER_WAVE * 64];
    small_role(A, P, (int)blockIdx.x, lds);
}
// Small batches (a single tracking window): every role in ONE launch, the role of a wave follows from its block index —
// one kernel and no fork / join events per linearisation, which is what a latency-bound 2-frame window pays for.
template <bool BOTH>
__global__ __launch_bounds__(64, 2) void k_lin_all(LinArgs A, DevParams P, int G, int n_laser, int n_imu) {
    __shared__ double lds[IMU_PER_WAVE * IMU_X];   // IMU / small roles; the laser role brings its own static LDS
    const int v = (int)blockIdx.x;
    if (v < n_laser) laser_wave_local<BOTH>(A, P, G, v);
    else if (v < n_laser + n_imu) imu_role(A, P, v - n_laser, lds);
    else small_role(A, P, v - n_laser - n_imu, lds);
}

// laser block range of every (window, frame): first block of window b owned by a frame >= i
__global__ void k_group_offsets(int B, int n, const int* laser_off, const int* laser_frame, int* group_off) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * (n + 1)) return;
    const int b = t / (n + 1), i = t % (n + 1);
    int lo = laser_off[b], hi = laser_off[b + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (laser_frame[mid] < i) lo = mid + 1; else hi = mid;
    }
    group_off[t] = lo;
}

// One linearisation.  Large batches: the laser, IMU and wheel+ground role kernels write disjoint partial-sum slots, so
// they run concurrently (main stream + two side streams joined by events); the matrix-core phase of the IMU kernel then
// overlaps the fp64 VALU work of the laser kernel on the same CUs.  Small batches: one launch for everything (k_lin_all).
// defer_join: the laser role stays on `s`, the IMU / small roles on the side streams, and the join is left to launch_linearize_join —
// a factor-sharded driver puts its exchange of the laser partial sums on `s` in between, so that it overlaps the small roles.
void launch_linearize(const LinArgs& A, const DevParams& P, hipStream_t s, const LinFork* fk, bool defer_join) {
    const int n = A.n, B = A.B;
    // groups per wave: one for small batches (latency), up to LASER_GMAX for large ones (no ragged last pass per group)
    int G = 1;
    if (A.mode != LIW_MODE_TRACK) while (G < LASER_GMAX && (long)B * ((n + 2 * G - 1) / (2 * G)) >= 4096) G *= 2;
    const int laser_waves = B * ((n + G - 1) / G);
    const int imu_waves = (A.eval_small && n > 1) ? B * ((n - 1 + IMU_PER_WAVE - 1) / IMU_PER_WAVE) : 0;
    const int small_waves = A.eval_small ? B * ((n - 1 + WHEEL_PER_WAVE - 1) / WHEEL_PER_WAVE + (n + 7) / 8) : 0;
    if (A.eval_small && laser_waves + imu_waves + small_waves <= 256) {
        const unsigned tot = (unsigned)(laser_waves + imu_waves + small_waves);
        if (A.mode == LIW_MODE_INIT) hipLaunchKernelGGL(k_lin_all<true>, dim3(tot), dim3(64), 0, s, A, P, G, laser_waves, imu_waves);
        else hipLaunchKernelGGL(k_lin_all<false>, dim3(tot), dim3(64), 0, s, A, P, G, laser_waves, imu_waves);
        return;
    }
    const bool fork = fk && fk->side[0] && A.eval_small;
    if (fork) {
        hipEventRecord(fk->ev_fork, s);
        hipStreamWaitEvent(fk->side[0], fk->ev_fork, 0);
        hipStreamWaitEvent(fk->side[1], fk->ev_fork, 0);
    }
    hipStream_t s_imu = fork ? fk->side[0] : s, s_small = fork ? fk->side[1] : s;
    hipLaunchKernelGGL(k_frame_tf, dim3((B * n * 2 * 4 + 255) / 256), dim3(256), 0, s, B, n, A.x, A.match_pose, A.ftf, P, A.lm);
    if (A.mode == LIW_MODE_INIT) hipLaunchKernelGGL(k_lin_laser<true>, dim3((unsigned)laser_waves), dim3(64), 0, s, A, P, G);
    else hipLaunchKernelGGL(k_lin_laser<false>, dim3((unsigned)laser_waves), dim3(64), 0, s, A, P, G);
    if (imu_waves) hipLaunchKernelGGL(k_lin_imu, dim3((unsigned)imu_waves), dim3(64), 0, s_imu, A, P);
    if (small_waves) hipLaunchKernelGGL(k_lin_small, dim3((unsigned)small_waves), dim3(64), 0, s_small, A, P);
    if (fork) {
        hipEventRecord(fk->ev_join[0], fk->side[0]);
        hipEventRecord(fk->ev_join[1], fk->side[1]);
        if (!defer_join) {
            hipStreamWaitEvent(s, fk->ev_join[0], 0);
            hipStreamWaitEvent(s, fk->ev_join[1], 0);
        }
    }
}
void launch_linearize_join(hipStream_t s, const LinFork* fk) {
    if (!fk || !fk->side[0]) return;
    hipStreamWaitEvent(s, fk->ev_join[0], 0);   // events never recorded count as complete
    hipStreamWaitEvent(s, fk->ev_join[1], 0);
}

// ------------------------------------------------------------------------------------------- factor-sharded exchange
// A laser group record (LP = 128 slots) is a signed expansion of NP pair totals (45 with both poses free, 21 with one): the
// exchange between ranks moves the NP totals only.  pack: record -> totals (representative slot and sign per total, table built on
// the host with the same slot map the laser kernel uses); unpack: sum over `world` gathered copies in rank order -> record.
__global__ void k_exchange_pack(int groups, int n, int np, LaserPackTable tb, const double* PL, const LmState* lm, double* buf) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= groups * np) return;
    const int grp = t / np, p = t % np;
    const bool dead = lm && lm[grp / n].done;   // finished windows are not re-linearised: their stale sums must not accumulate
    const double v = PL[(size_t)grp * LP + tb.slot[p]];
    buf[t] = dead ? 0.0 : (tb.neg[p] ? -v : v);
}
template <bool BOTH>
__global__ void k_exchange_unpack(int groups, int np, int world, size_t stride, const double* buf, double* PL) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= groups * LP) return;
    const int grp = t / LP, s = t % LP;
    const int code = laser_slot_code<BOTH>(s);
    double v = 0.0;
    if (code >= 0) {
        const double* src = buf + (size_t)grp * np + (code & 63);
        for (int r = 0; r < world; ++r) v += src[(size_t)r * stride];   // fixed rank order: every rank forms the same bits
        if (code & 64) v = -v;
    }
    PL[t] = v;
}
__global__ void k_count_active(int B, const LmState* lm, double* out) {
    __shared__ int cnt;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int c = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) c += lm[b].done ? 0 : 1;
    atomicAdd(&cnt, c);   // integer: order-independent
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (double)cnt;
}
void launch_exchange_pack(int B, int n, bool both, const double* PL, const LmState* lm, double* buf, hipStream_t s) {
    const int np = both ? 45 : 21, groups = B * n;
    LaserPackTable tb{};
    for (int p = 0; p < np; ++p) tb.slot[p] = -1;
    for (int sl = 0; sl < LP; ++sl) {
        const int code = both ? laser_slot_code<true>(sl) : laser_slot_code<false>(sl);
        if (code >= 0 && tb.slot[code & 63] < 0) { tb.slot[code & 63] = sl; tb.neg[code & 63] = (code & 64) ? 1 : 0; }
    }
    hipLaunchKernelGGL(k_exchange_pack, dim3((groups * np + 255) / 256), dim3(256), 0, s, groups, n, np, tb, PL, lm, buf);
    if (lm) hipLaunchKernelGGL(k_count_active, dim3(1), dim3(256), 0, s, B, lm, buf + (size_t)groups * np);
}
void launch_exchange_unpack(int B, int n, bool both, int world, size_t stride, const double* buf, double* PL, hipStream_t s) {
    const int np = both ? 45 : 21, groups = B * n;
    if (both) hipLaunchKernelGGL(k_exchange_unpack<true>, dim3((groups * LP + 255) / 256), dim3(256), 0, s, groups, np, world, stride, buf, PL);
    else hipLaunchKernelGGL(k_exchange_unpack<false>, dim3((groups * LP + 255) / 256), dim3(256), 0, s, groups, np, world, stride, buf, PL);
}
#ifdef LIW_CLK
extern "C" void liw_debug_clk_lin(long long* out, int nn) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk_lin), sizeof(long long) * nn); }
#endif
void launch_group_offsets(int B, int n, const int* laser_off, const int* laser_frame, int* group_off, hipStream_t s) {
    const int tot = B * (n + 1);
    hipLaunchKernelGGL(k_group_offsets, dim3((tot + 255) / 256), dim3(256), 0, s, B, n, laser_off, laser_frame, group_off);
}

}  // namespace liw
