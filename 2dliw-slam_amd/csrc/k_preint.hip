// k_preint.hip — batched pre-integration on the device ("batch replay" form of SURVEY §8 rows a5 / a6).
//
// The reference integrates one interval at a time on the CPU while messages arrive
// (imu_preintegraption::update, src/factor/imu_preintegraption.h:170-208; wheel_odom_preintegration,
// src/factor/wheel_odom_preintegration.h:62-152).  When whole logs are replayed, every interval between two frames is
// independent, so M intervals are integrated in parallel here:
//   k_preint_imu       4 intervals per wave, 16 lanes each; lane c owns COLUMN c of J and of the covariance P.
//                      F = I + dt*Fc only mixes rows, so J <- F J and T = F P are per-lane; P' = F P F^T uses the
//                      symmetry P' = (F T^T)^T: one 15x15 transpose through LDS, then the same per-lane row update.
//   k_preint_imu_sqrt  one wave per interval: sqrt_inverse_P = LLT(P^-1).matrixL().transpose() (:149) with the
//                      register-resident fused Cholesky: P = L L^T with identity right-hand sides gives W = L^-1,
//                      P^-1 = W^T W on the fp64 matrix cores, a second Cholesky gives the result.
//   k_preint_wheel     one thread per interval (a handful of odometry samples each).
// Same quirks as the reference (SURVEY Appendix C 5,6): previous sample drives the whole Euler step, F(gamma,gamma)
// is built from hat_gyro - last_ba, P0 = 1e-5 I.  Replay semantics = the host pre-integrators of liw_preint.cpp:
// sample 0 of an interval seeds last_info, the accumulator is reset at t_start, integrated to t_end.
#include "liw_kernels.hpp"

namespace liw {

typedef double d4p __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double rdl(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// rows alpha(0:3) beta(3:6) gamma(6:9) <- F * (old column); ba, bw rows are unchanged.  col = one column (15 rows).
__device__ __forceinline__ void apply_F(double (&col)[15], double dt, const double (&Fbg)[3][3], const double (&Fbb)[3][3], const double (&Fgg)[3][3]) {
    double na[3], nb[3], ng[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        na[i] = col[i] + dt * col[3 + i];
        double sb = col[3 + i], sg = -dt * col[12 + i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            sb += Fbg[i][k] * col[6 + k] + Fbb[i][k] * col[9 + k];
            sg += Fgg[i][k] * col[6 + k];
        }
        nb[i] = sb; ng[i] = sg;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { col[i] = na[i]; col[3 + i] = nb[i]; col[6 + i] = ng[i]; }
}

__global__ __launch_bounds__(64) void k_preint_imu(int M, const int* sample_off, const double* samples, const double* t_start,
                                                   const double* t_end, const double* bias6, PreintNoise N, double* Xout, double* Jout,
                                                   double* Pout, double* Dtout) {
    __shared__ double Tt[4][16 * 16];
    const int lane = threadIdx.x & 63, grp = lane >> 4, c = lane & 15;
    const int m = blockIdx.x * 4 + grp;
    const bool on = m < M;
    const int mm = on ? m : 0;
    const int s0 = sample_off[mm], s1 = sample_off[mm + 1];
    double X[15], Jc[15], Pc[15];
#pragma unroll
    for (int r = 0; r < 15; ++r) { X[r] = 0.0; Jc[r] = r == c ? 1.0 : 0.0; Pc[r] = r == c ? 0.00001 : 0.0; }
#pragma unroll
    for (int k = 0; k < 6; ++k) X[9 + k] = bias6[mm * 6 + k];
    double last_t = t_start[mm], Dt = 0.0;
    double la[3] = {samples[(size_t)s0 * 7 + 1], samples[(size_t)s0 * 7 + 2], samples[(size_t)s0 * 7 + 3]};
    double lg[3] = {samples[(size_t)s0 * 7 + 4], samples[(size_t)s0 * 7 + 5], samples[(size_t)s0 * 7 + 6]};
    // the wave walks as many steps as its longest interval; shorter intervals idle (exec-masked through `act`)
    int nsteps = on ? (s1 - s0) : 0;   // samples 1..cnt-1 plus the final update_only_t
    int nmax = nsteps;
#pragma unroll
    for (int o = 32; o >= 16; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
    for (int step = 1; step <= nmax; ++step) {
        const bool act = step <= nsteps;
        const bool is_last = step == nsteps;
        const size_t si = (size_t)(s0 + (act && !is_last ? step : 0)) * 7;
        const double tn = (act && !is_last) ? samples[si] : t_end[mm];
        const double dt = act ? tn - last_t : 0.0;
        // ---- update(dt) with the PREVIOUS sample (imu_preintegraption.h:170-208)
        const V3<double> a_unb(la[0] - X[9], la[1] - X[10], la[2] - X[11]);
        const V3<double> w_unb(lg[0] - X[12], lg[1] - X[13], lg[2] - X[14]);
        const M3<double> Rz = exp_so3(V3<double>(X[6], X[7], X[8]));
        const V3<double> Ra = mul(Rz, a_unb);
        const double ra[3] = {Ra.x, Ra.y, Ra.z};
        const double b0[3] = {X[3], X[4], X[5]};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            X[k] = X[k] + b0[k] * dt + 0.5 * ra[k] * dt * dt;
            X[3 + k] = b0[k] + ra[k] * dt;
        }
        const V3<double> g2 = log_SO3(mul(Rz, exp_so3(V3<double>(w_unb.x * dt, w_unb.y * dt, w_unb.z * dt))));
        // a lane group whose interval is finished runs the step with dt = 0: F = I and the added noise is 0, so X, J, P
        // stay bit-identical except gamma (log(exp(gamma)) round trip), which is therefore only written while active
        X[6] = act ? g2.x : X[6]; X[7] = act ? g2.y : X[7]; X[8] = act ? g2.z : X[8];
        // F blocks
        double Fbg[3][3], Fbb[3][3], Fgg[3][3];
        {
            const double ax[3] = {a_unb.x, a_unb.y, a_unb.z};
            const double wx[3] = {lg[0] - X[9], lg[1] - X[10], lg[2] - X[11]};   // sic: gyro minus ACC bias (:192)
            // skew(v)[i][j]
            auto sk = [](const double* v, int i, int j) {
                if (i == j) return 0.0;
                if (i == 0 && j == 1) return -v[2];
                if (i == 1 && j == 0) return v[2];
                if (i == 0 && j == 2) return v[1];
                if (i == 2 && j == 0) return -v[1];
                if (i == 1 && j == 2) return -v[0];
                return v[0];
            };
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double rax = Rz(i, 0) * sk(ax, 0, j) + Rz(i, 1) * sk(ax, 1, j) + Rz(i, 2) * sk(ax, 2, j);
                    Fbg[i][j] = -rax * dt;
                    Fbb[i][j] = -Rz(i, j) * dt;
                    Fgg[i][j] = (i == j ? 1.0 : 0.0) - sk(wx, i, j) * dt;
                }
        }
        // J <- F J ; T = F P (both per column, i.e. per lane)
        apply_F(Jc, dt, Fbg, Fbb, Fgg);
        apply_F(Pc, dt, Fbg, Fbb, Fgg);
        // P' = T F^T = (F T^T)^T, P' symmetric: transpose T through LDS, apply F again
        __syncthreads();
        if (c < 15) {
#pragma unroll
            for (int r = 0; r < 15; ++r) Tt[grp][c * 16 + r] = Pc[r];     // T[r][c] stored as Tt[c][r]
        }
        __syncthreads();
        if (c < 15) {
#pragma unroll
            for (int r = 0; r < 15; ++r) Pc[r] = Tt[grp][r * 16 + c];     // column c of T^T: T[c][r]
        }
        apply_F(Pc, dt, Fbg, Fbb, Fgg);                                    // column c of F T^T = row c of P' = column c of P'
        // + (G dt) Q (G dt)^T : (beta,beta) += dt^2 Rz Q_na Rz^T ; diag gamma / ba / bw += q dt^2   (:195-206)
        const double dt2 = dt * dt;
        if (c >= 3 && c < 6) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 3; ++k) s += Rz(i, k) * N.q_na[k] * (c == 3 ? Rz(0, k) : (c == 4 ? Rz(1, k) : Rz(2, k)));
                Pc[3 + i] += s * dt2;
            }
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (c == 6 + k) Pc[6 + k] += N.q_nw[k] * dt2;
            if (c == 9 + k) Pc[9 + k] += N.q_nba[k] * dt2;
            if (c == 12 + k) Pc[12 + k] += N.q_nbw[k] * dt2;
        }
        if (act) {
            Dt += dt;
            last_t = tn;
            if (!is_last) {
                la[0] = samples[si + 1]; la[1] = samples[si + 2]; la[2] = samples[si + 3];
                lg[0] = samples[si + 4]; lg[1] = samples[si + 5]; lg[2] = samples[si + 6];
            }
        }
    }
    if (on && c < 15) {
#pragma unroll
        for (int r = 0; r < 15; ++r) { Jout[(size_t)m * 225 + r * 15 + c] = Jc[r]; Pout[(size_t)m * 225 + r * 15 + c] = Pc[r]; }
        if (c == 0) {
#pragma unroll
            for (int r = 0; r < 15; ++r) Xout[(size_t)m * 15 + r] = X[r];
            Dtout[m] = Dt;
        }
    }
}

// right-looking Cholesky + forward substitution, column-per-lane (same routine as k_lm.hip's fused_chol_solve)
__device__ __forceinline__ void fused_chol15(double (&a)[15]) {
#pragma unroll
    for (int k = 0; k < 15; ++k) {
        const double piv = rdl(a[k], k);
        const double inv = 1.0 / sqrt(piv);
        const double wk = a[k] * inv;
        a[k] = wk;
#pragma unroll
        for (int r = k + 1; r < 15; ++r) a[r] -= rdl(wk, r) * wk;
    }
}

__global__ __launch_bounds__(64) void k_preint_imu_sqrt(int M, const double* P, double* sqrtP) {
    __shared__ double W[256];
    const int m = blockIdx.x, lane = threadIdx.x & 63;
    if (m >= M) return;
    for (int e = lane; e < 256; e += 64) W[e] = 0.0;
    __syncthreads();
    double col[15];
#pragma unroll
    for (int r = 0; r < 15; ++r) {
        double v = 0.0;
        if (lane < 15) v = P[(size_t)m * 225 + r * 15 + lane];
        else if (lane >= 16 && lane < 31) v = (r == lane - 16) ? 1.0 : 0.0;     // identity right-hand sides
        col[r] = v;
    }
    fused_chol15(col);                       // lanes 16..30: columns of L^-1
    if (lane >= 16 && lane < 31) {
#pragma unroll
        for (int r = 0; r < 15; ++r) W[r * 16 + (lane - 16)] = col[r];
    }
    __syncthreads();
    // P^-1 = L^-T L^-1 = W^T W  (fp64 MFMA 16x16x4, operand = W[k][i])
    d4p acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
        const int k = (lane >> 4) + 4 * cc;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(W[k * 16 + (lane & 15)], W[k * 16 + (lane & 15)], acc, 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) W[((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[r];   // W <- P^-1
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 15; ++r) col[r] = lane < 15 ? W[r * 16 + lane] : 0.0;
    fused_chol15(col);                       // lane j: row j of M, P^-1 = M M^T ; sqrt_inverse_P = M^T
    if (lane < 15) {
#pragma unroll
        for (int k = 0; k < 15; ++k) sqrtP[(size_t)m * 225 + k * 15 + lane] = k <= lane ? col[k] : 0.0;   // U[k][j] = M[j][k]
    }
}

// wheel_odom_preintegration replay, one thread per interval (samples: t, R(9 row-major), t(3))
__global__ void k_preint_wheel(int M, const int* sample_off, const double* samples, const double* t_start, const double* t_end,
                               PreintNoise N, double* T12, double* sq9, double* Dtout) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const int s0 = sample_off[m], s1 = sample_off[m + 1];
    Iso<double> delta, last_pose;
#pragma unroll
    for (int k = 0; k < 9; ++k) { delta.R.m[k] = (k % 4 == 0) ? 1.0 : 0.0; last_pose.R.m[k] = delta.R.m[k]; }
    delta.t = V3<double>(0.0, 0.0, 0.0); last_pose.t = delta.t;
    double last_update = -1.0, last_add = 0.0, Dt = 0.0, v[3] = {0, 0, 0}, om[3] = {0, 0, 0};
    bool did_reset = false;
    auto update_by_v = [&](double dt) {
        if (dt <= 0 || dt >= 10) return;
        Dt += dt;
        Iso<double> dT = make_tf(V3<double>(v[0] * dt, v[1] * dt, v[2] * dt), V3<double>(om[0] * dt, om[1] * dt, om[2] * dt));
        delta = mul(delta, dT);
    };
    auto set_identity = [&]() {
#pragma unroll
        for (int k = 0; k < 9; ++k) delta.R.m[k] = (k % 4 == 0) ? 1.0 : 0.0;
        delta.t = V3<double>(0.0, 0.0, 0.0);
    };
    auto reset = [&](double t) { last_update = t; set_identity(); Dt = 0.0; };
    const double ts = t_start[m], te = t_end[m];
    for (int s = s0; s < s1; ++s) {
        const double* q = samples + (size_t)s * 13;
        if (!did_reset && q[0] > ts) {
            if (last_update >= 0) { update_by_v(ts - last_update); }
            reset(ts);
            did_reset = true;
        }
        Iso<double> pose = cast_iso<double>(q + 1, q + 10);
        if (last_update < 0) {
            last_pose = pose; last_add = q[0]; last_update = q[0];
            set_identity();
#pragma unroll
            for (int k = 0; k < 3; ++k) v[k] = om[k] = 0.0;
            continue;
        }
        const double dt = q[0] - last_add;
        Iso<double> rel = mul(inverse(last_pose), pose);
        const V3<double> dth = log_SO3(rel.R);
        if (dt < 0.05) continue;
        v[0] = rel.t.x / dt; v[1] = rel.t.y / dt; v[2] = rel.t.z / dt;
        om[0] = dth.x / dt; om[1] = dth.y / dt; om[2] = dth.z / dt;
        update_by_v(q[0] - last_update);
        last_pose = pose; last_add = q[0]; last_update = q[0];
    }
    if (!did_reset) { if (last_update >= 0) update_by_v(ts - last_update); reset(ts); }
    if (last_update >= 0) { update_by_v(te - last_update); last_update = te; }
    const V3<double> dq = log_SO3(delta.R);
    const double len_norm = fmax(delta.t.x * delta.t.x + delta.t.y * delta.t.y + delta.t.z * delta.t.z, 0.005 * 0.005);
    const double yaw_norm = fmax(dq.x * dq.x + dq.y * dq.y + dq.z * dq.z, 0.005 * 0.005);
    const double kd[3] = {len_norm, len_norm, yaw_norm};
#pragma unroll
    for (int k = 0; k < 9; ++k) { sq9[(size_t)m * 9 + k] = 0.0; T12[(size_t)m * 12 + k] = delta.R.m[k]; }
#pragma unroll
    for (int k = 0; k < 3; ++k) sq9[(size_t)m * 9 + k * 4] = sqrt(1.0 / (N.wheel_cov[k] * kd[k]));
    T12[(size_t)m * 12 + 9] = delta.t.x; T12[(size_t)m * 12 + 10] = delta.t.y; T12[(size_t)m * 12 + 11] = delta.t.z;
    Dtout[m] = Dt;
}

void launch_preint_imu(int M, const int* sample_off, const double* samples, const double* t_start, const double* t_end, const double* bias6,
                       const PreintNoise& N, double* X, double* J, double* Pscratch, double* sqrtP, double* Dt, hipStream_t s) {
    hipLaunchKernelGGL(k_preint_imu, dim3((M + 3) / 4), dim3(64), 0, s, M, sample_off, samples, t_start, t_end, bias6, N, X, J, Pscratch, Dt);
    hipLaunchKernelGGL(k_preint_imu_sqrt, dim3(M), dim3(64), 0, s, M, Pscratch, sqrtP);
}
void launch_preint_wheel(int M, const int* sample_off, const double* samples, const double* t_start, const double* t_end,
                         const PreintNoise& N, double* T12, double* sq9, double* Dt, hipStream_t s) {
    hipLaunchKernelGGL(k_preint_wheel, dim3((M + 127) / 128), dim3(128), 0, s, M, sample_off, samples, t_start, t_end, N, T12, sq9, Dt);
}

}  // namespace liw
