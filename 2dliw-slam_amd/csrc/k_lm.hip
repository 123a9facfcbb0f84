// k_lm.hip — normal-equation assembly, Levenberg–Marquardt step, Schur marginalisation (gfx950, fp64).
//
// One wavefront per window.  The normal equations of a window are block tri-diagonal (IMU / wheel blocks couple
// frames i-1,i) plus, in the init topology, an "arrow" from frame 0's pose to every laser frame (reference
// src/factor/solver.cpp:93-106).  Frames are eliminated n-1 .. 1 and frame 0 last, so no fill is created
// beyond a 6x15 arrow row.  Per eliminated frame: 15x15 Cholesky in LDS, 22 forward substitutions on 22 lanes,
// and the Schur-complement products  [Wo|z]^T[Wo|z],  [Wr|0]^T[Wo|z],  [Wr|0]^T[Wr|0]  as 16x16x16 fp64 MFMA
// tiles (v_mfma_f64_16x16x4_f64; 15 padded to 16 is the natural tile of this problem).
//
// What is restated from Ceres (third-party, not vendored; call sites solver.cpp:161-168, :795-802): the
// TRUST_REGION/LEVENBERG_MARQUARDT loop with default options — Jacobi scaling 1/(1+|col|) fixed at iteration 0,
// LM diagonal clamp(diag, 1e-6, 1e32)/radius, rho = cost change / model change, accept if rho > 1e-3 with
// radius / max(1/3, 1-(2rho-1)^3), reject -> radius/nu, nu*=2; parameter / function tolerance are tested on the
// candidate BEFORE acceptance and end the solve without applying that candidate; so3 Plus = normalize_so3(x+d)
// (src/factor/factor_common.h:41-53).  Marginalisation restates solver.cpp:4-40 and :390-402 on the same
// block structure (chain elimination 0..n-2, eigen floor 1e-8).
#include "liw_kernels.hpp"
#include "k_lm_common.hpp"

namespace liw {

typedef double d4 __attribute__((ext_vector_type(4)));
// optional phase timing (tools/clk_probe.py): build with -DLIW_CLK to record s_memtime stamps of window 0
#ifdef LIW_CLK
__device__ long long g_clk[8192];
__device__ long long g_span[3 * 16384];   // per window: start, end (s_memtime), hardware id of the wave
#ifndef LIW_CLK_IT
#define LIW_CLK_IT 3
#endif
#define STAMP(id) do { if (b == 0 && lane == 0 && iteration_dbg == LIW_CLK_IT) g_clk[(id)] = clock64(); } while (0)
#define STAMPE(id) do { if (b == 0 && lane == 0 && iteration == LIW_CLK_IT) g_clk[(id)] = clock64(); } while (0)   // before iteration_dbg exists
#define STAMPM(id) do { if (b == 0 && lane == 0) g_clk[(id)] = clock64(); } while (0)                              // marginalisation kernel
// inside a round of the four-wave Jacobi (thread 0, second sweep, fourth round); JSTAMPV first waits for the value v
#define JSTAMP(id) do { if (t == 0 && blockIdx.x == 0 && sweep == 1 && rnd == 3) g_clk[(id)] = clock64(); } while (0)
#define JSTAMPV(id, v) do { double w_ = (v); asm volatile("" : "+v"(w_)); if (t == 0 && blockIdx.x == 0 && sweep == 1 && rnd == 3) g_clk[(id)] = clock64(); } while (0)
#define SPAN(k) do { if (lane == 0 && iteration_dbg == LIW_CLK_IT && b < 16384) { g_span[3 * b + (k)] = clock64(); \
                     if ((k) == 0) g_span[3 * b + 2] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); } } while (0)
#else
#define STAMP(id) do { } while (0)
#define STAMPE(id) do { } while (0)
#define STAMPM(id) do { } while (0)
#define JSTAMP(id) do { } while (0)
#define JSTAMPV(id, v) do { } while (0)
#define SPAN(k) do { } while (0)
#endif


struct AsmCtx {
    int n, mode, fast, b, buf;
    int pif;                  // 1: PI holds per-frame IMU records (PIF_*, liw_kernels.hpp), 0: per-block records (PI_*)
    const double* PL; const double* PI; const double* PW; const double* PG;   // of window b already offset? no: batch base
    const double* x;          // states the partials were evaluated at, window base [n][15]
    const double* pJ; const double* pX; bool prior_on;
};

// prior residual r = linearized_J (X - linearized_X)  (marginalization_factor.h:22-53, linearized_R omitted there)
__device__ __forceinline__ double prior_r(const AsmCtx& c, int k) {
    const double* xs = c.x + (size_t)(c.n - 2) * 15;
    double s = 0.0;
    for (int j = 0; j < 15; ++j) s += c.pJ[k * 15 + j] * (xs[j] - c.pX[j]);
    return s;
}

// Assemble frame i's blocks (ambient -> tangent, constants masked), UNSCALED, into 16x16 LDS tiles (ld 16):
//   Dm = H[i,i], Om = H[i-1,i] (rows: frame i-1), Rm rows 0..5 = H[0(pose), i] (init arrow, i >= 2), gv = g_i.
struct FrameExtra { double sc_i, sc_m, dg_i, x_i; };   // per lane v < 15: scale of frame i / i-1, LM diagonal, state entry

// Where assemble_frame puts a frame's blocks.  LAYOUT 0: three 16x16 tiles D = H[i,i], O = H[i-1,i] (rows: frame i-1),
// R rows 0..5 = H[0(pose), i] and g[16].  LAYOUT 1 ("lane layout", k_lm_step): one [15][64] matrix whose column is the
// lane that will own it in the elimination — cols 0..14 = D, 16..30 = O^T, 32..37 = R^T, 40 = g.
constexpr int MS = 41;   // row stride of the lane-layout tiles of k_lm_step: lanes 0..40 carry columns (D 0-14, O^T 16-30, R^T 32-37, g 40)
template <int LAYOUT> struct Tiles {
    double* D; double* O; double* R; double* g;
    __device__ __forceinline__ double& d(int r, int c) const { return LAYOUT ? D[r * MS + c] : D[r * 16 + c]; }
    __device__ __forceinline__ double& o(int r, int c) const { return LAYOUT ? D[c * MS + 16 + r] : O[r * 16 + c]; }
    __device__ __forceinline__ double& rr(int r, int c) const { return LAYOUT ? D[c * MS + 32 + r] : R[r * 16 + c]; }
    __device__ __forceinline__ double& gg(int r) const { return LAYOUT ? D[r * MS + 40] : g[r]; }
};

// Raw values of one frame's assembly, as loaded (asm_issue) and before they are combined (asm_commit): splitting the
// two lets k_lm_step issue frame i-1's loads before it factorises frame i, hiding the memory round trip.
struct AsmRegs {
    double v5[4], v6[4], v7[4];             // IMU partial entries: block (i-1,i) jj / block (i,i+1) ii / block (i-1,i) ij
    double t1, t2, t3, t4, t8, t9, t10;     // 6x6 pose-block terms (laser, wheel, ground)
    double g1, g2, g3, g4, g5, g6;          // gradient terms
    double xq, sc_i, sc_m, dg_i;            // state entries, Jacobi scales, LM diagonal
    double e1[2], e2;                       // upward sweep, frame 1 only: H[frame 0 pose, frame 1] of the IMU / wheel block (0,1)
};
// frame n-2 with a prior: linearized_jacobians (this lane's elements of the 16x16 tile), x - linearized_X.  Kept out of AsmRegs: the
// throughput instantiation of k_lm_step has no registers to spare and loads them inside asm_commit instead.
struct PriorRegs { double pj[4], pdx; };
__device__ __forceinline__ PriorRegs prior_issue(const AsmCtx& c, int lane) {
    PriorRegs P;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q, r = e >> 4, cc = e & 15;
        P.pj[q] = (r < 15 && cc < 15) ? c.pJ[r * 15 + cc] : 0.0;
    }
    const int v = lane < 15 ? lane : 0;
    P.pdx = c.x[(size_t)(c.n - 2) * 15 + v] - c.pX[v];
    return P;
}
constexpr int ASM_TMP = 16;                 // asm_commit scratch: r_prior

// All loads of a frame are issued up front, branch-free (clamped addresses; masking happens in asm_commit), so the
// wave pays ONE memory round trip per frame instead of one per conditional term.
// dir: which chain neighbour the O tile couples frame i to.  -1 (default): frame i-1 (frames are eliminated n-1 .. 1, then 0);
// +1: frame i+1 (upward sweep of the two-wave kernel k_lm_step_tw, where frame 0 is split into its pose — the hub every laser
// frame's arrow points to — and the rest, see there).
__device__ __forceinline__ AsmRegs asm_issue(const AsmCtx& c, int i, const double* scl, const double* dgl, int lane = threadIdx.x & 63, int dir = -1) {
    const int n = c.n;
    const bool up = dir > 0;
    const double* PLb = c.PL + (size_t)c.b * n * LP;
    const bool pif = c.pif != 0;
    const bool hasm = i >= 1, hasp = i <= n - 2;
    // per-block records: PIm = block (i-1, i), PIp = block (i, i+1).  Per-frame records: PIm = frame i's record (diagonal tile complete,
    // coupling to frame i-1, both gradient parts), PIp = frame i+1's (coupling of block (i, i+1), for the upward sweep)
    const double* PIb = c.PI + (size_t)c.b * (pif ? (size_t)n * PIFS : (size_t)(n - 1) * PIS);
    const double* PWb = c.PW + (size_t)c.b * (n - 1) * PWS;
    const double* PGb = c.PG + (size_t)c.b * n * PGS;
    const double* PIm = pif ? PIb + (size_t)i * PIFS : PIb + (size_t)(hasm ? i - 1 : 0) * PIS;
    const double* PIp = pif ? PIb + (size_t)(hasp ? i + 1 : i) * PIFS : PIb + (size_t)(hasp ? i : 0) * PIS;
    const int oJJ = pif ? PIF_D : PI_JJ, oIJ = pif ? PIF_IJ : PI_IJ;
    const double* PWm = PWb + (size_t)(hasm ? i - 1 : 0) * PWS;
    const double* PWp = PWb + (size_t)(hasp ? i : 0) * PWS;
    // frame-level bases are wave-uniform (SGPR pairs), the lane-dependent part of every address is an unsigned 32-bit element offset:
    // global_load with scalar base + vector offset instead of a 64-bit vector address per load (148 of the kernel's 236 loads)
    const double* PLi = PLb + (size_t)i * LP;
    const double* PL1 = PLb + (size_t)(n > 1 ? 1 : 0) * LP;
    const double* PGi = PGb + (size_t)i * PGS;
    AsmRegs R;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q, r = e >> 4, cc = e & 15;
        const bool valid = r < 15 && cc < 15;
        const int rs = valid ? r : 0, cs = valid ? cc : 0;
        const int tq = pi_tri(rs, cs);
        R.v5[q] = PIm[(unsigned)(oJJ + tq)];
        R.v6[q] = pif ? 0.0 : PIp[(unsigned)(PI_II + tq)];
        R.v7[q] = up ? PIp[(unsigned)(oIJ + cs * 15 + rs)] : PIm[(unsigned)(oIJ + rs * 15 + cs)];   // (row r = neighbour's entry, column cc = frame i's)
    }
    {   // the 6x6 pose block terms: one element per lane (lanes 0..35)
        const bool pl = lane < 36;
        const int r = pl ? lane / 6 : 0, cc = pl ? lane % 6 : 0;
        R.t1 = PLi[(unsigned)(36 + r * 6 + cc)];
        R.t2 = PWm[(unsigned)PW_JJ(r, cc)];
        R.t3 = PWp[(unsigned)PW_II(r, cc)];
        R.t4 = PGi[(unsigned)PG_H(r, cc)];
        R.t8 = up ? PWp[(unsigned)PW_IJ(cc, r)] : PWm[(unsigned)PW_IJ(r, cc)];
        R.e2 = up ? PWm[(unsigned)PW_IJ(r, cc)] : 0.0;
        R.t9 = PL1[(unsigned)(72 + r * 6 + cc)];
        R.t10 = PLi[(unsigned)(72 + r * 6 + cc)];
    }
    {
        const int r = lane < 15 ? lane : 0, r6 = r < 6 ? r : 0;
        R.g1 = PLi[(unsigned)(114 + r6)];
        R.g2 = PWm[(unsigned)PW_G(6 + r6)];
        R.g3 = PWp[(unsigned)PW_G(r6)];
        R.g4 = PGi[(unsigned)PG_G(r6)];
        R.g5 = PIm[(unsigned)((pif ? PIF_GJ : PI_G + 15) + r)];
        R.g6 = pif ? PIm[(unsigned)(PIF_GI + r)] : PIp[(unsigned)(PI_G + r)];
    }
    const int nbf = up ? (hasp ? i + 1 : i) : (hasm ? i - 1 : i);   // the sweep neighbour (or i itself at the end of the chain)
    {   // states (rotation vectors of frames i, its neighbour, 0 for the so3 Plus Jacobian test) and the LM scales
        int idx = i * 15 + (lane < 15 ? lane : 0);
        if (lane >= 16 && lane < 19) idx = nbf * 15 + 3 + (lane - 16);
        if (lane >= 20 && lane < 23) idx = 3 + (lane - 20);
        R.xq = c.x[(unsigned)idx];
        const int v = lane < 15 ? lane : 0;
        R.sc_i = scl ? scl[(unsigned)(i * 15 + v)] : 1.0;
        R.sc_m = scl ? scl[(unsigned)(nbf * 15 + v)] : 1.0;
        R.dg_i = dgl ? dgl[(unsigned)(i * 15 + v)] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {   // rows 0..5 (pose of frame i-1) of the IMU block (i-1, i): element e = r * 15 + c < 90
        const int e = lane + 64 * q;
        R.e1[q] = up ? PIm[(unsigned)(oIJ + (e < 90 ? e : 0))] : 0.0;
    }
    return R;
}

__device__ __forceinline__ d4 xty16(const double* X, const double* Y);
// Combine the loaded values into frame i's blocks (ambient -> tangent, constants masked), UNSCALED.
template <int LAYOUT>
__device__ void asm_commit(const AsmCtx& c, int i, const AsmRegs& R, const Tiles<LAYOUT>& T_, double* tmp, FrameExtra* ex = nullptr,
                           int lane = threadIdx.x & 63, int dir = -1, const PriorRegs* pr = nullptr) {
    const int n = c.n;
    const double* PLb = c.PL + (size_t)c.b * n * LP;
    const bool hasm = i >= 1, hasp = i <= n - 2;
    const bool up = dir > 0, hasnb = up ? hasp : hasm;
    const int nbf = up ? (hasp ? i + 1 : i) : (hasm ? i - 1 : i);
    const bool arrow1 = up && i == 1;   // upward sweep: frame 1 is tied to the hub (frame 0's pose) by the IMU / wheel block (0,1) too
    const bool prior_here = c.prior_on && i == n - 2;
    d4 jtj = {0.0, 0.0, 0.0, 0.0};
    double gprior = 0.0;
    if (prior_here) {
        // linearized_jacobians as a zero-padded 16x16 tile, x - linearized_X behind it: staged in the storage of the output tiles (dead
        // until the stores below), so the prior costs the step kernels no LDS.  r_prior = J (x - X) into tmp[0..14]; J^T J on the matrix
        // cores (element layout = the one of the loop below); gradient term J^T r_prior.
        double* Jt = T_.D;
        double* dxp = LAYOUT ? T_.D + 256 : T_.O;
        const PriorRegs P_ = pr ? *pr : prior_issue(c, lane);   // pr: issued by the caller one frame ahead, with the partial sums
#pragma unroll
        for (int q = 0; q < 4; ++q) Jt[lane + 64 * q] = P_.pj[q];
        if (lane < 16) dxp[lane] = lane < 15 ? P_.pdx : 0.0;
        lds_sync();
        if (lane < 15) {
            double sp = 0.0;
#pragma unroll
            for (int j = 0; j < 15; ++j) sp += Jt[lane * 16 + j] * dxp[j];
            tmp[lane] = sp;
        }
        jtj = xty16(Jt, Jt);
        lds_sync();
        if (lane < 15) {
#pragma unroll
            for (int k = 0; k < 15; ++k) gprior += Jt[k * 16 + lane] * tmp[k];
        }
        lds_sync();
    }
    double dI[4], oI[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q, r = e >> 4, cc = e & 15;
        const bool valid = r < 15 && cc < 15;
        dI[q] = ((valid && (c.pif ? n > 1 : hasm)) ? R.v5[q] : 0.0) + ((valid && hasp && !c.pif) ? R.v6[q] : 0.0);   // (per-frame records: v5 is the whole tile; a 1-frame window has no IMU block and nobody wrote its record)
        oI[q] = (valid && hasnb) ? R.v7[q] : 0.0;
    }
    double dP = 0.0, oP = 0.0, rP = 0.0;
    {
        const bool pl = lane < 36;
        const int r = pl ? lane / 6 : 0, cc = pl ? lane % 6 : 0;
        dP = R.t1 + (hasm ? R.t2 : 0.0) + (hasp ? R.t3 : 0.0) + R.t4;
        // (only the init topology has a free pose `a`: in the tracking / marginalisation topologies the Haa | ga slots are structural zeros,
        //  laser_slot_code<false>, and the n dependent loads behind the main batch are skipped)
        if (i == 0 && c.mode == LIW_MODE_INIT) {   // every laser frame's Haa lands on frame 0's pose: 4 independent partial sums keep the n loads in flight together
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            int j = 0;
            for (; j + 4 <= n; j += 4) {
                s0 += PLb[(size_t)j * LP + r * 6 + cc]; s1 += PLb[(size_t)(j + 1) * LP + r * 6 + cc];
                s2 += PLb[(size_t)(j + 2) * LP + r * 6 + cc]; s3 += PLb[(size_t)(j + 3) * LP + r * 6 + cc];
            }
            for (; j < n; ++j) s0 += PLb[(size_t)j * LP + r * 6 + cc];
            dP += (s0 + s1) + (s2 + s3);
        }
        oP = hasnb ? R.t8 + ((!up && i == 1) ? R.t9 : 0.0) : 0.0;
        rP = (i >= 2 || arrow1) ? R.t10 + (arrow1 ? R.e2 : 0.0) : 0.0;
        if (!pl) { dP = 0.0; oP = 0.0; rP = 0.0; }
    }
    double gg = 0.0;
    {
        const int r = lane < 15 ? lane : 0;
        if (r < 6) {
            gg = R.g1 + (hasm ? R.g2 : 0.0) + (hasp ? R.g3 : 0.0) + R.g4;
            if (i == 0 && c.mode == LIW_MODE_INIT) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
                int j = 0;
                for (; j + 4 <= n; j += 4) {
                    s0 += PLb[(size_t)j * LP + 108 + r]; s1 += PLb[(size_t)(j + 1) * LP + 108 + r];
                    s2 += PLb[(size_t)(j + 2) * LP + 108 + r]; s3 += PLb[(size_t)(j + 3) * LP + 108 + r];
                }
                for (; j < n; ++j) s0 += PLb[(size_t)j * LP + 108 + r];
                gg += (s0 + s1) + (s2 + s3);
            }
        }
        gg += (hasm ? R.g5 : 0.0) + (hasp ? R.g6 : 0.0);
    }
    const double xq = R.xq;
    if (ex) { ex->sc_i = R.sc_i; ex->sc_m = R.sc_m; ex->dg_i = R.dg_i; ex->x_i = xq; }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = lane + 64 * q;
        const int r = e >> 4, cc = e & 15;
        const bool valid = r < 15 && cc < 15;
        double d = dI[q];
        if (prior_here && valid) d += jtj[q];
        if (LAYOUT == 0) { T_.D[e] = valid ? d : 0.0; T_.O[e] = valid ? oI[q] : 0.0; T_.R[e] = 0.0; }
        else if (valid) { T_.d(r, cc) = d; T_.o(r, cc) = oI[q]; if (r < 6) T_.rr(r, cc) = 0.0; }
    }
    if (lane < 16) {
        double g = lane < 15 ? gg : 0.0;
        if (prior_here && lane < 15) g += gprior;
        if (LAYOUT == 0) T_.g[lane] = g; else if (lane < 15) T_.gg(lane) = g;
    }
    lds_sync();
    if (lane < 36) {
        const int r = lane / 6, cc = lane % 6;
        T_.d(r, cc) += dP; T_.o(r, cc) += oP; T_.rr(r, cc) = rP;
    }
    lds_sync();
    if (arrow1) {   // + rows 0..5 of the IMU block (0,1): frame 0's pose against all 15 entries of frame 1
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = lane + 64 * q;
            if (e < 90) T_.rr(e / 15, e % 15) += R.e1[q];
        }
        lds_sync();
    }
    if (c.mode == LIW_MODE_MARG) return;
    // ---- so3 local parameterisation (identity unless |q| > pi) on the q rows/cols (3..5)
    double Pi[9], Pm[9], P0[9];
    const double qa0 = rdlane(xq, 3), qa1 = rdlane(xq, 4), qa2 = rdlane(xq, 5);
    const double qb0 = rdlane(xq, 16), qb1 = rdlane(xq, 17), qb2 = rdlane(xq, 18);
    const double qc0 = rdlane(xq, 20), qc1 = rdlane(xq, 21), qc2 = rdlane(xq, 22);
    const double pi2 = kPi * kPi;
    bool li = qa0 * qa0 + qa1 * qa1 + qa2 * qa2 > pi2;
    bool lm = hasnb && qb0 * qb0 + qb1 * qb1 + qb2 * qb2 > pi2;
    bool l0 = (i >= 2 || arrow1) && qc0 * qc0 + qc1 * qc1 + qc2 * qc2 > pi2;
    if (li || lm || l0) {   // rare path (|q| > pi), one lane
        li = so3_plus_jac(c.x + (size_t)i * 15 + 3, Pi);
        lm = hasnb && so3_plus_jac(c.x + (size_t)nbf * 15 + 3, Pm);
        l0 = (i >= 2 || arrow1) && so3_plus_jac(c.x + 3, P0);
        if (lane == 0) {
            // kind: 0 = D, 1 = O, 2 = R (rows 0..5 only)
            auto at = [&](int kind, int r, int cc) -> double& { return kind == 0 ? T_.d(r, cc) : (kind == 1 ? T_.o(r, cc) : T_.rr(r, cc)); };
            auto right = [&](int kind, int nrows, const double* P) {   // X[:,3:6] <- X[:,3:6] P
                for (int r = 0; r < nrows; ++r) {
                    double t[3];
                    for (int k = 0; k < 3; ++k) t[k] = at(kind, r, 3) * P[k] + at(kind, r, 4) * P[3 + k] + at(kind, r, 5) * P[6 + k];
                    for (int k = 0; k < 3; ++k) at(kind, r, 3 + k) = t[k];
                }
            };
            auto left = [&](int kind, const double* P) {    // X[3:6,:] <- P^T X[3:6,:]
                for (int cc = 0; cc < 15; ++cc) {
                    double t[3];
                    for (int k = 0; k < 3; ++k) t[k] = P[k] * at(kind, 3, cc) + P[3 + k] * at(kind, 4, cc) + P[6 + k] * at(kind, 5, cc);
                    for (int k = 0; k < 3; ++k) at(kind, 3 + k, cc) = t[k];
                }
            };
            if (li) {
                right(0, 15, Pi); left(0, Pi); right(1, 15, Pi); right(2, 6, Pi);
                double t[3];
                for (int k = 0; k < 3; ++k) t[k] = Pi[k] * T_.gg(3) + Pi[3 + k] * T_.gg(4) + Pi[6 + k] * T_.gg(5);
                for (int k = 0; k < 3; ++k) T_.gg(3 + k) = t[k];
            }
            if (lm) left(1, Pm);
            if (l0) left(2, P0);
        }
        lds_sync();
    }
    // ---- constant parameter blocks (solver.cpp:787-794): drop their rows / columns
    if (c.mode == LIW_MODE_TRACK) {
        for (int e = lane; e < 256; e += 64) {
            const int r = e >> 4, cc = e & 15;
            if (r < 15 && cc < 15) {
                const bool cr = var_is_const(c.mode, c.fast, n, i, r), ccn = var_is_const(c.mode, c.fast, n, i, cc);
                if (cr || ccn) T_.d(r, cc) = 0.0;
                if ((hasnb && var_is_const(c.mode, c.fast, n, nbf, r)) || ccn) T_.o(r, cc) = 0.0;
                if (arrow1 && r < 6 && (var_is_const(c.mode, c.fast, n, 0, r) || ccn)) T_.rr(r, cc) = 0.0;
            }
        }
        if (lane < 15 && var_is_const(c.mode, c.fast, n, i, lane)) T_.gg(lane) = 0.0;
        lds_sync();
    }
}

template <int LAYOUT>
__device__ void assemble_frame(const AsmCtx& c, int i, const Tiles<LAYOUT>& T_, double* tmp,
                               const double* scl = nullptr, const double* dgl = nullptr, FrameExtra* ex = nullptr) {
    const AsmRegs R = asm_issue(c, i, scl, dgl);
    __builtin_amdgcn_sched_barrier(0);   // keep every load in ONE batch: a single memory round trip per frame
    asm_commit<LAYOUT>(c, i, R, T_, tmp, ex);
}

// cost = 1/2 sum r^2 over the residual blocks ceres keeps (blocks whose parameters are all constant are dropped).
// *gchk (optional): a checksum over every gradient slot J^T r of the same partial sums.  Ceres rejects an evaluation whose residuals
// or Jacobians hold a non-finite value (ResidualBlock::Evaluate -> IsEvaluationValid); any such entry makes its block's J^T r
// non-finite (NaN * r = NaN, Inf * 0 = NaN), so isfinite(cost) && isfinite(*gchk) is that test on the fused partial sums — e.g. the
// NaN derivative of norm() at an exactly stationary wheel increment (wheel_factor.h:52,58,63).
__device__ double window_cost(const AsmCtx& c, double* gchk = nullptr) {
    const int lane = threadIdx.x & 63;
    const int n = c.n;
    const double* PLb = c.PL + (size_t)c.b * n * LP;
    const bool pif = c.pif != 0;
    const double* PIb = c.PI + (size_t)c.b * (pif ? (size_t)n * PIFS : (size_t)(n - 1) * PIS);
    const double* PWb = c.PW + (size_t)c.b * (n - 1) * PWS;
    const double* PGb = c.PG + (size_t)c.b * n * PGS;
    const bool track = c.mode == LIW_MODE_TRACK;
    double s = 0.0, gs = 0.0;
    for (int i = lane; i < n; i += 64) {
        // (tracking: only the newest frame has laser / ground blocks with a free parameter.  The older frames' laser records are zero —
        //  or, LinArgs::marg_older, hold their marginalisation-topology sums, which are no part of the tracking problem)
        const bool gon = !(track && i < n - 1);
        if (gon) s += PLb[(size_t)i * LP + 120];
        if (gon) s += PGb[(size_t)i * PGS + PG_C];
        if (gchk) {
            if (gon) {
#pragma unroll
                for (int k = 0; k < 12; ++k) gs += PLb[(size_t)i * LP + 108 + k];
            }
            if (gon) {
#pragma unroll
                for (int k = 0; k < 6; ++k) gs += PGb[(size_t)i * PGS + PG_G(k)];
            }
        }
    }
    for (int k = lane; k < n - 1; k += 64) {
        s += pif ? PIb[(size_t)(k + 1) * PIFS + PIF_C] : PIb[(size_t)k * PIS + PI_C];   // (per-frame records: block (k, k+1)'s cost sits in frame k+1's)
        const bool won = !(track && k < n - 2);
        if (won) s += PWb[(size_t)k * PWS + PW_C];
        if (gchk) {
#pragma unroll
            for (int e = 0; e < 30; ++e)   // g_i of block (k, k+1) | g_j of block (k, k+1)
                gs += pif ? (e < 15 ? PIb[(size_t)k * PIFS + PIF_GI + e] : PIb[(size_t)(k + 1) * PIFS + PIF_GJ + e - 15]) : PIb[(size_t)k * PIS + PI_G + e];
            if (won) {
#pragma unroll
                for (int e = 0; e < 12; ++e) gs += PWb[(size_t)k * PWS + PW_G(e)];
            }
        }
    }
    if (c.prior_on && lane < 15) { const double r = prior_r(c, lane); s += r * r; }
    if (gchk) *gchk = wave_sum(gs);
    return 0.5 * wave_sum(s);
}

// P = X^T Y for two [16(k)][16] LDS tiles with fp64 MFMA 16x16x4; lane l gets P[(l>>4)+4r][l&15] in acc[r]
__device__ __forceinline__ d4 xty16(const double* X, const double* Y) {
    const int lane = threadIdx.x & 63;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = (lane >> 4) + 4 * c;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[k * 16 + (lane & 15)], Y[k * 16 + (lane & 15)], acc, 0, 0, 0);
    }
    return acc;
}

// the same product for 15-row tiles with row strides XS / YS whose k = 15 row is the shared zero row Z
template <int XS, int YS>
__device__ __forceinline__ d4 xty15(const double* X, const double* Y, const double* Z) {
    const int lane = threadIdx.x & 63, m = lane & 15;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = (lane >> 4) + 4 * c;
        const double* xr = X + k * XS;
        const double* yr = Y + k * YS;
        if (c == 3) { const bool z = (lane >> 4) == 3; xr = z ? Z : xr; yr = z ? Z : yr; }
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xr[m], yr[m], acc, 0, 0, 0);
    }
    return acc;
}

struct LdsTiles {   // export / marginalisation kernels (tile layout)
    double D[256], O[256], R[256], W[256], Wa[256], CD[256], CR[256];
    double g[16], Cg[16], y0[16], yprev[16], tmp[ASM_TMP], D0acc[36], g0acc[8], sci[16], scm[16], sc0[16], dgi[16];
};
struct LdsStep {    // k_lm_step (lane layout): M = assembled frame, C = carried Schur terms
    // The MFMA operand tiles live in M's storage (M is dead once the lanes hold their columns): W = L^-1 [O^T | g] (15 rows,
    // stride 16), Li = L^-1, Wa = L^-1 R^T (6 columns), all with row stride 16 so that every lane writes its column with the same
    // 15 immediate offsets (ONE masked store per row).  Their 16th row (k = 15) is the shared zero row Z; columns a product does
    // not use may hold stale words (each output depends on one column of either operand only).
    double M[720], C[15 * MS], Z[16];
    double tmp[ASM_TMP], D0acc[36], g0acc[8];
};
constexpr int LW = 0, LLI = 240, LWA = 480;   // offsets of W / Li / Wa inside LdsStep::M
struct LdsDense2 {  // k_lm_step_dense2: both frames of a two-frame window assembled in the tile layout (+ LdsStep for the shared helpers)
    LdsStep S;
    double D1[256], O1[256], R1[256], D0[256], O0[256], R0[256], g1[16], g0[16];
};
__device__ __forceinline__ LdsStep& step_lds(LdsStep& T) { return T; }
__device__ __forceinline__ LdsStep& step_lds(LdsDense2& T) { return T.S; }

// ---------------------------------------------------------------------------------------------------
// Right-looking Cholesky of the 15x15 matrix whose column j lives in lane j (a[r] = A[r][j]) fused with the forward
// substitution of every other lane's column (right-hand sides): step k broadcasts the pivot, every lane forms
// w_k = a[k]/L_kk (matrix lane j >= k: L[j][k]; rhs lane: (L^-1 b)[k]) and updates a[r] -= L[r][k] w_k with L[r][k]
// read from lane r.  Afterwards matrix lane j holds row j of L in a[0..j]; rhs lanes hold L^-1 b.  No LDS, no barrier.
// NP < 15: only the leading NP pivots are real, the rest are inert unit pivots with no coupling (the hub of k_lm_step_tw): skipping
// them leaves exactly the same registers
template <int NP = 15>
__device__ __forceinline__ bool fused_chol_solve(double (&a)[15]) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < NP; ++k) {
        const double piv = rdlane(a[k], k);
        // a non-positive or non-finite pivot turns 1/sqrt into NaN (or 0 * inf), which every later entry inherits through the rank-1
        // updates: testing the LAST pivot is testing all of them (4 instructions per pivot less on the dependent chain)
        if (k == NP - 1 && (!(piv > 0.0) || !isfinite(piv))) ok = false;
        const double inv = fast_rsqrt(piv);
        const double wk = a[k] * inv;
        a[k] = wk;
#pragma unroll
        for (int r = k + 1; r < 15; ++r) a[r] -= rdlane(wk, r) * wk;
    }
    return ok;
}

// The same fused pass for a DENSE system of 30 unknowns (a two-frame window, k_lm_step_dense2): matrix lanes 0..29, any other lane a
// right-hand side.
// inert (uniform): bit k set = unknown k is a constant of the problem — a unit pivot with a zero row and column (solver.cpp:787-794: the
// older frame's pose while tracking, its biases too in fast mode).  Its step would change nothing (L_kk = 1, every L[r][k] = 0): skipped,
// 159 of the 435 row updates for the six pose entries alone.
__device__ __forceinline__ bool fused_chol_solve30(double (&a)[30], unsigned inert) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 30; ++k) {
        if (k < 15 && ((inert >> k) & 1u)) continue;   // (only frame 0 has constants)
        const double piv = rdlane(a[k], k);
        if (k == 29 && (!(piv > 0.0) || !isfinite(piv))) ok = false;   // (a bad pivot poisons every later one)
        const double inv = fast_rsqrt(piv);
        const double wk = a[k] * inv;
        a[k] = wk;
#pragma unroll
        for (int r = k + 1; r < 30; ++r) a[r] -= rdlane(wk, r) * wk;
    }
    return ok;
}

// diag(H) of frame i in tangent space (lane v < 15 returns H_vv), for the Jacobi scaling fixed at iteration 0
// WAVE_ONLY: called by one wave of a multi-wave work-group (k_lm_step_tw): no work-group barrier
template <bool WAVE_ONLY = false>
__device__ double frame_diag(const AsmCtx& c, int i, LdsStep& T) {
    const int lane = threadIdx.x & 63, n = c.n;
    double Pq[9];
    if (so3_plus_jac(c.x + (size_t)i * 15 + 3, Pq)) {   // rare: |q| > pi -> full tangent assembly
        assemble_frame<1>(c, i, Tiles<1>{T.M, nullptr, nullptr, nullptr}, T.tmp);
        const double d = lane < 15 ? T.M[lane * MS + lane] : 0.0;
        lds_sync();
        return d;
    }
    if (lane >= 15) return 0.0;
    const int r = lane;
    const double* PLb = c.PL + (size_t)c.b * n * LP;
    const double* PIb = c.PI + (size_t)c.b * (c.pif ? (size_t)n * PIFS : (size_t)(n - 1) * PIS);
    const double* PWb = c.PW + (size_t)c.b * (n - 1) * PWS;
    const double* PGb = c.PG + (size_t)c.b * n * PGS;
    double d = 0.0;
    if (r < 6) {
        d += PLb[(size_t)i * LP + 36 + r * 7];
        if (i == 0) for (int j = 0; j < n; ++j) d += PLb[(size_t)j * LP + r * 7];
        if (i >= 1) d += PWb[(size_t)(i - 1) * PWS + PW_JJ(r, r)];
        if (i <= n - 2) d += PWb[(size_t)i * PWS + PW_II(r, r)];
        d += PGb[(size_t)i * PGS + PG_H(r, r)];
    }
    if (c.pif) { if (n > 1) d += PIb[(size_t)i * PIFS + PIF_D + pi_tri(r, r)]; }
    else {
        if (i >= 1) d += PIb[(size_t)(i - 1) * PIS + PI_JJ + pi_tri(r, r)];
        if (i <= n - 2) d += PIb[(size_t)i * PIS + PI_II + pi_tri(r, r)];
    }
    if (c.prior_on && i == n - 2) { double s = 0.0; for (int k = 0; k < 15; ++k) s += c.pJ[k * 15 + r] * c.pJ[k * 15 + r]; d += s; }
    return d;
}

// THROUGHPUT = false: 2 waves per SIMD, the next frame's loads are issued one frame ahead in the elimination sweep (lowest
// latency of one window).  THROUGHPUT = true: 3 waves per SIMD (<= 168 VGPRs, no look-ahead there): the other waves hide
// the round trip instead (large batches).
// DENSE2 (n == 2 only; k_lm_step_dense2): the two-frame window the reference's tracking loop solves every laser frame
// (trajectory.cpp:525-560, solver.cpp:631-820) as ONE dense 30 x 30 system instead of two chained frame eliminations — see the sweep below.
template <bool THROUGHPUT, bool DENSE2 = false, class LDS = LdsStep>
__device__ __forceinline__ void lm_step_body(const StepArgs& a, const int b, LDS& TL) {   // one wave = one window
    constexpr bool LIW_PF1 = !THROUGHPUT, LIW_PF2 = true;
    LdsStep& T = step_lds(TL);
    const int lane = threadIdx.x & 63;
    LmState& st = a.w.lm[b];
    if constexpr (DENSE2) {
        double pf_sink = 0.0;
        int done0 = st.done;
        // The step of a tracking window is a chain of DEPENDENT memory round trips (LM state -> cost slots of the buffer it names -> candidate
        // states -> partial sums / prior / scales), and the producers ran on other XCDs: every trip goes past this XCD's L2 (~1.7 k cycles
        // each, tools/clk_probe_track.py).  All of it is ~15 kB at addresses known up front: one batch of line-touching loads for BOTH
        // partial buffers at entry, one wait, and the dependent loads below hit the vector L1.
        const int n2 = a.n;   // (= 2: every region below is at most 64 lines)
        const bool pr = a.mode == LIW_MODE_TRACK && !a.fast_mode;
        double v[14];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            v[4 * k + 0] = touch_lines(a.w.PL[k] + (size_t)b * n2 * LP, sizeof(double) * n2 * LP);
            v[4 * k + 1] = touch_lines(a.w.PI[k] + (size_t)b * (n2 - 1) * PIS, sizeof(double) * (n2 - 1) * PIS);
            v[4 * k + 2] = touch_lines(a.w.PW[k] + (size_t)b * (n2 - 1) * PWS, sizeof(double) * (n2 - 1) * PWS);
            v[4 * k + 3] = touch_lines(a.w.PG[k] + (size_t)b * n2 * PGS, sizeof(double) * n2 * PGS);
        }
        v[8] = touch_lines(a.x + (size_t)b * n2 * 15, sizeof(double) * n2 * 15);
        v[9] = touch_lines(a.w.x_cand + (size_t)b * n2 * 15, sizeof(double) * n2 * 15);
        v[10] = touch_lines(st.scale, sizeof(double) * n2 * 15);
        v[11] = touch_lines(st.diagonal, sizeof(double) * n2 * 15);
        // (without a prior the two slots touch the states once more: the prior arrays may be null then)
        {
            const double* const xb = a.x + (size_t)b * n2 * 15;
            const double* const PJ = a.prior_J; const double* const PX = a.prior_X;
            const bool pj = pr && PJ != nullptr, px = pr && PX != nullptr;
            const size_t qj = (size_t)(PJ + (size_t)b * 225), qx = (size_t)(PX + (size_t)b * 15), q0 = (size_t)xb;
            v[12] = touch_lines(reinterpret_cast<const void*>(pj ? qj : q0), pj ? sizeof(double) * 225 : sizeof(double) * 8);
            v[13] = touch_lines(reinterpret_cast<const void*>(px ? qx : q0), px ? sizeof(double) * 15 : sizeof(double) * 8);
        }
#pragma unroll
        for (int k = 0; k < 14; ++k) pf_sink += v[k];
        asm volatile("" : "+v"(pf_sink), "+v"(done0));   // ONE wait for the whole batch (and the loads cannot be dropped or sunk below a branch)
        if (done0) return;
    }
    if (st.done) return;
    if (a.only_slow) {   // behind k_lm_step_quad: that kernel marked the windows it stepped; the others (a rotation vector outside |theta| <= pi) are this one's
        const int taken = st.pad_;
        wave_mem_sync();
        if (taken) { if (lane == 0) st.pad_ = 0; return; }
    }
    const int n = a.n;
    double* xw = a.x + (size_t)b * n * 15;
    double* xc = a.w.x_cand + (size_t)b * n * 15;

    AsmCtx c;
    c.pif = a.w.pi_frame;
    c.n = n; c.mode = a.mode; c.fast = a.fast_mode; c.b = b;
    c.pJ = a.prior_J + (size_t)b * 225; c.pX = a.prior_X + (size_t)b * 15;
    c.prior_on = a.mode == LIW_MODE_TRACK && a.has_prior[b] && !a.fast_mode;

    double radius = st.radius, dec = st.decrease_factor, x_cost = st.x_cost, x_norm = st.x_norm;
    // wave-uniform by construction, but loaded with vector loads: made scalar explicitly, so that the partial-buffer base pointers
    // selected by `cur` (and everything derived from them in the frame loop) stay in SGPRs instead of 64-bit VALU address arithmetic
    int reuse = __builtin_amdgcn_readfirstlane(st.reuse_diagonal), iteration = __builtin_amdgcn_readfirstlane(st.iteration),
        cur = __builtin_amdgcn_readfirstlane(st.cur);
    bool last_successful = true;
    bool fresh = false;
    STAMPE(4000);

    if (iteration == 0 && !st.have_candidate) {
        // ---- iteration 0: cost at the initial point
        c.buf = cur; c.PL = a.w.PL[cur]; c.PI = a.w.PI[cur]; c.PW = a.w.PW[cur]; c.PG = a.w.PG[cur]; c.x = xw;
        x_cost = window_cost(c);
        if (lane == 0) { st.initial_cost = x_cost; st.minimum_cost = x_cost; }
        if (!isfinite(x_cost)) {   // IterationZero: "Residual and Jacobian evaluation failed." -> FAILURE, nothing applied (non-finite residual;
            if (lane == 0) { st.done = 1; st.termination = 6; st.x_cost = x_cost; }   // a non-finite Jacobian is caught in the pass below)
            return;
        }
        double s = 0.0;
        for (int e = lane; e < n * 15; e += 64) {
            const double xv = xw[e];
            st.x0[e] = xv;
            if (!var_is_const(a.mode, a.fast_mode, n, e / 15, e % 15)) s += xv * xv;
        }
        x_norm = sqrt(wave_sum(s));
        fresh = true;
        if (a.w.history && a.w.history_records > 0)
            for (int e = lane; e < n * 15; e += 64) a.w.history[((size_t)b) * n * 15 + e] = xw[e];
    } else if (st.have_candidate) {
        // ---- candidate evaluated by the previous linearise launch
        const int cb = 1 - cur;
        c.buf = cb; c.PL = a.w.PL[cb]; c.PI = a.w.PI[cb]; c.PW = a.w.PW[cb]; c.PG = a.w.PG[cb]; c.x = xc;
        double cand_cost = window_cost(c);
        STAMPE(4001);
        if (!isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
        int term = 0;
        if (st.cand_step_norm <= kParamTol * (x_norm + kParamTol)) term = 3;
        else if (fabs(x_cost - cand_cost) <= kFuncTol * x_cost) term = 2;
        if (term) {
            if (a.w.history && iteration < a.w.history_records)
                for (int e = lane; e < n * 15; e += 64) a.w.history[((size_t)iteration * a.B + b) * n * 15 + e] = xw[e];
            if (lane == 0) { st.done = 1; st.termination = term; }
            return;
        }
        const double rho = (x_cost - cand_cost) / st.model_cost_change;
        if (rho > kMinRelDec) {
            for (int e = lane; e < n * 15; e += 64) xw[e] = xc[e];
            cur = cb;
            x_cost = cand_cost;
            double s = 0.0;
            for (int e = lane; e < n * 15; e += 64) if (!var_is_const(a.mode, a.fast_mode, n, e / 15, e % 15)) s += xc[e] * xc[e];
            x_norm = sqrt(wave_sum(s));
            { const double t3 = 2.0 * rho - 1.0; radius = radius / fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3); }   // (not pow(): 200 instructions of one wave)
            radius = fmin(kMaxRadius, radius);
            dec = 2.0; reuse = 0;
            if (lane == 0) { st.successful += 1; if (x_cost < st.minimum_cost) st.minimum_cost = x_cost; }
            last_successful = true;
        } else {
            radius = radius / dec; dec *= 2.0; reuse = 1;
            last_successful = false;
        }
        wave_mem_sync();
        if (a.w.history && iteration < a.w.history_records)
            for (int e = lane; e < n * 15; e += 64) a.w.history[((size_t)iteration * a.B + b) * n * 15 + e] = xw[e];
    } else {
        last_successful = false;   // previous step was invalid
        if (a.w.history && iteration < a.w.history_records)
            for (int e = lane; e < n * 15; e += 64) a.w.history[((size_t)iteration * a.B + b) * n * 15 + e] = xw[e];
    }

    // ---- FinalizeIterationAndCheckIfMinimizerCanContinue, part 1 (the gradient test needs the assembled g and is
    //      applied after the pass below; it can only pre-empt the min-radius exit, never the iteration cap)
    if (iteration >= st.max_iters && !(iteration == 0 && fresh)) {
        if (lane == 0) {
            st.done = 1; st.termination = 4; st.radius = radius; st.decrease_factor = dec; st.x_cost = x_cost; st.x_norm = x_norm;
            st.reuse_diagonal = reuse; st.cur = cur; st.have_candidate = 0;
        }
        return;
    }

    // current linearisation
    STAMPE(4002);
    wave_mem_sync();
    STAMPE(4003);
    c.buf = cur; c.PL = a.w.PL[cur]; c.PI = a.w.PI[cur]; c.PW = a.w.PW[cur]; c.PG = a.w.PG[cur]; c.x = xw;

    if (fresh) {   // Jacobi scaling 1/(1+sqrt(H_jj)), computed once per solve
        for (int i = 0; i < n; ++i) {
            const double hjj = frame_diag(c, i, T);
            if (lane < 15) st.scale[i * 15 + lane] = var_is_const(a.mode, a.fast_mode, n, i, lane) ? 1.0 : 1.0 / (1.0 + sqrt(hjj));
        }
        wave_mem_sync();
    }
    const double* scl = st.scale;
    double* dgl = st.diagonal;
    double* sws = a.w.solve_ws + (size_t)b * n * SOLVE_WS;

    const int iteration_dbg = iteration; (void)iteration_dbg;
    bool solved = true;
    double gmax = 0.0, gsum = 0.0;
    const double inv_radius = 1.0 / radius;   // (one division per step instead of one per frame; 1 ulp from diagonal / radius)
    // DENSE2 state that lives from the elimination to the back substitution (lane e < 30 = unknown e = entry e % 15 of frame e / 15)
    double dcol[DENSE2 ? 30 : 1];
    double d_x = 0.0, d_sc = 1.0, d_dg = 0.0, d_gs = 0.0;
    bool d_cst = false;
    if constexpr (DENSE2) {
        // ---- the whole window as ONE dense system.  Both frames are assembled in the tile layout by the SAME code as everywhere else
        // (asm_issue / asm_commit<0>: tangent space, constants masked, prior, the init topology's arrow folded into the coupling tile,
        // hub sums on frame 0), then: lane e < 30 owns column e of  A = S H S + D^2 / radius  (unit pivots on constant entries), lane 30
        // the scaled gradient, lanes 31 .. 60 the unit vectors; ONE fused Cholesky / forward substitution over 30 pivots (v_readlane
        // broadcasts, no LDS) leaves z = L^-1 g_s in lane 30 and column m of L^-1 in lane 31 + m, so that the solution is 30 FMAs per
        // lane: y_m = (L^-1 e_m) . z.  No Schur products, no factor record, no second sweep through memory: the chained kernel spent
        // 39 k of its 50 k cycles per step on a two-frame window in those (tools/clk_probe_track.py).
        STAMP(0);
        // one memory round trip for everything the step reads: both frames' partial sums, the prior, this lane's state / scale / diagonal
        const int e = lane < 30 ? lane : 0, ei = e / 15, ev = e % 15;
        const AsmRegs r1 = asm_issue(c, 1, nullptr, nullptr);
        const AsmRegs r0 = asm_issue(c, 0, nullptr, nullptr);
        PriorRegs preg;
        if (c.prior_on) preg = prior_issue(c, lane);
        d_x = xw[e]; d_sc = scl[e]; d_dg = dgl[e];
        unsigned inert = 0;
        for (int v = 0; v < 15; ++v) inert |= var_is_const(a.mode, a.fast_mode, n, 0, v) ? (1u << v) : 0u;
        __builtin_amdgcn_sched_barrier(0);
        asm_commit<0>(c, 1, r1, Tiles<0>{TL.D1, TL.O1, TL.R1, TL.g1}, T.tmp);
        STAMP(20);
        asm_commit<0>(c, 0, r0, Tiles<0>{TL.D0, TL.O0, TL.R0, TL.g0}, T.tmp, nullptr, lane, -1, c.prior_on ? &preg : nullptr);
        STAMP(21);
        const double* De = ei ? TL.D1 : TL.D0;
        const double gl = lane < 30 ? (ei ? TL.g1[ev] : TL.g0[ev]) : 0.0;         // tangent gradient entry (constants masked)
        d_cst = lane < 30 && var_is_const(a.mode, a.fast_mode, n, ei, ev);
        gsum = gl;
        if (lane < 30 && !reuse) { d_dg = fmin(fmax(De[ev * 16 + ev] * d_sc * d_sc, kMinDiag), kMaxDiag); dgl[e] = d_dg; }
        d_gs = lane < 30 ? gl * d_sc : 0.0;
        {   // gradient max-norm |x - Plus(x, -g)| (rotation entries through so3 Plus), constants left out
            double m = fabs(gl);
#pragma unroll
            for (int fi = 0; fi < 2; ++fi) {
                const double qv[3] = {rdlane(d_x, 15 * fi + 3), rdlane(d_x, 15 * fi + 4), rdlane(d_x, 15 * fi + 5)};
                const double ng[3] = {-rdlane(gl, 15 * fi + 3), -rdlane(gl, 15 * fi + 4), -rdlane(gl, 15 * fi + 5)};
                double qn[3];
                so3_plus(qv, ng, qn);
                if (lane >= 15 * fi + 3 && lane < 15 * fi + 6) m = fabs(d_x - qn[lane - 15 * fi - 3]);
            }
            if (lane < 30 && !d_cst) gmax = m;
        }
        {
            const double diag_add = d_cst ? 1.0 : d_dg * inv_radius;
            const int cc = lane < 15 ? lane : (lane < 30 ? lane - 15 : 0);
            // H[r][column of this lane]: D0 | O1 (rows: frame 0, columns: frame 1) | its transpose | D1 — per lane two bases and a stride
            const double* const top = (lane < 15 ? TL.D0 : TL.O1) + cc;                      // rows 0 .. 14: element r at top[16 r]
            const double* const bot = lane < 15 ? TL.O1 + cc * 16 : TL.D1 + cc;              // rows 15 .. 29: element r' at bot[bs r']
            const int bs = lane < 15 ? 1 : 16;
            // one formula for every lane: matrix lanes (column scale csc = own scale, diagonal term on row = lane), the gradient lane (m30),
            // the unit-vector lanes (csc = 0, "diagonal term" 1 on row = lane - 31) — as three nested selects per row this loop was ~20
            // instructions per row.  (Row scales by v_readlane: through LDS broadcasts the loop was slower, measured.)
            const double csc = lane < 30 ? d_sc : 0.0, m30 = lane == 30 ? 1.0 : 0.0;
            const int ridx = lane < 30 ? lane : lane - 31;
            const double dadd = lane < 30 ? diag_add : (lane == 30 ? 0.0 : 1.0);
#pragma unroll
            for (int r = 0; r < 30; ++r) {
                const double sr = rdlane(d_sc, r), gr = rdlane(d_gs, r);   // (broadcasts in uniform control flow)
                const double h = r < 15 ? top[r * 16] : bot[(r - 15) * bs];
                dcol[r] = __builtin_fma(gr, m30, h * (sr * csc) + (ridx == r ? dadd : 0.0));
            }
        }
        STAMP(22);
        solved = fused_chol_solve30(dcol, inert);
        STAMP(23);
    } else {
    // ---- single pass: eliminate frames n-1 .. 1, then 0, of (S H S + D^2) y = S g.
        // Register-resident elimination: lane j < 15 owns column j of the damped diagonal tile, lanes 16..30 the columns of
        // O^T, lanes 32..37 the columns of R^T, lane 40 the gradient.  One fused pass (fused_chol_solve) turns the matrix
        // lanes into the rows of L and every right-hand-side lane into L^-1 b, using v_readlane broadcasts only.
        for (int e = lane; e < 720; e += 64) { T.M[e] = 0.0; if (e < 15 * MS) T.C[e] = 0.0; }
        if (lane < 16) T.Z[lane] = 0.0;
        if (lane < 36) T.D0acc[lane] = 0.0;
        if (lane < 8) T.g0acc[lane] = 0.0;
        const double sc0reg = scl[lane < 15 ? lane : 0];           // scale of frame 0 (rows of the arrow block)
        lds_sync();
        STAMP(0); SPAN(0);
        const Tiles<1> TM{T.M, nullptr, nullptr, nullptr};
        AsmRegs areg;
        PriorRegs preg;
        if constexpr (LIW_PF1) areg = asm_issue(c, n - 1, scl, dgl);
        for (int i = n - 1; i >= 0; --i) {
            STAMP(10 + i * 8 + 0);
            // the lane id is laundered once per frame: lane-derived addresses and masks are recomputed (a few integer ops)
            // instead of being hoisted out of the loop into registers that then spill
            int ln = lane;
            asm volatile("" : "+v"(ln));
            FrameExtra ex;
            if constexpr (!LIW_PF1) areg = asm_issue(c, i, scl, dgl, ln);
            asm_commit<1>(c, i, areg, TM, T.tmp, &ex, ln, -1, (LIW_PF1 && c.prior_on && i == n - 2) ? &preg : nullptr);
            STAMP(10 + i * 8 + 1);
            // LM diagonal of this frame (LevenbergMarquardtStrategy::ComputeStep), |x - Plus(x,-g)|
            const bool cstl = ln < 15 && var_is_const(a.mode, a.fast_mode, n, i, ln);
            const double gl = ln < 15 ? T.M[ln * MS + 40] : 0.0;          // tangent gradient entry of this lane
            gsum += gl;
            double dgv = ex.dg_i;
            if (ln < 15) {
                if (!reuse) { dgv = fmin(fmax(T.M[ln * MS + ln] * ex.sc_i * ex.sc_i, kMinDiag), kMaxDiag); dgl[i * 15 + ln] = dgv; }
                sws[(size_t)i * SOLVE_WS + REC_GS + ln] = gl * ex.sc_i;          // original scaled gradient (model decrease)
            }
            {
                const double qv[3] = {rdlane(ex.x_i, 3), rdlane(ex.x_i, 4), rdlane(ex.x_i, 5)};
                const double ng[3] = {-rdlane(gl, 3), -rdlane(gl, 4), -rdlane(gl, 5)};
                double qn[3];
                so3_plus(qv, ng, qn);                                          // uniform: every lane, no divergence
                double m = fabs(gl);
                if (ln >= 3 && ln < 6) m = fabs(ex.x_i - (ln == 3 ? qn[0] : (ln == 4 ? qn[1] : qn[2])));
                if (ln < 15 && !cstl) gmax = fmax(gmax, m);
            }
            STAMP(10 + i * 8 + 2);
            // this lane's column: scale (Jacobi), damp (LM), add the carried Schur terms.  Lane roles: j < 15 column j of
            // the diagonal tile, 16..30 columns of O^T, 32..37 columns of R^T, 40 the gradient.
            // (shuffles run in uniform control flow: ds_bpermute only sees data of active source lanes)
            const double s_m = __shfl(ex.sc_m, (ln - 16) & 63, 64), s_0 = __shfl(sc0reg, (ln - 32) & 63, 64);
            double slane = 0.0;
            if (ln < 15) slane = ex.sc_i;
            else if (ln >= 16 && ln < 31) slane = i >= 1 ? s_m : 0.0;
            else if (ln >= 32 && ln < 38) slane = i >= 2 ? s_0 : 0.0;
            else if (ln == 40) slane = 1.0;
            // LM damping and the unit pivots of constant entries go into the carried-terms tile BEFORE the columns are read: patching
            // col[ln] afterwards is a dynamic register index = 7 instructions per row.  (A constant entry's row / column of M and of
            // the carried terms is zero, and the hub accumulators only exist in the init topology, which has no constants.)
            if (ln < 15) {
                double& cd = T.C[ln * MS + ln];
                cd = cstl ? 1.0 : cd + dgv * inv_radius;
            }
            lds_sync();
            double col[15];
            const int lc = ln < MS ? ln : MS - 1;   // lanes beyond the last column read a valid word they never use
#pragma unroll
            for (int r = 0; r < 15; ++r) col[r] = T.M[r * MS + lc] * (rdlane(ex.sc_i, r) * slane) + T.C[r * MS + lc];
            if (i == 1) {   // frame 0 is both the chain neighbour and the arrow target: fold R^T into O^T
                const bool mg = ln >= 16 && ln < 22;
                const int src = mg ? ln + 16 : lc;
                const double s0 = __shfl(sc0reg, mg ? ln - 16 : 0, 64);
#pragma unroll
                for (int r = 0; r < 15; ++r) {
                    const double v = T.M[r * MS + src] * (rdlane(ex.sc_i, r) * s0) + T.C[r * MS + src];
                    if (mg) col[r] += v;
                    if (ln >= 32 && ln < 38) col[r] = 0.0;
                }
            }
            if (i == 0) {
                if (ln < 6) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) col[r] += T.D0acc[r * 6 + ln];
                }
                if (ln == 40) {
#pragma unroll
                    for (int r = 0; r < 6; ++r) col[r] += T.g0acc[r];
                }
            }
            STAMP(10 + i * 8 + 3);
            // software pipeline: the next frame's loads are in flight while this one is factorised
            if constexpr (LIW_PF1) {
            __builtin_amdgcn_sched_barrier(0);
            if (i > 0) areg = asm_issue(c, i - 1, scl, dgl, ln);
            if (c.prior_on && i - 1 == n - 2) preg = prior_issue(c, ln);
            __builtin_amdgcn_sched_barrier(0);
            }
            // lanes 41..55 carry the unit vectors: the fused pass leaves the columns of L^-1 in them
            if (ln >= 41 && ln < 56) {
#pragma unroll
                for (int r = 0; r < 15; ++r) col[r] = (r == ln - 41) ? 1.0 : 0.0;
            }
            if (!fused_chol_solve(col)) { solved = false; break; }
            STAMP(10 + i * 8 + 4);
            // MFMA operand tiles: W = L^-1 [O^T | g], Wa = L^-1 R^T, Li = L^-1
            {
                const int toff = (ln >= 16 && ln < 31) ? LW + (ln - 16) : ((ln >= 32 && ln < 38) ? LWA + (ln - 32) : (ln == 40 ? LW + 15 : ((ln >= 41 && ln < 56) ? LLI + (ln - 41) : -1)));
                if (toff >= 0) {
#pragma unroll
                    for (int r = 0; r < 15; ++r) T.M[toff + r * 16] = col[r];
                }
            }
            lds_sync();
            STAMP(10 + i * 8 + 5);
            // Back-substitution operators of this frame, y_i = yz - Yo y_{i-1} - Yr y_0 with [Yo | Yr | yz] = D^-1 [O^T | R^T | g]
            // = L^-T (L^-1 [..]): two more products on the matrix cores instead of a second triangular solve, so the record is
            // 22 columns (not L + W: 37) and the second sweep is a matrix-vector product.  rec[r][22]: 0..14 Yo, 15..20 Yr, 21 yz.
            {
                const d4 y1 = xty15<16, 16>(T.M + LLI, T.M + LW, T.Z), y2 = xty15<16, 16>(T.M + LLI, T.M + LWA, T.Z);
                double* f = sws + (size_t)i * SOLVE_WS;
                const int colx = ln & 15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (ln >> 4) + 4 * r;
                    if (row < 15) {
                        f[row * REC_LD + (colx < 15 ? colx : 21)] = y1[r];
                        if (colx < 6) f[row * REC_LD + 15 + colx] = y2[r];
                    }
                }
            }
            d4 p1 = {0.0, 0.0, 0.0, 0.0}, p2 = p1, p3 = p1;
            if (i >= 1) {   // Schur products on the matrix cores
                p1 = xty15<16, 16>(T.M + LW, T.M + LW, T.Z);      // [Wo|z]^T [Wo|z]
                p2 = xty15<16, 16>(T.M + LWA, T.M + LW, T.Z);     // Wr^T [Wo|z]   (rows < 6)
                p3 = xty15<16, 16>(T.M + LWA, T.M + LWA, T.Z);    // Wr^T Wr       (rows, cols < 6)
            }
            if (i >= 1) {
                // written back in the lane layout of the next frame
                lds_sync();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (ln >> 4) + 4 * r, colx = ln & 15;
                    if (row < 15) T.C[row * MS + (colx < 15 ? colx : 40)] = -p1[r];      // diagonal tile (+ gradient, column 15 -> lane 40) of frame i-1
                    if (row < 6 && colx < 15) T.C[colx * MS + 32 + row] = -p2[r];        // arrow block H[0, i-1] (as R^T)
                    if (i >= 2) {
                        if (row < 6 && colx == 15) T.g0acc[row] -= p2[r];
                        if (row < 6 && colx < 6) T.D0acc[row * 6 + colx] -= p3[r];
                    }
                }
                lds_sync();
            }
        }
    }
    STAMP(1);
    gmax = wave_max(gmax);
    // ---- was the evaluation this linearisation came from valid?  Ceres rejects residuals / Jacobians with a non-finite entry
    //      (ResidualBlock::Evaluate -> IsEvaluationValid); any such entry makes its block's J^T r non-finite (NaN * r = NaN, Inf * 0 = NaN),
    //      i.e. the assembled gradient that just went through this wave — e.g. the NaN derivative of norm() at an exactly stationary wheel
    //      increment (wheel_factor.h:52,58,63).  A sweep cut short by a failed pivot re-reads the gradient slots of every frame instead.
    {
        double gchk = wave_sum(gsum);
        if (!solved) (void)window_cost(c, &gchk);
        if (!isfinite(gchk)) {
            // IterationZero / HandleSuccessfulStep: "Residual and Jacobian evaluation failed." -> FAILURE: the iteration is not recorded and
            // the solve hands back the states it started from
            if (!fresh) for (int e = lane; e < n * 15; e += 64) xw[e] = st.x0[e];
            wave_mem_sync();
            if (lane == 0) {
                st.done = 1; st.termination = 6; st.have_candidate = 0; st.cur = cur;
                if (!fresh) { st.iteration = iteration - 1; st.x_cost = st.initial_cost; }
            }
            return;
        }
    }
    // ---- FinalizeIterationAndCheck, part 2
    {
        int term = 0;
        if (iteration == 0 && fresh) { if (gmax <= kGradTol) term = 1; }
        else if (last_successful && gmax <= kGradTol) term = 1;
        if (!term && !(radius > kMinRadius)) term = 5;
        if (term) {
            if (lane == 0) {
                st.done = 1; st.termination = term; st.radius = radius; st.decrease_factor = dec; st.x_cost = x_cost; st.x_norm = x_norm;
                st.reuse_diagonal = reuse; st.cur = cur; st.have_candidate = 0;
            }
            return;
        }
    }
    ++iteration;

    double model_cost_change = 0.0, step_norm = 0.0;
    bool valid = false;
    STAMP(4004);
    wave_mem_sync();   // factor records (global) written above are read by other lanes below
    STAMP(4005);
    if (solved) {
      if constexpr (DENSE2) {
        // y = L^-T z: lane 31 + m holds column m of L^-1, lane 30 holds z
        double yv = 0.0;
#pragma unroll
        for (int k = 0; k < 30; ++k) yv = __builtin_fma(dcol[k], rdlane(dcol[k], 30), yv);
        const double t = __shfl(yv, (lane + 31) & 63, 64);                      // unknown e's solution into lane e
        const bool lv = lane < 30;
        const double del = (lv && !d_cst) ? -t * d_sc : 0.0;
        double xnew = d_x + del;
#pragma unroll
        for (int fi = 0; fi < 2; ++fi) {                                         // so3 Plus on the rotation entries of both frames
            const double qv[3] = {rdlane(d_x, 15 * fi + 3), rdlane(d_x, 15 * fi + 4), rdlane(d_x, 15 * fi + 5)};
            const double dq[3] = {rdlane(del, 15 * fi + 3), rdlane(del, 15 * fi + 4), rdlane(del, 15 * fi + 5)};
            double qn[3];
            so3_plus(qv, dq, qn);
            if (lane >= 15 * fi + 3 && lane < 15 * fi + 6) xnew = var_is_const(a.mode, a.fast_mode, n, fi, 3) ? d_x : qn[lane - 15 * fi - 3];
        }
        double sn2 = 0.0, ytg = 0.0, dsum = 0.0;
        if (lv) {
            xc[lane] = xnew;
            if (!d_cst) { sn2 = (d_x - xnew) * (d_x - xnew); ytg = t * d_gs; dsum = d_dg * inv_radius * t * t; }
        }
        step_norm = sqrt(wave_sum(sn2));
        model_cost_change = 0.5 * (wave_sum(ytg) + wave_sum(dsum));
        valid = model_cost_change > 0.0 && isfinite(model_cost_change);
      } else {
        // ---- back substitution, frame 0 first.  Lane r owns unknown r: row r of Yo / Yr.
        double ytg = 0.0, dsum = 0.0, sn2 = 0.0;
        double yprev = 0.0, y0v = 0.0;   // lane r < 15: y_{i-1}[r], y_0[r]
        STAMP(2);
        struct BsRegs { double t, Yo[15], Yr[6], gsv, xold, scv, dgv; };
        // one batch of loads per frame: yz, row r of Yo / Yr, scaled gradient, state, scale, LM diagonal
        auto bs_issue = [&](int i) {
            const double* f = sws + (size_t)i * SOLVE_WS;
            const int r = lane < 15 ? lane : 0;
            BsRegs R;
            // a record row is 22 consecutive doubles on a 16-byte boundary (REC_LD and SOLVE_WS are even, the workspace is 256-byte
            // aligned): 11 128-bit loads instead of 22 64-bit ones
            static_assert(REC_LD % 2 == 0 && SOLVE_WS % 2 == 0, "16-byte aligned record rows");
            const double2* f2 = reinterpret_cast<const double2*>(f + r * REC_LD);
            double row[22];
#pragma unroll
            for (int k = 0; k < 11; ++k) { const double2 v = f2[k]; row[2 * k] = v.x; row[2 * k + 1] = v.y; }
            R.t = row[21];
#pragma unroll
            for (int k = 0; k < 15; ++k) R.Yo[k] = row[k];
#pragma unroll
            for (int k = 0; k < 6; ++k) R.Yr[k] = row[15 + k];
            R.gsv = f[REC_GS + r];
            R.xold = xw[(size_t)i * 15 + r];
            R.scv = scl[i * 15 + r];
            R.dgv = dgl[i * 15 + r];
            return R;
        };
        BsRegs cur;
        if (LIW_PF2) cur = bs_issue(0);
        for (int i = 0; i < n; ++i) {
            if (!LIW_PF2) cur = bs_issue(i);
            STAMP(300 + i * 4);
            // software pipeline: frame i+1's record is in flight while frame i is solved
            BsRegs nxt;
            if (LIW_PF2) {
            nxt = cur;
            __builtin_amdgcn_sched_barrier(0);
            if (i + 1 < n) nxt = bs_issue(i + 1);
            __builtin_amdgcn_sched_barrier(0);
            }
            double t = cur.t;
            const double* Yo = cur.Yo;
            const double* Yr = cur.Yr;
            const double gsv = cur.gsv, xold = cur.xold, scv = cur.scv, dgv = cur.dgv;
            if (i >= 1) {
#pragma unroll
                for (int k = 0; k < 15; ++k) t -= Yo[k] * rdlane(yprev, k);
            }
            if (i >= 2) {
#pragma unroll
                for (int k = 0; k < 6; ++k) t -= Yr[k] * rdlane(y0v, k);
            }
            STAMP(300 + i * 4 + 1);
            yprev = t;
            if (i == 0) y0v = t;
            // candidate of this frame: delta = -y * scale ; Plus
            const bool cst = lane < 15 && var_is_const(a.mode, a.fast_mode, n, i, lane);
            const double del = (lane < 15 && !cst) ? -t * scv : 0.0;
            const double qv[3] = {rdlane(xold, 3), rdlane(xold, 4), rdlane(xold, 5)};
            double dq[3] = {rdlane(del, 3), rdlane(del, 4), rdlane(del, 5)}, qn[3];
            so3_plus(qv, dq, qn);
            if (lane < 15) {
                double xnew = xold + del;
                if (lane >= 3 && lane < 6) xnew = var_is_const(a.mode, a.fast_mode, n, i, 3) ? xold : qn[lane - 3];
                xc[(size_t)i * 15 + lane] = xnew;
                if (!cst) {
                    sn2 += (xold - xnew) * (xold - xnew);
                    ytg += t * gsv;
                    dsum += dgv * inv_radius * t * t;
                }
            }
            if (LIW_PF2) cur = nxt;
        }
        STAMP(3); SPAN(1);
        step_norm = sqrt(wave_sum(sn2));
        // model cost change -(s'g_s + s'A s/2) with s = -y and (A + D^2) y = g_s  ==  (y'g_s + y'D^2 y)/2
        model_cost_change = 0.5 * (wave_sum(ytg) + wave_sum(dsum));
        valid = model_cost_change > 0.0 && isfinite(model_cost_change);
      }
    }
    // max_num_consecutive_invalid_steps reached: FAILURE, which hands back the states the solve started from
    const bool fail5 = !valid && st.invalid_steps + 1 >= 5;
    if (fail5 && st.successful > 0) {
        for (int e = lane; e < n * 15; e += 64) xw[e] = st.x0[e];
        x_cost = st.initial_cost;
    }
    wave_mem_sync();   // (every lane has read st.invalid_steps / st.successful before lane 0 updates the state)
    if (lane == 0) {
        st.iteration = iteration; st.cur = cur; st.x_cost = x_cost; st.x_norm = x_norm;
        if (valid) {
            st.radius = radius; st.decrease_factor = dec; st.reuse_diagonal = 1;
            st.model_cost_change = model_cost_change; st.cand_step_norm = step_norm; st.have_candidate = 1; st.invalid_steps = 0;
        } else {
            st.invalid_steps += 1;
            st.have_candidate = 0;
            if (fail5) { st.done = 1; st.termination = 6; st.iteration = iteration - 1; }
            st.radius = radius / dec; st.decrease_factor = dec * 2.0; st.reuse_diagonal = 1;
        }
    }
    STAMP(4006);
}
template <bool THROUGHPUT>
// (THROUGHPUT was built for three waves per SIMD — 168 registers, 60 B of scratch — while it carried the large batches; those run
//  k_lm_step_quad now, the instantiation only takes the windows that kernel hands over: two waves, 208 registers, no scratch)
__global__ __launch_bounds__(64, 2) void k_lm_step(StepArgs a) {
    __shared__ LdsStep T;
    if ((int)blockIdx.x >= a.B) return;
    lm_step_body<THROUGHPUT>(a, (int)blockIdx.x, T);
}

// Behind k_lm_step_quad: the windows that kernel left out of this launch (a rotation vector outside the |theta| <= pi ball; it marked the
// ones it stepped in LmState::pad_) are stepped by the one-wave body.  One wave LOOKS at 64 windows (a lane each) and then takes the few
// that are left, one after the other: as a launch of one wave per window (until late round 5) the 49 152 waves of the bench batch, nearly
// all of which returned at once, kept the chip busy for 27 us of every LM iteration.
__global__ __launch_bounds__(64, 1) void k_lm_step_slow(StepArgs a) {
    __shared__ LdsStep T;
    const int lane = threadIdx.x & 63, b = (int)blockIdx.x * 64 + lane;
    bool mine = false;
    if (b < a.B) {
        LmState& st = a.w.lm[b];
        if (!st.done) { if (st.pad_) st.pad_ = 0; else mine = true; }
    }
    unsigned long long m = __ballot(mine);
    if (!m) return;
    wave_mem_sync();
    // (a.only_slow stays set — a modified copy of the kernel arguments would live in scratch —: the body's own look at pad_ finds 0)
    while (m) {
        const int l = __builtin_amdgcn_readfirstlane(__builtin_ctzll(m));
        m &= m - 1;
        lm_step_body<true>(a, (int)blockIdx.x * 64 + l, T);
        lds_sync();
    }
}

// The two-frame window as one dense system (lm_step_body<.., DENSE2>): the steady-state tracking frame of the reference's front end.
__global__ __launch_bounds__(64, 1) void k_lm_step_dense2(StepArgs a) {
    __shared__ LdsDense2 T;
    if ((int)blockIdx.x >= a.B) return;
    lm_step_body<false, true, LdsDense2>(a, (int)blockIdx.x, T);
}

// ---------------------------------------------------------------------------------------------------
// Two-wave ("twisted") LM step for small batches — the latency of ONE window (the reference's own call pattern: one
// ceres::Solve at a time, solver.cpp:168, :802).  The serial chain of k_lm_step is n dependent 15x15 eliminations; here the chain
// is cut in the middle frame m and eliminated from both ends at once by the two waves of a work-group:
//   wave 0:  frames n-1, n-2, .., m+1            (neighbour i-1, exactly the steps of k_lm_step)
//   wave 1:  frame 0's non-pose entries ("0c"), then frames 1, 2, .., m-1   (neighbour i+1)
//   wave 0:  frame m (both neighbours gone), then the hub = frame 0's pose, which every laser frame's arrow points to
// Frame 0 is split because its velocity / bias entries couple to frame 1 through the IMU block: eliminated first (as a pseudo frame
// whose pose entries are inert unit pivots) they leave a chain 1 .. n-1 whose only common neighbour is the 6-entry hub, so both
// sweeps carry the same 6-wide arrow as before and the upward sweep creates no wider fill.  Back substitution runs hub -> m -> both
// halves outwards, again one half per wave.  Same arithmetic as k_lm_step up to the order of the Schur updates (round-off).
struct LdsTwSide {               // one sweep: a producer wave assembles / scales frame s+1 while the eliminator works on frame s
    double PM[720], Ptmp[ASM_TMP];    // producer: assembled frame (lane layout of k_lm_step)
    double PT[2][15 * MS];       // prepared columns, double-buffered: S H S + D^2 of the frame in lane layout, constants / inert entries as unit pivots
    LdsStep T;                   // eliminator: MFMA operand tiles (T.M), carried Schur terms (T.C), hub accumulators
};
struct LdsTw {
    LdsTwSide side[2];
    double Haa[36], ga[8];        // frame 0: pose block / pose gradient of the assembled frame (step 0c -> hub)
    double sc0[16], dg0[16];      // frame 0: Jacobi scale, LM diagonal
    double ya[8];                 // solution of the hub
    double red[4][4];             // per wave: step norm^2, y'g, y'D^2y, gradient max-norm
    double ctld[4];               // prologue -> all waves: radius
    int ctl[8];                   // proceed, reuse, cur, solved[2], termination
};

// 256 threads: waves 0 / 1 eliminate the downward / upward sweep (k_lm_step's chain cut in the middle, see above), waves 2 / 3 are their
// producers.  Per step and sweep, the part of the work that does not depend on the carried Schur terms — assembling the frame from the
// partial sums, so3 / constant masks, LM diagonal, gradient norm, Jacobi scaling, damping — runs one step ahead in the producer; the
// eliminator only adds the carried terms, factorises, forms the back-substitution record and the Schur products.  One work-group
// barrier per step hands the prepared columns over.
__global__ __launch_bounds__(256, 1) void k_lm_step_tw(StepArgs a) {
    __shared__ LdsTw S;
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (b >= a.B) return;
    LmState& st = a.w.lm[b];
    if (st.done) return;
    const int n = a.n;
    double* xw = a.x + (size_t)b * n * 15;
    double* xc = a.w.x_cand + (size_t)b * n * 15;
    const int sd = w & 1;                       // sweep: 0 downward (frames n-1 .. m+1, then m and the hub), 1 upward (0c, 1 .. m-1)
    const bool producer = w >= 2;
    LdsTwSide& SS = S.side[sd];
    LdsStep& T = SS.T;

    AsmCtx c;
    c.pif = a.w.pi_frame;
    c.n = n; c.mode = a.mode; c.fast = a.fast_mode; c.b = b;
    c.pJ = a.prior_J + (size_t)b * 225; c.pX = a.prior_X + (size_t)b * 15;
    c.prior_on = a.mode == LIW_MODE_TRACK && a.has_prior[b] && !a.fast_mode;

    double radius = 0.0, dec = 0.0, x_cost = 0.0, x_norm = 0.0;
    int reuse = 0, iteration = 0, cur = 0;
    bool last_successful = true, fresh = false;

    // ---- prologue on wave 0 (the candidate test / accept / reject logic of k_lm_step, statement for statement)
    if (w == 0) {
        radius = st.radius; dec = st.decrease_factor; x_cost = st.x_cost; x_norm = st.x_norm;
        reuse = st.reuse_diagonal; iteration = st.iteration; cur = st.cur;
        int proceed = 1;
        if (iteration == 0 && !st.have_candidate) {
            c.buf = cur; c.PL = a.w.PL[cur]; c.PI = a.w.PI[cur]; c.PW = a.w.PW[cur]; c.PG = a.w.PG[cur]; c.x = xw;
            double gchk;
            x_cost = window_cost(c, &gchk);
            if (lane == 0) { st.initial_cost = x_cost; st.minimum_cost = x_cost; }
            double sq = 0.0;
            for (int e = lane; e < n * 15; e += 64) {
                const double xv = xw[e];
                st.x0[e] = xv;
                if (!var_is_const(a.mode, a.fast_mode, n, e / 15, e % 15)) sq += xv * xv;
            }
            x_norm = sqrt(wave_sum(sq));
            fresh = true;
            if (a.w.history && a.w.history_records > 0)
                for (int e = lane; e < n * 15; e += 64) a.w.history[((size_t)b) * n * 15 + e] = xw[e];
            if (!isfinite(x_cost) || !isfinite(gchk)) {   // IterationZero: evaluation failed -> FAILURE, nothing applied (see k_lm_step)
                if (lane == 0) { st.done = 1; st.termination = 6; st.x_cost = x_cost; }
                proceed = 0;
            }
        } else if (st.have_candidate) {
            const int cb = 1 - cur;
            c.buf = cb; c.PL = a.w.PL[cb]; c.PI = a.w.PI[cb]; c.PW = a.w.PW[cb]; c.PG = a.w.PG[cb]; c.x = xc;
            double cand_gchk;
            double cand_cost = window_cost(c, &cand_gchk);
            if (!isfinite(cand_cost)) cand_cost = 1.7976931348623157e308;
            int term = 0;
            if (st.cand_step_norm <= kParamTol * (x_norm + kParamTol)) term = 3;
            else if (fabs(x_cost - cand_cost) <= kFuncTol * x_cost) term = 2;
            const double rho = term ? 0.0 : (x_cost - cand_cost) / st.model_cost_change;
            if (term) {
                if (a.w.history && iteration < a.w.history_records)
                    for (int e = lane; e < n * 15; e += 64) a.w.history[((size_t)iteration * a.B + b) * n * 15 + e] = xw[e];
                if (lane == 0) { st.done = 1; st.termination = term; }
                proceed = 0;
            } else if (rho > kMinRelDec && !isfinite(cand_gchk)) {   // evaluation failed at the accepted point -> FAILURE (see k_lm_step)
                if (st.successful > 0)
                    for (int e = lane; e < n * 15; e += 64) xw[e] = st.x0[e];
                wave_mem_sync();
                if (lane == 0) { st.done = 1; st.termination = 6; st.iteration = iteration - 1; st.x_cost = st.initial_cost; st.have_candidate = 0; }
                proceed = 0;
            } else {
                if (rho > kMinRelDec) {
                    for (int e = lane; e < n * 15; e += 64) xw[e] = xc[e];
                    cur = cb;
                    x_cost = cand_cost;
                    double sq = 0.0;
                    for (int e = lane; e < n * 15; e += 64) if (!var_is_const(a.mode, a.fast_mode, n, e / 15, e % 15)) sq += xc[e] * xc[e];
                    x_norm = sqrt(wave_sum(sq));
                    { const double t3 = 2.0 * rho - 1.0; radius = radius / fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3); }   // (not pow(): 200 instructions of one wave)
                    radius = fmin(kMaxRadius, radius);
                    dec = 2.0; reuse = 0;
                    if (lane == 0) { st.successful += 1; if (x_cost < st.minimum_cost) st.minimum_cost = x_cost; }
                    last_successful = true;
                } else {
                    radius = radius / dec; dec *= 2.0; reuse = 1;
                    last_successful = false;
                }
                if (a.w.history && iteration < a.w.history_records)
                    for (int e = lane; e < n * 15; e += 64) a.w.history[((size_t)iteration * a.B + b) * n * 15 + e] = xw[e];
            }
        } else {
            last_successful = false;   // previous step was invalid
            if (a.w.history && iteration < a.w.history_records)
                for (int e = lane; e < n * 15; e += 64) a.w.history[((size_t)iteration * a.B + b) * n * 15 + e] = xw[e];
        }
        if (proceed && iteration >= st.max_iters && !(iteration == 0 && fresh)) {
            if (lane == 0) {
                st.done = 1; st.termination = 4; st.radius = radius; st.decrease_factor = dec; st.x_cost = x_cost; st.x_norm = x_norm;
                st.reuse_diagonal = reuse; st.cur = cur; st.have_candidate = 0;
            }
            proceed = 0;
        }
        if (proceed && fresh) {   // Jacobi scaling 1/(1+sqrt(H_jj)), computed once per solve
            c.buf = cur; c.PL = a.w.PL[cur]; c.PI = a.w.PI[cur]; c.PW = a.w.PW[cur]; c.PG = a.w.PG[cur]; c.x = xw;
            for (int i = 0; i < n; ++i) {
                const double hjj = frame_diag<true>(c, i, T);
                if (lane < 15) st.scale[i * 15 + lane] = var_is_const(a.mode, a.fast_mode, n, i, lane) ? 1.0 : 1.0 / (1.0 + sqrt(hjj));
            }
        }
        if (lane == 0) { S.ctl[0] = proceed; S.ctl[1] = reuse; S.ctl[2] = cur; S.ctld[0] = radius; }
    }
    __syncthreads();   // (drains wave 0's global writes: accepted states, laser partial copy, scales)
    if (!S.ctl[0]) return;
    reuse = S.ctl[1]; cur = S.ctl[2]; radius = S.ctld[0];
    c.buf = cur; c.PL = a.w.PL[cur]; c.PI = a.w.PI[cur]; c.PW = a.w.PW[cur]; c.PG = a.w.PG[cur]; c.x = xw;
    const double* scl = st.scale;
    double* dgl = st.diagonal;
    double* sws = a.w.solve_ws + (size_t)b * n * SOLVE_WS;

    if (!producer) {
        for (int e = lane; e < 720; e += 64) { T.M[e] = 0.0; if (e < 15 * MS) T.C[e] = 0.0; }
        if (lane < 16) T.Z[lane] = 0.0;
        if (lane < 36) T.D0acc[lane] = 0.0;
        if (lane < 8) T.g0acc[lane] = 0.0;
    } else {
        for (int e = lane; e < 720; e += 64) SS.PM[e] = 0.0;
    }
    const double sc0reg = scl[lane < 15 ? lane : 0];           // scale of frame 0 (rows of the arrow block)
    lds_sync();
    bool solved = true;
    double gmax = 0.0;
    // m <= nA: frame m (step nA of the downward sweep) must come after the last step (m - 1) of the upward sweep
    const int m = (n - 1) / 2, nA = n - 1 - m;
    const int nsteps = sd == 0 ? nA + 2 : m, NS = nA + 2;
    // step s of a sweep: frame i, neighbour direction dir (0: none), kind 0 frame / 1 = 0c / 2 hub / 3 middle
    auto sched = [&](int s_, int& i_, int& dir_, int& kind_) {
        if (sd == 0) {
            if (s_ < nA) { i_ = n - 1 - s_; dir_ = -1; kind_ = 0; }
            else if (s_ == nA) { i_ = m; dir_ = 0; kind_ = 3; }
            else { i_ = 0; dir_ = 0; kind_ = 2; }
        } else {
            i_ = s_; dir_ = 1; kind_ = s_ == 0 ? 1 : 0;
        }
    };
    const Tiles<1> TP{SS.PM, nullptr, nullptr, nullptr};
    AsmRegs areg;
    if (producer) {
        int i0, d0, k0;
        sched(0, i0, d0, k0);
        areg = asm_issue(c, i0, scl, dgl, lane, d0 > 0 ? 1 : -1);
    }
    // ---- producer: columns of step s_ into PT[s_ & 1]
    auto prepare = [&](int s_) {
        int i, dir, kind;
        sched(s_, i, dir, kind);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        FrameExtra ex;
        if (kind != 2) {
            asm_commit<1>(c, i, areg, TP, SS.Ptmp, &ex, ln, dir > 0 ? 1 : -1);
        } else {   // hub: frame 0's pose block and gradient as saved at step 0c, nothing else
            for (int e = ln; e < 15 * MS; e += 64) SS.PM[e] = 0.0;
            lds_sync();
            if (ln < 36) SS.PM[(ln / 6) * MS + ln % 6] = S.Haa[ln];
            if (ln < 6) SS.PM[ln * MS + 40] = S.ga[ln];
            lds_sync();
            ex.sc_i = S.sc0[ln < 15 ? ln : 15]; ex.sc_m = 1.0; ex.dg_i = S.dg0[ln < 15 ? ln : 15]; ex.x_i = xw[ln < 15 ? ln : 0];
        }
        // software pipeline: the loads of the next step's frame are in flight during the rest of this one
        __builtin_amdgcn_sched_barrier(0);
        AsmRegs nreg = areg;
        if (s_ + 1 < nsteps) {
            int i2, d2, k2;
            sched(s_ + 1, i2, d2, k2);
            if (k2 != 2) nreg = asm_issue(c, i2, scl, dgl, ln, d2 > 0 ? 1 : -1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // LM diagonal of this frame (LevenbergMarquardtStrategy::ComputeStep), |x - Plus(x,-g)|
        const bool cstl = ln < 15 && var_is_const(a.mode, a.fast_mode, n, i, ln);
        double dgv = ex.dg_i;
        if (kind != 2) {
            const double gl = ln < 15 ? SS.PM[ln * MS + 40] : 0.0;          // tangent gradient entry of this lane
            if (ln < 15) {
                if (!reuse) { dgv = fmin(fmax(SS.PM[ln * MS + ln] * ex.sc_i * ex.sc_i, kMinDiag), kMaxDiag); dgl[i * 15 + ln] = dgv; }
                sws[(size_t)i * SOLVE_WS + REC_GS + ln] = gl * ex.sc_i;          // original scaled gradient (model decrease)
            }
            const double qv[3] = {rdlane(ex.x_i, 3), rdlane(ex.x_i, 4), rdlane(ex.x_i, 5)};
            const double ng[3] = {-rdlane(gl, 3), -rdlane(gl, 4), -rdlane(gl, 5)};
            double qn[3];
            so3_plus(qv, ng, qn);
            double mg = fabs(gl);
            if (ln >= 3 && ln < 6) mg = fabs(ex.x_i - (ln == 3 ? qn[0] : (ln == 4 ? qn[1] : qn[2])));
            if (ln < 15 && !cstl) gmax = fmax(gmax, mg);
        }
        if (kind == 1) {
            // frame 0 -> pseudo frame "0c": keep the pose block / gradient for the hub, turn H[pose, rest] into the arrow (R^T rows
            // = the hub's entries), make the pose entries inert (zero rows / columns, unit pivots through the constant mask below)
            if (ln < 36) S.Haa[ln] = SS.PM[(ln / 6) * MS + ln % 6];
            if (ln < 6) S.ga[ln] = SS.PM[ln * MS + 40];
            if (ln < 15) { S.sc0[ln] = ex.sc_i; S.dg0[ln] = dgv; }
            if (ln == 15) { S.sc0[15] = 1.0; S.dg0[15] = 0.0; }
            double rv[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {   // R^T(cc, r) = D_0(r, cc), r < 6 <= cc
                const int e = ln + 64 * q, r = e / 15, cc = e % 15;
                rv[q] = (e < 90 && cc >= 6) ? SS.PM[r * MS + cc] : 0.0;
            }
            lds_sync();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = ln + 64 * q, r = e / 15, cc = e % 15;
                if (e < 90) SS.PM[cc * MS + 32 + r] = rv[q];
            }
            for (int e = ln; e < 6 * MS; e += 64) {             // rows 0..5 of every tile (D, O^T, R^T, g)
                const int cc = e % MS;
                if (cc < 15 || (cc >= 16 && cc < 31) || cc == 40) SS.PM[e] = 0.0;
            }
            for (int e = ln; e < 15 * 6; e += 64) SS.PM[(e / 6) * MS + e % 6] = 0.0;   // columns 0..5 of D
            lds_sync();
        }
        const bool dummy = (kind == 1 && ln < 6) || (kind == 2 && ln >= 6 && ln < 15);
        const bool cst2 = cstl || dummy;
        const bool hasnb = kind == 1 || (kind == 0 && (dir > 0 ? i <= n - 2 : i >= 1));
        const bool hasarrow = kind == 1 || ((kind == 0 || kind == 3) && (i >= 2 || (dir > 0 && i == 1)));
        const double s_m = __shfl(ex.sc_m, (ln - 16) & 63, 64), s_0 = __shfl(sc0reg, (ln - 32) & 63, 64);
        double slane = 0.0;
        if (ln < 15) slane = ex.sc_i;
        else if (ln >= 16 && ln < 31) slane = hasnb ? s_m : 0.0;
        else if (ln >= 32 && ln < 38) slane = hasarrow ? s_0 : 0.0;
        else if (ln == 40) slane = 1.0;
        const int lc = ln < MS ? ln : MS - 1;
        double* PT = SS.PT[s_ & 1];
        const double dmp = cst2 ? 0.0 : dgv / radius;
#pragma unroll
        for (int r = 0; r < 15; ++r) {
            double v = SS.PM[r * MS + lc] * (rdlane(ex.sc_i, r) * slane);
            if (ln < 15 && r == ln) v = cst2 ? 1.0 : v + dmp;
            if (ln < MS) PT[r * MS + ln] = v;
        }
        areg = nreg;
    };
    // ---- eliminator: step s_ from PT[s_ & 1]
    auto eliminate = [&](int s_) {
        int i, dir, kind;
        sched(s_, i, dir, kind);
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lc = ln < MS ? ln : MS - 1;
        const double* PT = SS.PT[s_ & 1];
        double col[15];
        if (kind == 3) {   // frame m closes both sweeps: the Schur terms of the upward sweep as well
            const double* C2 = S.side[1].T.C;
#pragma unroll
            for (int r = 0; r < 15; ++r) col[r] = PT[r * MS + lc] + (T.C[r * MS + lc] + C2[r * MS + lc]);
        } else if (kind == 2) {   // the hub takes its carried terms from the accumulators of both sweeps
#pragma unroll
            for (int r = 0; r < 15; ++r) col[r] = PT[r * MS + lc];
            if (ln < 6) {
#pragma unroll
                for (int r = 0; r < 6; ++r) col[r] += S.side[0].T.D0acc[r * 6 + ln] + S.side[1].T.D0acc[r * 6 + ln];
            }
            if (ln == 40) {
#pragma unroll
                for (int r = 0; r < 6; ++r) col[r] += S.side[0].T.g0acc[r] + S.side[1].T.g0acc[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 15; ++r) col[r] = PT[r * MS + lc] + T.C[r * MS + lc];
        }
        // lanes 41..55 carry the unit vectors: the fused pass leaves the columns of L^-1 in them
        if (ln >= 41 && ln < 56) {
#pragma unroll
            for (int r = 0; r < 15; ++r) col[r] = (r == ln - 41) ? 1.0 : 0.0;
        }
        if (!(kind == 2 ? fused_chol_solve<6>(col) : fused_chol_solve<15>(col))) solved = false;   // no early exit: every wave keeps meeting at the barriers
        {
            const int toff = (ln >= 16 && ln < 31) ? LW + (ln - 16) : ((ln >= 32 && ln < 38) ? LWA + (ln - 32) : (ln == 40 ? LW + 15 : ((ln >= 41 && ln < 56) ? LLI + (ln - 41) : -1)));
            if (toff >= 0) {
#pragma unroll
                for (int r = 0; r < 15; ++r) T.M[toff + r * 16] = col[r];
            }
        }
        lds_sync();
        {   // back-substitution operators [Yo | Yr | yz] = D^-1 [O^T | R^T | g] of this frame (the hub keeps its solution in LDS)
            const d4 y1 = xty15<16, 16>(T.M + LLI, T.M + LW, T.Z), y2 = xty15<16, 16>(T.M + LLI, T.M + LWA, T.Z);
            double* f = sws + (size_t)i * SOLVE_WS;
            const int colx = ln & 15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (ln >> 4) + 4 * r;
                if (row < 15) {
                    if (kind != 2) {
                        f[row * REC_LD + (colx < 15 ? colx : 21)] = y1[r];
                        if (colx < 6) f[row * REC_LD + 15 + colx] = y2[r];
                    } else if (colx == 15 && row < 6) {
                        S.ya[row] = y1[r];
                    }
                }
            }
        }
        if (kind != 2) {   // Schur products on the matrix cores
            const d4 p1 = xty15<16, 16>(T.M + LW, T.M + LW, T.Z);      // [Wo|z]^T [Wo|z]
            const d4 p2 = xty15<16, 16>(T.M + LWA, T.M + LW, T.Z);     // Wr^T [Wo|z]   (rows < 6)
            const d4 p3 = xty15<16, 16>(T.M + LWA, T.M + LWA, T.Z);    // Wr^T Wr       (rows, cols < 6)
            lds_sync();
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = (ln >> 4) + 4 * r, colx = ln & 15;
                if (kind != 3) {   // carried into the neighbour, in its lane layout
                    if (row < 15) T.C[row * MS + (colx < 15 ? colx : 40)] = -p1[r];
                    if (row < 6 && colx < 15) T.C[colx * MS + 32 + row] = -p2[r];
                }
                if (row < 6 && colx == 15) T.g0acc[row] -= p2[r];
                if (row < 6 && colx < 6) T.D0acc[row * 6 + colx] -= p3[r];
            }
            lds_sync();
        }
    };
    if (producer) prepare(0);
    __syncthreads();
    for (int s_ = 0; s_ < NS; ++s_) {
#ifdef LIW_CLK
        const long long tw0 = clock64();
#endif
        if (producer) { if (s_ + 1 < nsteps) prepare(s_ + 1); }
        else if (s_ < nsteps) eliminate(s_);
#ifdef LIW_CLK
        const long long tw1 = clock64();
#endif
        __syncthreads();
#ifdef LIW_CLK
        if (b == 0 && (threadIdx.x & 63) == 0 && s_ < 40) {   // per wave and step: busy, then waiting at the barrier
            g_clk[7000 + 100 * (int)(threadIdx.x >> 6) + 2 * s_] = tw1 - tw0;
            g_clk[7001 + 100 * (int)(threadIdx.x >> 6) + 2 * s_] = clock64() - tw1;
        }
#endif
    }
    gmax = wave_max(gmax);
    if (lane == 0) { S.red[w][3] = gmax; if (!producer) S.ctl[3 + sd] = solved ? 1 : 0; }
    __syncthreads();                                             // records of both sweeps, hub solution, flags
    // ---- FinalizeIterationAndCheck, part 2 (iteration / fresh / last_successful live on wave 0; the others follow through LDS)
    {
        if (w == 0) {
            const double gm = fmax(S.red[2][3], S.red[3][3]);
            int term = 0;
            if (iteration == 0 && fresh) { if (gm <= kGradTol) term = 1; }
            else if (last_successful && gm <= kGradTol) term = 1;
            if (!term && !(radius > kMinRadius)) term = 5;
            if (lane == 0) S.ctl[5] = term;
        }
        __syncthreads();
        const int term = S.ctl[5];
        if (term) {
            if (w == 0 && lane == 0) {
                st.done = 1; st.termination = term; st.radius = radius; st.decrease_factor = dec; st.x_cost = x_cost; st.x_norm = x_norm;
                st.reuse_diagonal = reuse; st.cur = cur; st.have_candidate = 0;
            }
            return;
        }
    }
    ++iteration;
    const bool all_solved = S.ctl[3] && S.ctl[4];
    double sn2 = 0.0, ytg = 0.0, dsum = 0.0;
    if (all_solved && !producer) {
        // ---- back substitution: hub -> frame m -> wave 0: m+1 .. n-1, wave 1: m-1 .. 1, then frame 0's other entries.
        // Lane r owns unknown r: row r of Yo / Yr of the frame's record.
        const int r15 = lane < 15 ? lane : 0;
        auto load_row = [&](int i_, double* row, double& gsv, double& xold, double& scv, double& dgv_) {
            const double* f = sws + (size_t)i_ * SOLVE_WS;
            const double2* f2 = reinterpret_cast<const double2*>(f + r15 * REC_LD);
#pragma unroll
            for (int k = 0; k < 11; ++k) { const double2 v = f2[k]; row[2 * k] = v.x; row[2 * k + 1] = v.y; }
            gsv = f[REC_GS + r15]; xold = xw[(size_t)i_ * 15 + r15]; scv = scl[i_ * 15 + r15]; dgv_ = dgl[i_ * 15 + r15];
        };
        double yav[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) yav[k] = S.ya[k];
        auto emit = [&](int i_, double t, double gsv, double xold, double scv, double dgv_) {
            const bool cst = lane < 15 && var_is_const(a.mode, a.fast_mode, n, i_, lane);
            const double del = (lane < 15 && !cst) ? -t * scv : 0.0;
            const double qv[3] = {rdlane(xold, 3), rdlane(xold, 4), rdlane(xold, 5)};
            double dq[3] = {rdlane(del, 3), rdlane(del, 4), rdlane(del, 5)}, qn[3];
            so3_plus(qv, dq, qn);
            if (lane < 15) {
                double xnew = xold + del;
                if (lane >= 3 && lane < 6) xnew = var_is_const(a.mode, a.fast_mode, n, i_, 3) ? xold : qn[lane - 3];
                xc[(size_t)i_ * 15 + lane] = xnew;
                if (!cst) {
                    sn2 += (xold - xnew) * (xold - xnew);
                    ytg += t * gsv;
                    dsum += dgv_ / radius * t * t;
                }
            }
        };
        // wave 0 walks m, m+1, .., n-1; wave 1 walks m (solution only: wave 0 emits that frame), m-1, .., 1, 0.  Frame k+1's record is in
        // flight while frame k is solved.
        struct BsRow { double row[22], gsv, xold, scv, dgv_; };
        auto fetch = [&](int i_) { BsRow R_; load_row(i_, R_.row, R_.gsv, R_.xold, R_.scv, R_.dgv_); return R_; };
        const int cnt = w == 0 ? n - m : m + 1;
        BsRow curr = fetch(m);
        double yprev = 0.0;
        for (int k = 0; k < cnt; ++k) {
            const int i = w == 0 ? m + k : m - k;
            BsRow nxt = curr;
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < cnt) nxt = fetch(w == 0 ? i + 1 : i - 1);
            __builtin_amdgcn_sched_barrier(0);
            double t = curr.row[21];
#pragma unroll
            for (int q = 0; q < 15; ++q) t -= curr.row[q] * rdlane(yprev, q);      // frame m: no chain neighbour left (Yo = 0)
#pragma unroll
            for (int q = 0; q < 6; ++q) t -= curr.row[15 + q] * yav[q];
            if (w == 1 && i == 0 && lane < 6) t = S.ya[lane];                        // frame 0: pose entries from the hub, the rest from "0c"
            yprev = t;
            if (!(w == 1 && k == 0)) emit(i, t, curr.gsv, curr.xold, curr.scv, curr.dgv_);
            curr = nxt;
        }
        sn2 = wave_sum(sn2); ytg = wave_sum(ytg); dsum = wave_sum(dsum);
    }
    if (lane == 0) { S.red[w][0] = sn2; S.red[w][1] = ytg; S.red[w][2] = dsum; }
    __syncthreads();
    if (w == 0 && lane == 0) {
        const double step_norm = sqrt(S.red[0][0] + S.red[1][0]);
        // model cost change -(s'g_s + s'A s/2) with s = -y and (A + D^2) y = g_s  ==  (y'g_s + y'D^2 y)/2
        const double model_cost_change = 0.5 * ((S.red[0][1] + S.red[1][1]) + (S.red[0][2] + S.red[1][2]));
        const bool valid = all_solved && model_cost_change > 0.0 && isfinite(model_cost_change);
        st.iteration = iteration; st.cur = cur; st.x_cost = x_cost; st.x_norm = x_norm;
        if (valid) {
            st.radius = radius; st.decrease_factor = dec; st.reuse_diagonal = 1;
            st.model_cost_change = model_cost_change; st.cand_step_norm = step_norm; st.have_candidate = 1; st.invalid_steps = 0;
        } else {
            st.invalid_steps += 1;
            st.have_candidate = 0;
            if (st.invalid_steps >= 5) {   // FAILURE hands back the states the solve started from (rare: one lane copies)
                st.done = 1; st.termination = 6; st.iteration = iteration - 1;
                if (st.successful > 0) { for (int e = 0; e < n * 15; ++e) xw[e] = st.x0[e]; st.x_cost = st.initial_cost; }
            }
            st.radius = radius / dec; st.decrease_factor = dec * 2.0; st.reuse_diagonal = 1;
        }
    }
}

__global__ void k_lm_begin(int B, int n, LmState* lm, int max_iters) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    lm_reset(lm[b], max_iters);
}

// everything liw_solve reads back, gathered into ONE record (one device-to-host copy per chunk of iterations):
// [done, marg status, -, - | liw_summary | states n*15 | match_pose n*12 | sqrt_H 36, Delta_H 225, Delta_g 15]
__device__ __forceinline__ void pack_result_body(const PackArgs& a) {   // the whole work-group
    const int t = threadIdx.x, n = a.n;
    // the write-backs of k_lm_finish for this one window first (one launch less per tracking frame), then the record
    if (t < n * 6) {
        const int k = t % 6, i = t / 6;
        if (a.mode == LIW_MODE_INIT) {
            if (a.has_match[i]) { a.match_pose[i * 12 + k] = a.x[k]; a.match_pose[i * 12 + 6 + k] = a.x[(size_t)i * 15 + k]; }
        } else if (a.mode == LIW_MODE_TRACK) {
            if (i == n - 1 && a.has_match[i]) a.match_pose[i * 12 + 6 + k] = a.x[(size_t)i * 15 + k];
        }
    }
    __syncthreads();
    int* hdr = reinterpret_cast<int*>(a.out);
    if (t == 0) { hdr[0] = a.lm[0].done; hdr[1] = a.marg_status ? a.marg_status[0] : 3; hdr[2] = 0; }
    if (t == 1) {
        const LmState& st = a.lm[0];
        liw_summary o;
        o.iterations = st.iteration; o.successful_steps = st.successful; o.termination = st.termination;
        o.initial_cost = st.initial_cost; o.final_cost = st.x_cost;
        *reinterpret_cast<liw_summary*>(a.out + 2) = o;
        a.info[0] = o;
    }
    double* o = a.out + LIW_RESULT_HDR;
    for (int e = t; e < n * 15; e += blockDim.x) o[e] = a.x[e];
    o += n * 15;
    for (int e = t; e < n * 12; e += blockDim.x) o[e] = a.match_pose[e];
    o += n * 12;
    if (a.marg) for (int e = t; e < 276; e += blockDim.x) o[e] = a.marg[e];
    // the sequence word goes LAST, behind a system-scope fence: a host that polls it in the page-locked record (liw_solve, zero-copy)
    // sees the whole record — and what earlier kernels of the stream stored there — without waiting for the stream's completion signal.
    // EVERY thread fences its own record stores before the barrier (the work-group-scope release of the barrier alone does not drain
    // the other waves' stores, which travel through other L2 channels, to system scope)
    __threadfence_system();
    __syncthreads();
    if (t == 0) { __threadfence_system(); __hip_atomic_store(hdr + 3, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}
__global__ void k_pack_result(PackArgs a) { pack_result_body(a); }

// write-backs the reference does after ceres::Solve (solver.cpp:176-190 init, :804-814 tracking) + summaries
__global__ void k_lm_finish(StepArgs a) {   // one thread per (window, frame, pose entry): a thread per window walked n frames serially (40 us at n = 30)
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = a.n;
    if (t >= a.B * n * 6) return;
    const int k = t % 6, i = (t / 6) % n, b = t / (6 * n);
    const double* xw = a.x + (size_t)b * n * 15;
    double* mp = a.match_pose + (size_t)b * n * 12;
    if (a.mode == LIW_MODE_INIT) {
        if (a.has_match[b * n + i]) { mp[i * 12 + k] = xw[k]; mp[i * 12 + 6 + k] = xw[(size_t)i * 15 + k]; }
    } else if (a.mode == LIW_MODE_TRACK) {
        if (i == n - 1 && a.has_match[b * n + i]) mp[i * 12 + 6 + k] = xw[(size_t)i * 15 + k];
    }
    if (i == 0 && k == 0) {
        const LmState& st = a.w.lm[b];
        liw_summary& o = a.w.info[b];
        o.iterations = st.iteration; o.successful_steps = st.successful; o.termination = st.termination;
        o.initial_cost = st.initial_cost; o.final_cost = st.x_cost;
    }
}

// ---------------------------------------------------------------------------------------------------
// dense export of the assembled normal equations (tests, liw_linearize): H [15n x 15n], g, cost
__global__ __launch_bounds__(64) void k_export_dense(ExportArgs a) {
    __shared__ LdsTiles T;
    const int b = blockIdx.x, lane = threadIdx.x & 63, n = a.n, N = 15 * n;
    AsmCtx c;
    c.pif = a.w.pi_frame;
    c.n = n; c.mode = a.mode; c.fast = a.fast_mode; c.b = b; c.buf = a.buf;
    c.PL = a.w.PL[0]; c.PI = a.w.PI[0]; c.PW = a.w.PW[0]; c.PG = a.w.PG[0];   // standalone linearise writes buffer 0
    c.x = a.x + (size_t)b * n * 15;
    c.pJ = a.prior_J + (size_t)b * 225; c.pX = a.prior_X + (size_t)b * 15;
    c.prior_on = a.has_prior[b] && ((a.mode == LIW_MODE_TRACK && !a.fast_mode) || a.mode == LIW_MODE_MARG);
    double* H = a.H + (size_t)b * N * N;
    double* g = a.g + (size_t)b * N;
    for (size_t e = lane; e < (size_t)N * N; e += 64) H[e] = 0.0;
    __syncthreads();
    const double sgn = a.mode == LIW_MODE_MARG ? -1.0 : 1.0;
    for (int i = 0; i < n; ++i) {
        assemble_frame<0>(c, i, Tiles<0>{T.D, T.O, T.R, T.g}, T.tmp);
        for (int e = lane; e < 256; e += 64) {
            const int r = e >> 4, cc = e & 15;
            if (r < 15 && cc < 15) {
                H[(size_t)(i * 15 + r) * N + i * 15 + cc] = T.D[e];
                if (i >= 1) { H[(size_t)((i - 1) * 15 + r) * N + i * 15 + cc] = T.O[e]; H[(size_t)(i * 15 + cc) * N + (i - 1) * 15 + r] = T.O[e]; }
                if (i >= 2 && r < 6) { H[(size_t)r * N + i * 15 + cc] = T.R[e]; H[(size_t)(i * 15 + cc) * N + r] = T.R[e]; }
            }
        }
        if (lane < 15) g[i * 15 + lane] = sgn * T.g[lane];
        __syncthreads();
    }
    const double cst = window_cost(c);
    if (lane == 0) a.cost[b] = cst;
}

// ---------------------------------------------------------------------------------------------------
// marginalisation: chain Schur complement of frames 0..n-2 onto frame n-1 (marginalization_matrix,
// solver.cpp:4-40, on the block tri-diagonal H), eigen square root (solver.cpp:390-402), prior update (:407-441)
// chain Schur complement of one window by ONE wave: Delta_H -> T.D (ld 16), Delta_g(+J^T R convention) -> T.g, outputs, status; false = a pivot failed
template <class TILES>
__device__ __forceinline__ bool marg_chain(const MargArgs& a, const int b, TILES& T) {
    const int lane = threadIdx.x & 63, n = a.n;
    AsmCtx c;
    c.pif = a.w.pi_frame;
    const int sel = a.use_cur ? __builtin_amdgcn_readfirstlane(a.w.lm[b].cur) : 0;
    c.n = n; c.mode = LIW_MODE_MARG; c.fast = 0; c.b = b; c.buf = sel;
    c.PL = sel ? a.w.PL[1] : a.w.PL[0]; c.PI = sel ? a.w.PI[1] : a.w.PI[0]; c.PW = sel ? a.w.PW[1] : a.w.PW[0]; c.PG = sel ? a.w.PG[1] : a.w.PG[0];
    c.x = a.x + (size_t)b * n * 15;
    c.pJ = a.prior_J + (size_t)b * 225; c.pX = a.prior_X + (size_t)b * 15;
    c.prior_on = a.has_prior[b] != 0;
    STAMPM(5000);
    for (int e = lane; e < 256; e += 64) { T.CD[e] = 0.0; T.W[e] = 0.0; }
    if (lane < 16) T.Cg[lane] = 0.0;
    lds_sync();
    bool ok = true;
    // Chain elimination 0 .. n-2 with the same register-resident fused Cholesky / forward substitution as k_lm_step:
    // lane j < 15 owns column j of D_i (+ carried Schur term), lanes 16..30 the columns of H[i, i+1], lane 40 g_i.
    // Two tile sets ping-pong so that every frame is assembled once: set `cur` holds D_i, g_i; assembling frame i+1
    // into `nxt` yields H[i, i+1] (its O tile) together with D_{i+1}, g_{i+1}.
    double* Dc = T.D; double* gc = T.g;          // cur
    double* Dn = T.Wa; double* gn = T.y0;        // nxt   (T.O receives H[i, i+1]; T.R is the unused arrow tile)
    assemble_frame<0>(c, 0, Tiles<0>{Dc, T.O, T.R, gc}, T.tmp);
    AsmRegs areg = asm_issue(c, n > 1 ? 1 : 0, nullptr, nullptr);
    for (int i = 0; i + 1 < n; ++i) {
        asm_commit<0>(c, i + 1, areg, Tiles<0>{Dn, T.O, T.R, gn}, T.tmp);
        // software pipeline: frame i+2's partial sums are in flight while frame i is eliminated
        __builtin_amdgcn_sched_barrier(0);
        if (i + 2 < n) areg = asm_issue(c, i + 2, nullptr, nullptr);
        __builtin_amdgcn_sched_barrier(0);
        // this lane's column as the sum of two strided LDS reads (matrix lanes: D + carried term; coupling lanes: O + the zero word
        // T.Cg[15] at stride 0; gradient lane: g + carried gradient, stride 1) — written as a chain of lane tests the loop was five reads
        // and a nest of selects per row.  (Until round 6 the coupling lanes added words of the arrow tile R, which holds the laser H_ab
        // slots of the partial buffer: zero after a one-pose linearisation, NOT after an init-topology one — ADVICE r5.  T.Cg[15] is
        // cleared at entry and never written.)
        double col[15];
        {
            const bool ml_ = lane < 15, ol_ = lane >= 16 && lane < 31, gl_ = lane == 40;
            const double* pA = ml_ ? Dc + lane : (ol_ ? T.O + (lane - 16) : (gl_ ? gc : T.R));
            const double* pB = ml_ ? T.CD + lane : (gl_ ? T.Cg : (ol_ ? T.Cg + 15 : T.R));
            const int st = gl_ ? 1 : 16, stB = ol_ ? 0 : st;
#pragma unroll
            for (int r = 0; r < 15; ++r) col[r] = pA[r * st] + pB[r * stB];
        }
        if (!fused_chol_solve(col)) { ok = false; break; }
        if ((lane >= 16 && lane < 31) || lane == 40) {
            double* const wc = T.W + (lane == 40 ? 15 : lane - 16);
#pragma unroll
            for (int r = 0; r < 15; ++r) wc[r * 16] = col[r];
        }
        lds_sync();
        const d4 p1 = xty16(T.W, T.W);           // [W | z]^T [W | z]: Schur terms for D_{i+1} and g_{i+1}
        lds_sync();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (lane >> 4) + 4 * r, colx = lane & 15;
            if (row < 15 && colx < 15) T.CD[row * 16 + colx] = -p1[r];
            if (row < 15 && colx == 15) T.Cg[row] = -p1[r];
        }
        lds_sync();
        double* t1 = Dc; Dc = Dn; Dn = t1;
        double* t2 = gc; gc = gn; gn = t2;
    }
    // Delta_H = D_{n-1} + carried term into T.D, Delta_g(+J^T R convention) into T.g
    {
        double dv[4], gv = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q, r = e >> 4, cc = e & 15;
            dv[q] = (r < 15 && cc < 15) ? Dc[e] + T.CD[e] : 0.0;
        }
        if (lane < 15) gv = gc[lane] + T.Cg[lane];
        lds_sync();
#pragma unroll
        for (int q = 0; q < 4; ++q) T.D[lane + 64 * q] = dv[q];
        if (lane < 16) T.g[lane] = lane < 15 ? gv : 0.0;
        lds_sync();
    }
    if (a.status && lane == 0) a.status[b] = ok ? 0 : 1;
    if (!ok) return false;
    // Delta_H = T.D (15x15), Delta_g = -T.g  (g = -J^T R in the reference)
    if (a.Delta_H) for (int e = lane; e < 225; e += 64) a.Delta_H[(size_t)b * 225 + e] = T.D[(e / 15) * 16 + e % 15];
    if (a.Delta_g && lane < 15) a.Delta_g[(size_t)b * 15 + lane] = -T.g[lane];
    return true;
}

// The plane rotation that annihilates a_pq, from d = a_qq - a_pp and h = 2 a_pq (h != 0): t = sgn(d) h / (|d| + r), r = sqrt(d^2 + h^2),
// cs = 1 / sqrt(1 + t^2), sn = t cs.  With u = |d| + r:  1 + t^2 = (u^2 + h^2) / u^2 = 2 r u / u^2, so  cs = u / sqrt(2 r u),
// sn = sgn(d) h / sqrt(2 r u): two reciprocal square roots on the dependent chain (until round 4: rsqrt, reciprocal, rsqrt — a round of
// the four-wave Jacobi is ~660 cycles of mostly this chain).  cs^2 + sn^2 = (u^2 + h^2) / (2 r u) = 1 up to the rounding of the products.
__device__ __forceinline__ void jacobi_rotation(double d, double h, double& cs, double& sn) {
    const double qq = d * d + h * h;
    const double r = qq * fast_rsqrt(qq);
    const double u = fabs(d) + r;
    const double winv = fast_rsqrt(2.0 * r * u);
    cs = u * winv;
    sn = (d >= 0.0 ? h : -h) * winv;
}
// cyclic Jacobi by ONE wave (large batches: a wave per window): T.D -> eigenvalues on the diagonal of Am, eigenvectors in the columns of V
template <class TILES>
__device__ __forceinline__ void jacobi15_wave(TILES& T, double* V, double* Am, double* rot) {
    const int lane = threadIdx.x & 63;
    // ---- symmetric eigen-decomposition by cyclic Jacobi (15x15), A -> Am, eigenvectors -> V (columns).
    // Jacobi, not tridiagonalisation + QL: Delta_H is graded over 1e11 (pose rows) ... 1e2 (bias rows), and only Jacobi keeps the small
    // blocks accurate RELATIVE to their own scale (a QL variant was 15 % faster, normwise-accurate like Eigen's solver in the reference,
    // and moved the tracking solve on the resulting prior by 1e-6 at cond(H_mm) = 1e7 — tests/soak/soak_batch.py found it).
    for (int e = lane; e < 256; e += 64) {
        const int r = e >> 4, cc = e & 15;
        Am[e] = (r < 15 && cc < 15) ? 0.5 * (T.D[r * 16 + cc] + T.D[cc * 16 + r]) : 0.0;
        V[e] = r == cc ? 1.0 : 0.0;
    }
    lds_sync();
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dgn = 0.0;
        for (int e = lane; e < 225; e += 64) {
            const int r = e / 15, cc = e % 15;
            const double v = Am[r * 16 + cc];
            if (cc > r) off += v * v;
            if (cc == r) dgn += v * v;
        }
        off = wave_sum(off); dgn = wave_sum(dgn);
        if (off <= 1e-34 * dgn || off == 0.0) break;   // sums of squares: off-diagonal below 1e-17 of the diagonal
        // parallel (round-robin) ordering: 15 rounds of 7 disjoint index pairs; the 7 plane rotations of a round are
        // computed from the same A and applied together (columns of A and V, then rows of A), 105 independent element
        // pairs per phase spread over the wave.  The pair (p, q) of slot i in round rnd is recomputed from the indices by every lane
        // (a few integer operations instead of an LDS round trip), only cos / sin travel through LDS.
        for (int rnd = 0; rnd < 15; ++rnd) {
            auto pair_of = [&](int i, int& p, int& q) {
                const int k = i + 1;
                p = rnd + k; p = p >= 15 ? p - 15 : p;
                q = rnd + 15 - k; q = q >= 15 ? q - 15 : q;
                if (p > q) { const int t_ = p; p = q; q = t_; }
            };
            if (lane < 7) {
                int p, q;
                pair_of(lane, p, q);
                const double apq = Am[p * 16 + q];
                double cs = 1.0, sn = 0.0;
                if (apq != 0.0) {
                    // t = sgn(tau) / (|tau| + sqrt(1 + tau^2)) with tau = (aqq - app) / (2 apq), written without the division by apq;
                    // square root, reciprocal and reciprocal square root from the hardware estimates + Newton steps (the rotation only
                    // has to be orthogonal to working precision: cs^2 + sn^2 = 1 holds by construction)
                    const double d = Am[q * 16 + q] - Am[p * 16 + p], h = 2.0 * apq;
                    jacobi_rotation(d, h, cs, sn);
                }
                rot[lane * 2] = cs; rot[lane * 2 + 1] = sn;
            }
            lds_sync();
            for (int it = lane; it < 105; it += 64) {   // columns p,q of A and V: (row k, pair i)
                const int k = it / 7, i = it - 7 * k;
                int p, q;
                pair_of(i, p, q);
                const double cs = rot[i * 2], sn = rot[i * 2 + 1];
                const double akp = Am[k * 16 + p], akq = Am[k * 16 + q];
                const double vkp = V[k * 16 + p], vkq = V[k * 16 + q];
                Am[k * 16 + p] = cs * akp - sn * akq;
                Am[k * 16 + q] = sn * akp + cs * akq;
                V[k * 16 + p] = cs * vkp - sn * vkq;
                V[k * 16 + q] = sn * vkp + cs * vkq;
            }
            lds_sync();
            for (int it = lane; it < 105; it += 64) {   // rows p,q of A: (pair i, column k)
                const int k = it / 7, i = it - 7 * k;
                int p, q;
                pair_of(i, p, q);
                const double cs = rot[i * 2], sn = rot[i * 2 + 1];
                const double apk = Am[p * 16 + k], aqk = Am[q * 16 + k];
                Am[p * 16 + k] = cs * apk - sn * aqk;
                Am[q * 16 + k] = sn * apk + cs * aqk;
            }
            lds_sync();
        }
    }
}

// eigen square root and prior write-back (solver.cpp:390-441) by ONE wave: Am (eigenvalues on the diagonal), V (eigenvectors in columns)
template <class TILES>
__device__ __forceinline__ void marg_tail(const MargArgs& a, const int b, TILES& T, const double* V, const double* Am) {
    const int lane = threadIdx.x & 63, n = a.n;
    double* oX = a.out_X ? a.out_X : a.prior_X; double* oJ = a.out_J ? a.out_J : a.prior_J; double* oR = a.out_R ? a.out_R : a.prior_R;
    int* oHas = a.out_has ? a.out_has : a.has_prior;
    const double* xw = a.x + (size_t)b * n * 15;
    STAMPM(5002);
    // sort ascending (rank by counting; ties by index), sign convention: largest |component| positive
    if (lane < 15) {
        const double w = Am[lane * 16 + lane];
        int rank = 0;
        for (int k = 0; k < 15; ++k) { const double wk = Am[k * 16 + k]; if (wk < w || (wk == w && k < lane)) ++rank; }
        int m = 0;
        for (int k = 1; k < 15; ++k) if (fabs(V[k * 16 + lane]) > fabs(V[m * 16 + lane])) m = k;
        const double sg = V[m * 16 + lane] < 0.0 ? -1.0 : 1.0;
        const double eps = 1e-8;
        const double S = w > eps ? w : 0.0, Sinv = w > eps ? 1.0 / w : 0.0;
        const double ssq = sqrt(S), sisq = sqrt(Sinv);
        double dotg = 0.0;
        for (int k = 0; k < 15; ++k) dotg += sg * V[k * 16 + lane] * (-T.g[k]);   // V^T Delta_g
        // linearized_jacobians row `rank` = sqrt(S) v^T ; linearized_residuals[rank] = -(S^-1/2 v^T Delta_g)
        for (int k = 0; k < 15; ++k) oJ[(size_t)b * 225 + rank * 15 + k] = ssq * sg * V[k * 16 + lane];
        oR[(size_t)b * 15 + rank] = -(sisq * dotg);
    }
    wave_mem_sync();
    __threadfence_block();
    if (lane < 15) oX[(size_t)b * 15 + lane] = xw[(size_t)(n - 1) * 15 + lane];
    if (a.sqrt_H) for (int e = lane; e < 36; e += 64) a.sqrt_H[(size_t)b * 36 + e] = oJ[(size_t)b * 225 + (e / 6) * 15 + e % 6];
    if (lane == 0) oHas[b] = 1;
    STAMPM(5003);
}
template <class TILES>
__device__ __forceinline__ void marg_schur_body(const MargArgs& a, const int b, TILES& T, double* V, double* Am, double* rot) {
    const int lane = threadIdx.x & 63;
    if (a.gate && !a.gate[b].done) { if (a.status && lane == 0) a.status[b] = 2; return; }
    if (!marg_chain(a, b, T)) return;
    STAMPM(5001);
    jacobi15_wave(T, V, Am, rot);
    marg_tail(a, b, T, V, Am);
}
// Cyclic Jacobi by a work-group of FOUR waves (a few windows: the marginalisation behind a tracking solve is on the caller's critical
// path): one matrix element per thread.  Same rotations in the same round-robin order as jacobi15_wave; but the parameters of the (at
// most two) rotations that touch element (r, c) are recomputed by the element's own thread from the current matrix, the two-sided update
//   A'[r][c] = gr (A[r][c] gc + A[r][c'] sc) + sr (A[r'][c] gc + A[r'][c'] sc),   V'[r][c] = V[r][c] gc + V[r][c'] sc
// reads the current buffer and writes the other one, so a round is ONE work-group barrier instead of three LDS hand-offs with 7 of 64
// lanes computing parameters.  A2 / V2: two 16x16 buffers each; returns the index (0 / 1) of the buffer that holds the result.
// value of lane 0 of this lane's 16-lane row (DP-ALU DPP: row_newbcast is the one control 64-bit moves take on gfx90a+).  volatile and in
// uniform control flow only: a DPP read of a disabled lane is not a read; s_nop: a VGPR written by a VALU instruction needs two wait states
// before a DPP instruction reads it, and the hazard recogniser does not look inside inline asm.
__device__ __forceinline__ double row_bc0(double v) {
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ int jacobi15_block(const double* D, double* A2, double* V2, double* red) {
    // thread t owns element (r, c) = (t >> 4, (t + r) & 15), stored at t_ = 16 r + c: the columns of a row are rotated so that the row's DIAGONAL
    // element sits in lane 0 of its 16-lane DPP row — what that thread computes reaches the row as a row_newbcast:0 operand (below)
    const int t = threadIdx.x, r = t >> 4, c = (t + r) & 15, t_ = r * 16 + c, lane = t & 63, wave = t >> 6;
    const bool in = r < 15 && c < 15;
    A2[t_] = in ? 0.5 * (D[r * 16 + c] + D[c * 16 + r]) : 0.0;
    V2[t_] = r == c ? 1.0 : 0.0;
    __syncthreads();
    int cur = 0;
    // coefficients of index i in round rnd: new_i = g * x_i + s * x_partner.  partner = (2 rnd - i) mod 15 (the pairs of a round are the
    // index pairs that sum to 2 rnd), none for i = rnd and for the padding index 15.
    // A round is ONE LDS round trip (round 5): the partner indices are integer arithmetic on (thread, round), so the three matrix entries
    // behind each of the thread's two rotations and the six entries of its two-sided update are all loaded up front, unconditionally
    // (an index without a partner pairs with itself: finite garbage, discarded by the selects below); until then the zero test of a_pq
    // and the "no partner" exits were branches in front of the diagonal loads, and the update's loads sat behind both rotations — three
    // dependent LDS round trips and two divergent regions in a round of ~1 190 cycles (tools/clk_probe_track.py: 4 sweeps = 60 rounds in
    // 71.6 k cycles).  Same rotations, same order, same arithmetic: bit-identical results.
    auto partner_of = [](int i, int rnd) {
        int pr = 2 * rnd - i; pr += pr < 0 ? 15 : 0; pr -= pr >= 15 ? 15 : 0;
        return i >= 15 ? i : pr;
    };
    // offsets (in doubles, inside the current A / V image) of what round rnd reads, and the round's three predicates
    struct RoundIdx { int pq, qq, pp, x01, x10, x11; bool con, c_is_p, r_is_p; };
    auto round_idx = [&](int rnd) {
        const int rp = partner_of(r, rnd), cp = partner_of(c, rnd);
        const int cP = c < cp ? c : cp, cQ = c < cp ? cp : c;
        RoundIdx I;
        I.pq = cP * 16 + cQ; I.qq = cQ * 17; I.pp = cP * 17;
        I.x01 = r * 16 + cp; I.x10 = rp * 16 + c; I.x11 = rp * 16 + cp;
        I.con = cp != c; I.c_is_p = c <= cp; I.r_is_p = r <= rp;   // (an index without a partner pairs with itself: "is p")
        return I;
    };
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double* A = A2 + 256 * cur;
        const double v = in ? A[t_] : 0.0;
        double off = c > r ? v * v : 0.0, dgn = c == r ? v * v : 0.0;
        off = wave_sum(off); dgn = wave_sum(dgn);
        if (lane == 0) { red[wave] = off; red[4 + wave] = dgn; }
        __syncthreads();
        off = (red[0] + red[1]) + (red[2] + red[3]); dgn = (red[4] + red[5]) + (red[6] + red[7]);
        __syncthreads();
        if (off <= 1e-34 * dgn || off == 0.0) break;   // (uniform) sums of squares: off-diagonal below 1e-17 of the diagonal
        RoundIdx I = round_idx(0);
        for (int rnd = 0; rnd < 15; ++rnd) {
            const double* Ac = A2 + 256 * cur; const double* Vc = V2 + 256 * cur;
            double* An = A2 + 256 * (1 - cur); double* Vn = V2 + 256 * (1 - cur);
            // every load of the round
            const double c_pq = Ac[I.pq], c_qq = Ac[I.qq], c_pp = Ac[I.pp];
            const double x00 = Ac[t_], x01 = Ac[I.x01], x10 = Ac[I.x10], x11 = Ac[I.x11];
            const double v0 = Vc[t_], v1 = Vc[I.x01];
            JSTAMP(5020);
            // the NEXT round's indices in the shadow of the loads: a wave is alone on its SIMD, and as the first ~35 instructions of a round
            // this integer arithmetic sat in front of its loads (~150 of a round's ~890 cycles)
            __builtin_amdgcn_sched_barrier(0);
            const RoundIdx In = round_idx(rnd < 14 ? rnd + 1 : 0);
            __builtin_amdgcn_sched_barrier(0);
            // A thread computes the rotation of its COLUMN's pair; the rotation of its row's pair is the one the row's diagonal thread
            // (column index = r: the same pair, the same three entries, the same instructions) computes for its column — lane 0 of the
            // thread's DPP row (the rotated column order above), i.e. a row_newbcast:0 move instead of a second evaluation by all 16 threads
            // of the row (until late round 5) or a ds_bpermute (~200 of a round's ~890 cycles, stamps of tools/clk_probe_track.py):
            // a wave is alone on its SIMD here, so a round costs what it ISSUES.
            JSTAMP(5021);
            double ccs, csn;
            jacobi_rotation(c_qq - c_pp, 2.0 * c_pq, ccs, csn);
            JSTAMPV(5022, ccs + csn);
            const bool con = I.con && c_pq != 0.0;
            const double gc = con ? ccs : 1.0, scm = con ? csn : 0.0;
            const double sc = I.c_is_p ? -scm : scm;
            const double gr = row_bc0(gc), srm = row_bc0(scm);
            const double sr = I.r_is_p ? -srm : srm;
            JSTAMPV(5023, sr + gr);
            const double b0 = gc * x00 + sc * x01;      // (A G)[r][c]
            const double b1 = gc * x10 + sc * x11;      // (A G)[r'][c]
            An[t_] = gr * b0 + sr * b1;
            Vn[t_] = gc * v0 + sc * v1;
            JSTAMP(5024);
            __syncthreads();
            JSTAMP(5025);
            cur = 1 - cur;
            I = In;
        }
#ifdef LIW_CLK
        if (t == 0 && blockIdx.x == 0) g_clk[5010] = sweep + 1;   // rotation sweeps executed (tools/clk_probe_track.py)
#endif
    }
    return cur;
}

// small batches: four waves per window — wave 0 the chain and the eigen square root, all four the Jacobi sweeps in between
__global__ __launch_bounds__(256, 1) void k_marg_schur4(MargArgs a) {
    __shared__ LdsTiles T;
    __shared__ double A2[512], V2[512], red[8];
    __shared__ int okf;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (a.gate && !a.gate[b].done) { if (a.status && threadIdx.x == 0) a.status[b] = 2; return; }   // (uniform)
    if (wave == 0) { const bool ok = marg_chain(a, b, T); if (lane == 0) okf = ok ? 1 : 0; }
    STAMPM(5001);
    __syncthreads();
    if (!okf) return;
    const int cur = jacobi15_block(T.D, A2, V2, red);
    if (wave == 0) marg_tail(a, b, T, V2 + 256 * cur, A2 + 256 * cur);
}

// One wave per window (large batches).  The kernel is a chain of dependent 15x15 steps on ONE wave — latency-bound — so what it needs is
// waves per SIMD: 151 registers allow three, the 20 kB of LdsTiles + eigen buffers allowed two (eight waves per CU).  LdsMarg keeps only
// the tiles the marginalisation topology touches (no arrow / carried-arrow tiles, no LM vectors) and the eigen phase re-uses tiles that are
// dead after the chain (V in the never-read arrow tile R, the rotated matrix in O, the rotation parameters in Cg): 12.9 kB, twelve waves per CU.
struct LdsMarg {
    double D[256], O[256], R[256], W[256], Wa[256], CD[256];
    double g[16], Cg[16], y0[16], tmp[ASM_TMP];
};
#ifndef LIW_MARG_OCC
#define LIW_MARG_OCC 3
#endif
__global__ __launch_bounds__(64, LIW_MARG_OCC) void k_marg_schur(MargArgs a) {
    __shared__ LdsMarg T;
    static_assert(sizeof(LdsMarg) * 12 <= 160 * 1024, "twelve waves per CU");
    marg_schur_body(a, (int)blockIdx.x, T, T.R, T.O, T.Cg);
}


// ---------------------------------------------------------------------------------------------------
// Large batches, since the end of round 5: the chain and the eigen square root are TWO kernels.  k_marg_schur_chain is marg_chain by one wave
// per window as before and leaves Delta_H | Delta_g | ok in the window's (dead) factorisation scratch; k_marg_schur_eigq takes FOUR windows
// per wave, 16 lanes per window: lane r holds row r of the rotated matrix and row r of the eigenvector matrix in REGISTERS.  The one-wave
// Jacobi spent a round on three LDS hand-offs with 7 of 64 lanes computing the rotations and 105 of 128 lane slots updating elements
// (~270 instructions per window and round); here every lane computes the rotation of the pair its own row is in (both lanes of a pair
// from the same three matrix words, so they agree bit for bit), takes the partner's row from LDS (eight 16-byte reads, issued with the
// three words of the rotation), combines the two rows (the row half of G^T A G), receives the seven (cos, sin) of the round as
// row_newbcast DPP operands — the pairs of a round are compile-time constants, the 15 rounds are unrolled — and rotates its 7 column
// pairs of A and V in registers: ~190 instructions per round for four windows.  Same rotations, same round-robin order, same stop test
// (per window: a window that has met it keeps identity rotations while its neighbours in the wave finish) as jacobi15_wave.
constexpr int JQ_LD = 18;                        // row stride of a window's matrix image in LDS (doubles): 144 B — 16-byte row reads of 16 lanes touch 16 different bank groups
constexpr int JQ_WIN = 16 * JQ_LD;
constexpr int MARG_SCR_G = 225, MARG_SCR_OK = 240;   // hand-over record (in solve_ws of the window): Delta_H (row-major 15) | Delta_g (reference sign) | 1.0 / 0.0
static_assert(SOLVE_WS > MARG_SCR_OK, "the hand-over record fits the scratch of a one-frame window");

template <int N>
__device__ __forceinline__ double row_bc(double v) {   // value of lane N of this lane's 16-lane row (see row_bc0)
    double r;
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(N));
    return r;
}
__host__ __device__ constexpr int jq_pair(int rnd, int i, bool hi) {   // slot i of round rnd, as pair_of in jacobi15_wave
    const int k = i + 1;
    int p = rnd + k; p = p >= 15 ? p - 15 : p;
    int q = rnd + 15 - k; q = q >= 15 ? q - 15 : q;
    return hi ? (p > q ? p : q) : (p > q ? q : p);
}
// compile-time check of the schedule the unrolled rounds rely on: every round pairs 14 of the 15 indices into 7 disjoint (p < q) pairs and
// leaves index `rnd` idle, the partner of row r is (2 rnd - r) mod 15 (jq_round), and the 15 rounds meet each of the 105 pairs exactly once
constexpr bool jq_schedule_ok() {
    bool seen[15][15] = {};
    for (int rnd = 0; rnd < 15; ++rnd) {
        bool used[15] = {};
        for (int i = 0; i < 7; ++i) {
            const int p = jq_pair(rnd, i, false), q = jq_pair(rnd, i, true);
            if (!(p < q) || used[p] || used[q] || seen[p][q]) return false;
            int m = 2 * rnd - p;
            m = m < 0 ? m + 15 : m;
            m = m >= 15 ? m - 15 : m;
            int m2 = 2 * rnd - q;
            m2 = m2 < 0 ? m2 + 15 : m2;
            m2 = m2 >= 15 ? m2 - 15 : m2;
            if (m != q || m2 != p) return false;
            used[p] = used[q] = true;
            seen[p][q] = true;
        }
        if (used[rnd]) return false;
    }
    return true;
}
static_assert(jq_schedule_ok(), "round-robin schedule of k_marg_schur_eigq");
template <int RND, int I>
__device__ __forceinline__ void jq_cols(double (&t)[16], double (&V)[15], const double cs, const double sn) {
    constexpr int p = jq_pair(RND, I, false), q = jq_pair(RND, I, true);
    const double ci = row_bc<p>(cs), si = row_bc<p>(sn);
    const double tp = t[p], tq = t[q], vp = V[p], vq = V[q];
    t[p] = ci * tp - si * tq;
    t[q] = si * tp + ci * tq;
    V[p] = ci * vp - si * vq;
    V[q] = si * vp + ci * vq;
}
template <int RND>
__device__ __forceinline__ void jq_round(double (&A)[16], double (&V)[15], double* LAg, const int r_, const bool active) {
    typedef double __attribute__((ext_vector_type(2))) dbl2;
    // (the round's partner / address arithmetic is a dozen integer instructions of the lane index; opaque to the optimiser, or the
    // invariant parts of all 15 unrolled rounds are hoisted out of the sweep loop and live across it: 400 spilled registers)
    int r = r_;
    asm volatile("" : "+v"(r));
    // partner row: p + q = 2 RND (mod 15); row RND (and the pad lane 15) has none this round
    int m = 2 * RND - r;
    m = m < 0 ? m + 15 : m;
    m = m >= 15 ? m - 15 : m;
    m = r == 15 ? 15 : m;
    const double* rowm = LAg + m * JQ_LD;
    double T[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const dbl2 v = *reinterpret_cast<const dbl2*>(rowm + 2 * k);
        T[2 * k] = v.x; T[2 * k + 1] = v.y;
    }
    const int lo = r < m ? r : m, hi = r < m ? m : r;
    const double apq = LAg[lo * JQ_LD + hi], all = LAg[r * JQ_LD + r], amm = rowm[m];
    const bool isp = r < m;
    double cs = 1.0, sn = 0.0;
    if (active && m != r && apq != 0.0) jacobi_rotation(isp ? amm - all : all - amm, 2.0 * apq, cs, sn);
    // rows: row_p' = cs row_p - sn row_q, row_q' = sn row_p + cs row_q
    const double sr = isp ? -sn : sn;
    double t[16];
#pragma unroll
    for (int k = 0; k < 15; ++k) t[k] = cs * A[k] + sr * T[k];
    t[15] = 0.0;
    // columns of A and V: the seven rotations of the round from the lanes that hold their smaller index
    jq_cols<RND, 0>(t, V, cs, sn); jq_cols<RND, 1>(t, V, cs, sn); jq_cols<RND, 2>(t, V, cs, sn); jq_cols<RND, 3>(t, V, cs, sn);
    jq_cols<RND, 4>(t, V, cs, sn); jq_cols<RND, 5>(t, V, cs, sn); jq_cols<RND, 6>(t, V, cs, sn);
    // (pin the eigenvector row to this round: its rotations depend on nothing in LDS, and left alone the compiler sinks those of all 15
    // rounds to the end of the sweep, keeping 210 broadcast values alive in scratch)
#pragma unroll
    for (int k = 0; k < 15; ++k) asm volatile("" : "+v"(V[k]));
    lds_sync();                                   // (every lane has read its partner's row)
    double* rowr = LAg + r * JQ_LD;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        A[2 * k] = t[2 * k]; A[2 * k + 1] = t[2 * k + 1];
        dbl2 v; v.x = t[2 * k]; v.y = t[2 * k + 1];
        *reinterpret_cast<dbl2*>(rowr + 2 * k) = v;
    }
    lds_sync();
}

__global__ __launch_bounds__(64, LIW_MARG_OCC) void k_marg_schur_chain(MargArgs a) {
    __shared__ LdsMarg T;
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    double* S = a.w.solve_ws + (size_t)b * a.n * SOLVE_WS;
    if (a.gate && !a.gate[b].done) {
        if (lane == 0) { if (a.status) a.status[b] = 2; S[MARG_SCR_OK] = 0.0; }
        return;
    }
    const bool ok = marg_chain(a, b, T);
    if (ok) {
        for (int e = lane; e < 225; e += 64) S[e] = T.D[(e / 15) * 16 + e % 15];
        if (lane < 15) S[MARG_SCR_G + lane] = -T.g[lane];
    }
    if (lane == 0) S[MARG_SCR_OK] = ok ? 1.0 : 0.0;
}

__global__ __launch_bounds__(64, 4) void k_marg_schur_eigq(MargArgs a) {
    __shared__ __attribute__((aligned(16))) double LA[4 * JQ_WIN];
    __shared__ double Wd[64];
    const int lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15, n = a.n;
    const int b = (int)blockIdx.x * 4 + g;
    const double* S = a.w.solve_ws + (size_t)(b < a.B ? b : 0) * n * SOLVE_WS;
    const bool ok = b < a.B && S[MARG_SCR_OK] != 0.0;
    double* LAg = LA + g * JQ_WIN;
    double A[16], V[15];
    // A = (Delta_H + Delta_H^T) / 2, V = I
#pragma unroll
    for (int c = 0; c < 15; ++c) {
        const bool in = ok && r < 15;
        const double u = in ? S[r * 15 + c] : 0.0, l = in ? S[c * 15 + r] : 0.0;
        A[c] = 0.5 * (u + l);
        V[c] = c == r ? 1.0 : 0.0;
    }
    A[15] = 0.0;
    {
        typedef double __attribute__((ext_vector_type(2))) dbl2;
        double* rowr = LAg + r * JQ_LD;
#pragma unroll
        for (int k = 0; k < 8; ++k) { dbl2 v; v.x = A[2 * k]; v.y = A[2 * k + 1]; *reinterpret_cast<dbl2*>(rowr + 2 * k) = v; }
    }
    lds_sync();
    bool active = ok;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dgn = 0.0;
#pragma unroll
        for (int c = 0; c < 15; ++c) {
            const double sq = A[c] * A[c];
            off += c > r ? sq : 0.0;
            dgn += c == r ? sq : 0.0;
        }
        off = row_sum(off); dgn = row_sum(dgn);
        if (off <= 1e-34 * dgn || off == 0.0) active = false;   // sums of squares: off-diagonal below 1e-17 of the diagonal (per window)
        if (!__any(active)) break;
        jq_round<0>(A, V, LAg, r, active); jq_round<1>(A, V, LAg, r, active); jq_round<2>(A, V, LAg, r, active);
        jq_round<3>(A, V, LAg, r, active); jq_round<4>(A, V, LAg, r, active); jq_round<5>(A, V, LAg, r, active);
        jq_round<6>(A, V, LAg, r, active); jq_round<7>(A, V, LAg, r, active); jq_round<8>(A, V, LAg, r, active);
        jq_round<9>(A, V, LAg, r, active); jq_round<10>(A, V, LAg, r, active); jq_round<11>(A, V, LAg, r, active);
        jq_round<12>(A, V, LAg, r, active); jq_round<13>(A, V, LAg, r, active); jq_round<14>(A, V, LAg, r, active);
    }
    // eigenvalue r and the eigenvector matrix (row r from this lane) through LDS: lane k of a window takes eigenvector k = column k
    Wd[lane] = LAg[r * JQ_LD + r];
    lds_sync();
#pragma unroll
    for (int c = 0; c < 15; ++c) LAg[r * JQ_LD + c] = V[c];
    lds_sync();
    // eigen square root and prior write-back (solver.cpp:390-441), as marg_tail
    double* oX = a.out_X ? a.out_X : a.prior_X; double* oJ = a.out_J ? a.out_J : a.prior_J; double* oR = a.out_R ? a.out_R : a.prior_R;
    int* oHas = a.out_has ? a.out_has : a.has_prior;
    if (ok && r < 15) {
        const double* Wg = Wd + 16 * g;
        const double w = Wg[r];
        int rank = 0;
        for (int k = 0; k < 15; ++k) { const double wk = Wg[k]; if (wk < w || (wk == w && k < r)) ++rank; }
        int mx = 0;
        for (int k = 1; k < 15; ++k) if (fabs(LAg[k * JQ_LD + r]) > fabs(LAg[mx * JQ_LD + r])) mx = k;
        const double sg = LAg[mx * JQ_LD + r] < 0.0 ? -1.0 : 1.0;
        const double eps = 1e-8;
        const double Sv = w > eps ? w : 0.0, Sinv = w > eps ? 1.0 / w : 0.0;
        const double ssq = sqrt(Sv), sisq = sqrt(Sinv);
        double dotg = 0.0;
        for (int k = 0; k < 15; ++k) dotg += sg * LAg[k * JQ_LD + r] * S[MARG_SCR_G + k];   // V^T Delta_g
        for (int k = 0; k < 15; ++k) oJ[(size_t)b * 225 + rank * 15 + k] = ssq * sg * LAg[k * JQ_LD + r];
        oR[(size_t)b * 15 + rank] = -(sisq * dotg);
    }
    wave_mem_sync();
    __threadfence_block();
    if (ok) {
        const double* xw = a.x + (size_t)b * n * 15;
        if (r < 15) oX[(size_t)b * 15 + r] = xw[(size_t)(n - 1) * 15 + r];
        if (a.sqrt_H) for (int e = r; e < 36; e += 16) a.sqrt_H[(size_t)b * 36 + e] = oJ[(size_t)b * 225 + (e / 6) * 15 + e % 6];
        if (r == 0) oHas[b] = 1;
    }
}

#ifdef LIW_CLK
extern "C" void liw_debug_clk(long long* out, int nn) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clk), sizeof(long long) * nn); }
extern "C" void liw_debug_span(long long* out, int nn) { hipMemcpyFromSymbol(out, HIP_SYMBOL(g_span), sizeof(long long) * nn); }
#endif
void launch_pack_result(const PackArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_pack_result, dim3(1), dim3(512), 0, s, a); }
void launch_lm_begin(int B, int n, LmState* lm, int max_iters, hipStream_t s) {
    hipLaunchKernelGGL(k_lm_begin, dim3((B + 63) / 64), dim3(64), 0, s, B, n, lm, max_iters);
}
void launch_lm_step(const StepArgs& a, hipStream_t s) {
    // LIW_STEP_VARIANT (read per launch) 0 / 1 / 2 / 3 / 4: force the one-wave latency / one-wave throughput / four-wave / quad / dense two-frame variant (profiling, tests)
    const char* env = getenv("LIW_STEP_VARIANT");
    const bool tp = env ? env[0] == '1' : a.B > 2048;
    // two waves per window while that does not take CUs away from other windows, and the chain is long enough to be worth cutting
    const bool tw = (env ? env[0] == '2' : a.B <= 512) && a.n >= 6;   // (tools/step_variant_sweep.py: 97 k vs 67 k solves/s at 256 windows, 120 k vs 116 k at 512)
    // four windows per wave (k_lm_quad.hip) once the batch is large enough to fill the chip that way; the windows it leaves out
    // (a rotation vector outside the |theta| <= pi ball) are stepped by the one-wave kernel right behind it
    const bool quad = (env ? env[0] == '3' : a.B >= QUAD_MIN_BATCH) && lm_step_quad_fits(a);
    if (quad) {
        launch_lm_step_quad(a, s);
        StepArgs a2 = a;
        a2.only_slow = 1;
        hipLaunchKernelGGL(k_lm_step_slow, dim3((unsigned)((a.B + 63) / 64)), dim3(64), 0, s, a2);
        return;
    }
    // a two-frame window (what the reference's tracking loop solves every laser frame) is ONE dense 30 x 30 system; variant 4 forces it
    const bool dense2 = a.n == 2 && (env ? env[0] == '4' : true);
    if (dense2) hipLaunchKernelGGL(k_lm_step_dense2, dim3(a.B), dim3(64), 0, s, a);
    else if (tw) hipLaunchKernelGGL(k_lm_step_tw, dim3(a.B), dim3(256), 0, s, a);
    else if (tp) hipLaunchKernelGGL(k_lm_step<true>, dim3(a.B), dim3(64), 0, s, a);
    else hipLaunchKernelGGL(k_lm_step<false>, dim3(a.B), dim3(64), 0, s, a);
}
void launch_lm_finish(const StepArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_lm_finish, dim3((a.B * a.n * 6 + 255) / 256), dim3(256), 0, s, a); }
void launch_export_dense(const ExportArgs& a, hipStream_t s) { hipLaunchKernelGGL(k_export_dense, dim3(a.B), dim3(64), 0, s, a); }
void launch_marg_schur(const MargArgs& a, hipStream_t s) {
    static const char* env = getenv("LIW_MARG_WAVES");   // 1 / 4: force the one-wave / four-wave kernel (profiling / test aid)
    const bool four = env ? env[0] == '4' : a.B <= 256;  // four waves per window while that takes no CUs away from other windows
    const char* eig = getenv("LIW_MARG_EIG");            // (read per launch) 1: the eigen square root by the wave that ran the chain (k_marg_schur, until the end of round 5: A/B aid)
    if (four) hipLaunchKernelGGL(k_marg_schur4, dim3(a.B), dim3(256), 0, s, a);
    else if (eig && eig[0] == '1') hipLaunchKernelGGL(k_marg_schur, dim3(a.B), dim3(64), 0, s, a);
    else {
        hipLaunchKernelGGL(k_marg_schur_chain, dim3(a.B), dim3(64), 0, s, a);
        hipLaunchKernelGGL(k_marg_schur_eigq, dim3((a.B + 3) / 4), dim3(64), 0, s, a);
    }
}

}  // namespace liw
