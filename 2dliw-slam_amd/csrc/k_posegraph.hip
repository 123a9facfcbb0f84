// k_posegraph.hip — back-end pose-graph relinearisation on the MI355X (C ABI include/liw_posegraph.h).
//
// keyframe_manager::solve (reference src/trajectory/keyframe_manager.cpp:722-838) = Ceres trust-region LM over the key-frame
// poses.  Here:
//   k_pg_linearize   one 16-lane group per edge_factor block (src/factor/edge_factor.h:79-126): 12 dual directions + value,
//                    so3 local parameterisation applied in the group, Y = [J | r] (6 x 13) per edge to HBM; ground factors
//                    per key frame the same way (7 lanes, Y 2 x 7).
//   k_pg_assemble    dense H = sum Y^T Y (6 unknowns per key frame): one wave per key frame gathers its incident edges
//                    (CSR built on the host) -> deterministic diagonal blocks and gradient; off-diagonal 6x6 blocks are
//                    written by the edge that owns the pair.
//   k_pg_scale_damp  A = S H S + diag(D / radius)  (Jacobi scaling, LM diagonal), the matrix that is factorised.
//   blocked Cholesky k_potrf64 (register-resident, one wave: column per lane, v_readlane broadcasts), k_trsm64 (row per
//                    lane against the LDS-resident diagonal factor), k_syrk64 (trailing update on the fp64 matrix cores:
//                    v_mfma_f64_16x16x4_f64, operands staged through LDS), k_trisolve_pipe (L y = b, L^T x = y as a flag-synchronised pipeline over the blocks).
// The trust-region bookkeeping (radius, step acceptance, the three Ceres tolerances) runs on the host between launches;
// the per-iteration device->host traffic is the step vector (6N doubles) and two cost scalars.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/liw_posegraph.h"
#include "liw_kernels.hpp"

struct liw_ctx;
hipStream_t liw_ctx_stream(liw_ctx* c);
const liw::DevParams* liw_ctx_devparams(liw_ctx* c);
int liw_ctx_device(liw_ctx* c);
bool liw_ctx_has_device(liw_ctx* c);
int liw_ctx_fail(liw_ctx* c, int code, const char* what);

namespace liw {

typedef double d4g __attribute__((ext_vector_type(4)));
constexpr double kPiG = 3.141592653589793238462643383279, kTwoPiG = 6.283185307179586476925286766559;

struct PgNoise { double J[36]; double ground_on_p, ground_on_q; };

__device__ __forceinline__ double rdl64(double v, int l) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}
// d Plus(x, delta) / d delta at delta = 0 of the so3 parameterisation (src/factor/factor_common.h:37-60): identity unless |x| > pi
__device__ __forceinline__ bool plus_jac(const double* x, double* P9) {
    const double a = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    if (!(a > kPiG)) return false;
    const double k = floor((a + kPiG) / kTwoPiG);
    const double c = kTwoPiG * k / a;
    const double u[3] = {x[0] / a, x[1] / a, x[2] / a};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) P9[i * 3 + j] = (i == j ? 1.0 : 0.0) - c * ((i == j ? 1.0 : 0.0) - u[i] * u[j]);
    return true;
}

// ---- edge_factor::operator() (edge_factor.h:88-117): res = weight * J_noise * log_SE3(tf_j^-1 tf_i tf12)
template <class T>
__device__ __forceinline__ void edge_res(const double* tf12, double weight, const double* Jn, const T* pi, const T* qi, const T* pj, const T* qj, T* res) {
    Iso<T> tf_i = make_tf(V3<T>(pi[0], pi[1], pi[2]), V3<T>(qi[0], qi[1], qi[2]));
    Iso<T> tf_j = make_tf(V3<T>(pj[0], pj[1], pj[2]), V3<T>(qj[0], qj[1], qj[2]));
    Iso<T> err = mul(mul(inverse(tf_j), tf_i), cast_iso<T>(tf12, tf12 + 9));
    V3<T> rq = log_SO3(err.R);
    T raw[6] = {err.t.x, err.t.y, err.t.z, rq.x, rq.y, rq.z};
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        T s(0.0);
#pragma unroll
        for (int c = 0; c < 6; ++c) s = s + T(Jn[r * 6 + c]) * raw[c];
        res[r] = T(weight) * s;
    }
}
template <class T>
__device__ __forceinline__ void pg_ground_res(const DevParams& P, const T* p_, const T* q_, T* res) {   // ground_factor.h:27-82
    Iso<T> tf_w_o = mul(make_tf(V3<T>(p_[0], p_[1], p_[2]), V3<T>(q_[0], q_[1], q_[2])), cast_iso<T>(P.Riw, P.tiw));
    res[0] = T(P.ground_p_info) * tf_w_o.t.z;
    V3<T> ABC(T(0.0), T(0.0), T(1.0));
    V3<T> z_axis(tf_w_o.R(0, 2), tf_w_o.R(1, 2), tf_w_o.R(2, 2));
    T sinn = norm(cross(z_axis, ABC));
    res[1] = T(P.ground_q_info) * dasin(sinn);
}

// Y_edge [E][6][13], Y_ground [N][2][7]; jac = 0: residuals only (column 12 / 6)
__global__ __launch_bounds__(64) void k_pg_linearize(int N, int E, const double* x, const int* eidx, const double* etf, const double* ew, PgNoise noise,
                                                     DevParams P, double* Ye, double* Yg, int jac) {
    __shared__ double Y[4][6 * 13];
    const int lane = threadIdx.x & 63, grp = lane >> 4, d = lane & 15;
    const int edge_blocks = (E + 3) / 4;
    if ((int)blockIdx.x < edge_blocks) {
        const int e = blockIdx.x * 4 + grp;
        const bool on = e < E;
        const int i = on ? eidx[e * 2] : 0, j = on ? eidx[e * 2 + 1] : 0;
        const double* xi = x + (size_t)i * 6;
        const double* xj = x + (size_t)j * 6;
        if (on && (d == 12 || (jac && d < 12))) {
            LJ pi[3], qi[3], pj[3], qj[3], res[6];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                pi[k] = LJ(xi[k], d == k ? 1.0 : 0.0);
                qi[k] = LJ(xi[3 + k], d == 3 + k ? 1.0 : 0.0);
                pj[k] = LJ(xj[k], d == 6 + k ? 1.0 : 0.0);
                qj[k] = LJ(xj[3 + k], d == 9 + k ? 1.0 : 0.0);
            }
            edge_res<LJ>(etf + (size_t)e * 12, ew[e], noise.J, pi, qi, pj, qj, res);
#pragma unroll
            for (int r = 0; r < 6; ++r) Y[grp][r * 13 + d] = d < 12 ? res[r].d : res[r].v;
        }
        __syncthreads();
        if (on && jac && d < 12) {   // local parameterisation: J_theta <- J_theta * dPlus/ddelta  (lane = (row r, pose which))
            const int r = d % 6, which = d / 6;
            double Pq[9];
            if (plus_jac((which ? xj : xi) + 3, Pq)) {
                double* row = &Y[grp][r * 13 + 6 * which + 3];
                const double t0 = row[0] * Pq[0] + row[1] * Pq[3] + row[2] * Pq[6];
                const double t1 = row[0] * Pq[1] + row[1] * Pq[4] + row[2] * Pq[7];
                const double t2 = row[0] * Pq[2] + row[1] * Pq[5] + row[2] * Pq[8];
                row[0] = t0; row[1] = t1; row[2] = t2;
            }
        }
        __syncthreads();
        if (on && d < 13)
#pragma unroll
            for (int r = 0; r < 6; ++r) Ye[(size_t)e * 78 + r * 13 + d] = (jac || d == 12) ? Y[grp][r * 13 + d] : 0.0;
    } else {
        // ground factors: 8 key frames per wave, 6 dual directions + value
        const int sub = lane >> 3, dir = lane & 7;
        const int i = ((int)blockIdx.x - edge_blocks) * 8 + sub;
        const bool on = i < N;
        double* Yl = &Y[0][0] + sub * 16;
        if (on && (dir == 6 || (jac && dir < 6))) {
            const double* s_ = x + (size_t)i * 6;
            LJ p[3], q[3], res[2];
#pragma unroll
            for (int k = 0; k < 3; ++k) { p[k] = LJ(s_[k], dir == k ? 1.0 : 0.0); q[k] = LJ(s_[3 + k], dir == 3 + k ? 1.0 : 0.0); }
            pg_ground_res<LJ>(P, p, q, res);
            Yl[dir] = (dir < 6 ? res[0].d : res[0].v) * noise.ground_on_p;
            Yl[7 + dir] = (dir < 6 ? res[1].d : res[1].v) * noise.ground_on_q;
        }
        __syncthreads();
        if (on && jac && dir < 2) {
            double Pq[9];
            if (plus_jac(x + (size_t)i * 6 + 3, Pq)) {
                double* row = Yl + dir * 7 + 3;
                const double t0 = row[0] * Pq[0] + row[1] * Pq[3] + row[2] * Pq[6];
                const double t1 = row[0] * Pq[1] + row[1] * Pq[4] + row[2] * Pq[7];
                const double t2 = row[0] * Pq[2] + row[1] * Pq[5] + row[2] * Pq[8];
                row[0] = t0; row[1] = t1; row[2] = t2;
            }
        }
        __syncthreads();
        if (on && dir < 7) {
            Yg[(size_t)i * 14 + dir] = (jac || dir == 6) ? Yl[dir] : 0.0;
            Yg[(size_t)i * 14 + 7 + dir] = (jac || dir == 6) ? Yl[7 + dir] : 0.0;
        }
    }
}

// cost = 1/2 sum r^2 over all blocks (deterministic two-stage sum)
__global__ void k_pg_cost(int N, int E, const double* Ye, const double* Yg, int const_pose, double* partial) {
    __shared__ double red[256];
    double s = 0.0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < E * 6 + N * 2; t += gridDim.x * blockDim.x) {
        double r;
        if (t < E * 6) r = Ye[(size_t)(t / 6) * 78 + (t % 6) * 13 + 12];
        else { const int u = t - E * 6; r = (u / 2 == const_pose) ? 0.0 : Yg[(size_t)(u / 2) * 14 + (u % 2) * 7 + 6]; }   // all-constant blocks are not part of the cost
        s += r * r;
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// one wave per key frame: diagonal 6x6 block + gradient from every incident block (CSR: inc_off[N+1], inc[.] = edge*2+side),
// and the off-diagonal blocks of the edges whose index1 is this key frame.  Constant key frame: identity block, zero gradient.
__global__ __launch_bounds__(64) void k_pg_assemble(int N, int ld, const int* inc_off, const int* inc, const int* eidx, const double* Ye, const double* Yg,
                                                    int const_pose, double* H, double* g) {
    const int i = blockIdx.x, lane = threadIdx.x & 63;
    if (i >= N) return;
    const int r = lane / 6, c = lane % 6;     // lanes 0..35: entry (r, c) of the diagonal block; 36..41: gradient entry
    double acc = 0.0;
    const bool is_const = i == const_pose;
    if (!is_const) {
        if (lane < 36) acc = Yg[(size_t)i * 14 + r] * Yg[(size_t)i * 14 + c] + Yg[(size_t)i * 14 + 7 + r] * Yg[(size_t)i * 14 + 7 + c];
        else if (lane < 42) acc = Yg[(size_t)i * 14 + (lane - 36)] * Yg[(size_t)i * 14 + 6] + Yg[(size_t)i * 14 + 7 + (lane - 36)] * Yg[(size_t)i * 14 + 13];
        for (int t = inc_off[i]; t < inc_off[i + 1]; ++t) {
            const int e = inc[t] >> 1, side = inc[t] & 1;
            const double* Y = Ye + (size_t)e * 78;
            double s = 0.0;
            if (lane < 36) {
#pragma unroll
                for (int k = 0; k < 6; ++k) s += Y[k * 13 + 6 * side + r] * Y[k * 13 + 6 * side + c];
            } else if (lane < 42) {
#pragma unroll
                for (int k = 0; k < 6; ++k) s += Y[k * 13 + 6 * side + (lane - 36)] * Y[k * 13 + 12];
            }
            acc += s;
            // off-diagonal block H[j, i] of the pair (i, j), j > i (lower triangle): added by the wave of the LOWER key frame only, whichever
            // side of the edge it is, in the order of its incidence list (= edge order).  Any number of edges between the same pair —
            // in either direction — sums in that one order, by one wave: bit-reproducible without atomics (round 6; until then the side-0
            // wave added with fp64 atomicAdd, which is order-free for two terms only).  H is zeroed in front of this kernel.
            {
                const int j = eidx[e * 2 + (1 - side)];
                if (j != const_pose && j > i && lane < 36) {
                    double o = 0.0;   // (J_j^T J_i)(r, c): J_i = the columns of this key frame's side, J_j = the other side's
#pragma unroll
                    for (int k = 0; k < 6; ++k) o += Y[k * 13 + 6 * (1 - side) + r] * Y[k * 13 + 6 * side + c];
                    H[(size_t)(j * 6 + r) * ld + i * 6 + c] += o;
                }
            }
        }
    } else {
        if (lane < 36) acc = r == c ? 1.0 : 0.0;
    }
    if (lane < 36) { if (r >= c) H[(size_t)(i * 6 + r) * ld + i * 6 + c] = acc; }
    else if (lane < 42) g[i * 6 + (lane - 36)] = acc;
}

__global__ void k_pg_diag(int n, int ld, const double* H, double* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = H[(size_t)i * ld + i];
}
// A = S H S + diag(D / radius) on the lower triangle; padding rows get the identity
__global__ void k_pg_scale_damp(int n, int np, int ld, const double* H, const double* scale, const double* dgn, double radius, double* A) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (col >= np || row >= np || col > row) return;
    double v;
    if (row >= n) v = row == col ? 1.0 : 0.0;
    else {
        v = H[(size_t)row * ld + col] * scale[row] * scale[col];
        if (row == col) { const double lm = sqrt(dgn[row] / radius); v += lm * lm; }
    }
    A[(size_t)row * ld + col] = v;
}

// ---------------------------------------------------------------------------------------------- blocked Cholesky, NB = 64
// diagonal block: lane j owns column j (a[r] = A[r][j]); step k broadcasts the pivot and column k with v_readlane;
// afterwards lane j holds row j of L in a[0..j].  status[0] != 0 when a pivot is not positive.
__global__ __launch_bounds__(64, 1) void k_potrf64(double* A, int ld, int k0, int* status) {
    const int lane = threadIdx.x & 63;
    double* B = A + (size_t)k0 * ld + k0;
    double a[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) a[r] = r >= lane ? B[(size_t)r * ld + lane] : B[(size_t)lane * ld + r];   // full symmetric block from the lower triangle
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 64; ++k) {
        const double piv = rdl64(a[k], k);
        if (!(piv > 0.0)) ok = false;
        const double inv = 1.0 / sqrt(piv);
        const double wk = a[k] * inv;          // lane j >= k: A[k][j] / sqrt(piv) = L[j][k]
        a[k] = wk;
#pragma unroll
        for (int r = k + 1; r < 64; ++r) a[r] -= rdl64(wk, r) * wk;
    }
#pragma unroll
    for (int k = 0; k < 64; ++k) if (k <= lane) B[(size_t)lane * ld + k] = a[k];   // row `lane` of L
    if (!ok && lane == 0) status[0] = 1;
}

// panel: X L_kk^T = A_ik  for every block row i > k; one wave per block row, lane = row of the block
__global__ __launch_bounds__(64) void k_trsm64(double* A, int ld, int k0) {
    __shared__ double L[64 * 65];
    const int lane = threadIdx.x & 63;
    const int i0 = k0 + 64 * (blockIdx.x + 1);
    const double* D = A + (size_t)k0 * ld + k0;
    for (int r = 0; r < 64; ++r) L[r * 65 + lane] = lane <= r ? D[(size_t)r * ld + lane] : 0.0;
    __syncthreads();
    double* R = A + (size_t)(i0 + lane) * ld + k0;
    double x[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) x[c] = R[c];
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        double s = x[c];
#pragma unroll
        for (int m = 0; m < c; ++m) s -= x[m] * L[c * 65 + m];
        x[c] = s / L[c * 65 + c];
    }
#pragma unroll
    for (int c = 0; c < 64; ++c) R[c] = x[c];
}

// trailing update A_ij -= A_ik A_jk^T (j <= i), 64x64 tile per work-group of 4 waves, fp64 MFMA 16x16x4
__global__ __launch_bounds__(256) void k_syrk64(double* A, int ld, int k0, int nb_rem) {
    __shared__ double Pa[64 * 33], Pb[64 * 33];     // K halves of 32, padded
    const int ti = blockIdx.y, tj = blockIdx.x;
    if (tj > ti || ti >= nb_rem) return;
    const int i0 = k0 + 64 * (ti + 1), j0 = k0 + 64 * (tj + 1);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    d4g acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int kh = 0; kh < 2; ++kh) {
        __syncthreads();
        for (int e = tid; e < 64 * 32; e += 256) {
            const int r = e >> 5, c = e & 31;
            Pa[r * 33 + c] = A[(size_t)(i0 + r) * ld + k0 + 32 * kh + c];
            Pb[r * 33 + c] = A[(size_t)(j0 + r) * ld + k0 + 32 * kh + c];
        }
        __syncthreads();
        // wave w: rows 16w..16w+15 of the tile; MFMA operand A[i][k] from lane (i = l & 15, k = l >> 4), B[k][j] likewise
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int kk = 4 * ks + (lane >> 4);
            const double av = Pa[(16 * wave + (lane & 15)) * 33 + kk];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const double bv = Pb[(16 * ct + (lane & 15)) * 33 + kk];
                acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[ct], 0, 0, 0);
            }
        }
    }
    // acc[ct][r] = C[16w + (lane>>4) + 4r][16ct + (lane&15)]
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * wave + (lane >> 4) + 4 * r, col = 16 * ct + (lane & 15);
            if (ti != tj || col <= row) A[(size_t)(i0 + row) * ld + j0 + col] -= acc[ct][r];
        }
}

// Triangular solves L y = b (DIR = 0) and L^T x = y (DIR = 1) as a pipeline over the 64-unknown blocks: work-group i owns
// block i, consumes the solved blocks it depends on as soon as their flag is published (tile of L through LDS, 64x64
// mat-vec), then solves its diagonal block in one wave (row / column per lane, v_readlane exchange) and publishes its own
// flag.  Work-groups are dispatched in index order and only wait on blocks that were dispatched before them (forward: lower
// indices; backward: the grid is reversed), so the spin-waits cannot deadlock even when not all groups are resident.
template <int DIR>
__global__ __launch_bounds__(256) void k_trisolve_pipe(const double* L, int ld, int nb, double* b, int* flags) {
    __shared__ double tile[64 * 65];
    __shared__ double xs[64], part[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = DIR == 0 ? (int)blockIdx.x : nb - 1 - (int)blockIdx.x;
    const int i0 = 64 * i;
    double acc = 0.0;   // thread (wave, lane): partial sum of unknown `lane` over the 16 tile rows / columns of its quarter
    const int kbeg = DIR == 0 ? 0 : nb - 1, kend = i, kstep = DIR == 0 ? 1 : -1;
    for (int k = kbeg; k != kend; k += kstep) {
        const int k0 = 64 * k;
        if (tid == 0) { while (__atomic_load_n(&flags[k], __ATOMIC_ACQUIRE) == 0) __builtin_amdgcn_s_sleep(1); }
        __syncthreads();
        // forward: tile = L[i-block rows][k-block cols]; backward: tile = L[k-block rows][i-block cols] (used transposed)
        const double* T = DIR == 0 ? L + (size_t)i0 * ld + k0 : L + (size_t)k0 * ld + i0;
        for (int e = tid; e < 64 * 64; e += 256) { const int r = e >> 6, c = e & 63; tile[r * 65 + c] = T[(size_t)r * ld + c]; }
        if (tid < 64) xs[tid] = __builtin_nontemporal_load(&b[k0 + tid]);
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const int q = 16 * wave + m;
            s += (DIR == 0 ? tile[lane * 65 + q] : tile[q * 65 + lane]) * xs[q];
        }
        acc += s;
        __syncthreads();
    }
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
        double v = b[i0 + lane] - ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]));
        double Ld[64];   // forward: row `lane` of the diagonal block; backward: column `lane`
#pragma unroll
        for (int m = 0; m < 64; ++m) Ld[m] = DIR == 0 ? L[(size_t)(i0 + lane) * ld + i0 + m] : L[(size_t)(i0 + m) * ld + i0 + lane];
        if (DIR == 0) {
#pragma unroll
            for (int m = 0; m < 64; ++m) {
                const double ym = rdl64(v, m) / rdl64(Ld[m], m);
                if (lane == m) v = ym;
                else if (lane > m) v -= Ld[m] * ym;
            }
        } else {
#pragma unroll
            for (int m = 63; m >= 0; --m) {
                const double xm = rdl64(v, m) / rdl64(Ld[m], m);
                if (lane == m) v = xm;
                else if (lane < m) v -= Ld[m] * xm;
            }
        }
        b[i0 + lane] = v;
        __threadfence();
        if (lane == 0) __atomic_store_n(&flags[i], 1, __ATOMIC_RELEASE);
    }
}

static int dense_cholesky_solve(liw_ctx* c, double* dA, int np, double* db, int* dstatus, hipStream_t s) {
    // dstatus: [0] = not-positive-definite flag, [1 .. nb] forward flags, [1 + nb .. 2 nb] backward flags
    const int nb = np / 64;
    // the pipelined triangular solves spin on flags of earlier blocks: keep every work-group co-resident (256 CUs x 8 groups of
    // 256 threads) so that the scheme does not depend on the dispatch order
    if (nb > 2048) return liw_ctx_fail(c, LIW_EINVAL, "dense solver: more than 131072 unknowns");
    (void)hipMemsetAsync(dstatus, 0, sizeof(int) * (1 + 2 * (size_t)nb), s);
    for (int k = 0; k < nb; ++k) {
        hipLaunchKernelGGL(k_potrf64, dim3(1), dim3(64), 0, s, dA, np, 64 * k, dstatus);
        const int rem = nb - k - 1;
        if (rem > 0) {
            hipLaunchKernelGGL(k_trsm64, dim3(rem), dim3(64), 0, s, dA, np, 64 * k);
            hipLaunchKernelGGL(k_syrk64, dim3(rem, rem), dim3(256), 0, s, dA, np, 64 * k, rem);
        }
    }
    hipLaunchKernelGGL(k_trisolve_pipe<0>, dim3(nb), dim3(256), 0, s, dA, np, nb, db, dstatus + 1);
    hipLaunchKernelGGL(k_trisolve_pipe<1>, dim3(nb), dim3(256), 0, s, dA, np, nb, db, dstatus + 1 + nb);
    if (hipGetLastError() != hipSuccess) return liw_ctx_fail(c, LIW_EHIP, "dense Cholesky launch");
    return LIW_OK;
}

}  // namespace liw

using namespace liw;

namespace {
struct DBuf {
    void* p = nullptr;
    ~DBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 8) == hipSuccess ? 0 : -1; }
    template <class T> T* as() { return (T*)p; }
};
// Plus of the so3 parameterisation on the host (src/factor/factor_common.h:41-53: the rotation VECTORS add, then wrap)
void so3_plus_host(const double* x, const double* d, double* out) {
    V3<double> r = normalize_so3(V3<double>(x[0] + d[0], x[1] + d[1], x[2] + d[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
}  // namespace

extern "C" {

int liw_dense_spd_solve(liw_ctx* c, int n, const double* A, const double* b, double* x) {
    if (!liw_ctx_has_device(c)) return liw_ctx_fail(c, LIW_ENODEV, "no usable gfx950 device (this library has no CPU fallback)");
    if (n < 1 || !A || !b || !x) return liw_ctx_fail(c, LIW_EINVAL, "liw_dense_spd_solve: bad argument");
    (void)hipSetDevice(liw_ctx_device(c));
    hipStream_t s = liw_ctx_stream(c);
    const int np = (n + 63) / 64 * 64;
    std::vector<double> Ap((size_t)np * np, 0.0), bp(np, 0.0);
    for (int i = 0; i < np; ++i) {
        if (i < n) { for (int j = 0; j <= i; ++j) Ap[(size_t)i * np + j] = A[(size_t)i * n + j]; bp[i] = b[i]; }
        else Ap[(size_t)i * np + i] = 1.0;
    }
    DBuf dA, db, dst;
    if (dA.alloc(sizeof(double) * (size_t)np * np) || db.alloc(sizeof(double) * np) || dst.alloc(sizeof(int) * (1 + 2 * (size_t)(np / 64)))) return liw_ctx_fail(c, LIW_ENOMEM, "hipMalloc");
    (void)hipMemcpyAsync(dA.p, Ap.data(), sizeof(double) * (size_t)np * np, hipMemcpyHostToDevice, s);
    (void)hipMemcpyAsync(db.p, bp.data(), sizeof(double) * np, hipMemcpyHostToDevice, s);
    if (int r = dense_cholesky_solve(c, dA.as<double>(), np, db.as<double>(), dst.as<int>(), s)) return r;
    int st = 0;
    (void)hipMemcpyAsync(bp.data(), db.p, sizeof(double) * np, hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(&st, dst.p, sizeof(int), hipMemcpyDeviceToHost, s);
    if (hipStreamSynchronize(s) != hipSuccess) return liw_ctx_fail(c, LIW_EHIP, "hipStreamSynchronize");
    if (st) return liw_ctx_fail(c, LIW_ESTATE, "matrix is not positive definite");
    std::memcpy(x, bp.data(), sizeof(double) * n);
    return LIW_OK;
}

}  // extern "C"
namespace liw {
// ------------------------------------------------------------------------------------------- sparse path
// The key frames form a chain (sequential edges k -> k+1, keyframe_manager.cpp:444-453) closed by a few loop edges.  Key frames
// that carry a loop edge, the constant one, both chain ends and every PG_STRIDE-th frame are SEPARATORS; the frames between two
// consecutive separators a < b are interior: tied to their chain neighbours only.  Per LM iteration
//   k_pg_seg_elim     one wave per segment eliminates its interior frames a+1 .. b-1 in chain order (6x6 pivots, register-resident
//                     Cholesky with v_readlane broadcasts; the coupling to the left border a is the only fill) and leaves the Schur
//                     terms on (a,a), (b,a), (b,b), g_a, g_b plus one 6x13 back-substitution record per interior frame;
//   k_pg_reduced(+loops)  assembles the dense system of the separators only (a few hundred unknowns instead of 6N);
//   the blocked MFMA Cholesky below solves that; k_pg_seg_backsub recovers the interior frames, again one wave per segment.
// Exact (the same linear system as the dense path, which stays as the checker: LIW_PG_DENSE=1 or a non-chain edge list).
constexpr int PG_STRIDE = 48;

__device__ __forceinline__ double rdl64(double v, int l);

// one wave per key frame: 6x6 diagonal block + gradient (as k_pg_assemble) into Dd / gd; the off-diagonal block H[index1, index2]
// of every edge whose index1 is this key frame into Os[index1] (sequential edge) or Lb[e - n_seq] (loop edge)
__global__ __launch_bounds__(64) void k_pg_assemble_blocks(int N, int n_seq, const int* inc_off, const int* inc, const int* eidx, const double* Ye, const double* Yg,
                                                           int const_pose, double* Dd, double* gd, double* Os, double* Lb) {
    const int i = blockIdx.x, lane = threadIdx.x & 63;
    if (i >= N) return;
    const int r = lane / 6, c = lane % 6;
    double acc = 0.0;
    const bool is_const = i == const_pose;
    if (!is_const) {
        if (lane < 36) acc = Yg[(size_t)i * 14 + r] * Yg[(size_t)i * 14 + c] + Yg[(size_t)i * 14 + 7 + r] * Yg[(size_t)i * 14 + 7 + c];
        else if (lane < 42) acc = Yg[(size_t)i * 14 + (lane - 36)] * Yg[(size_t)i * 14 + 6] + Yg[(size_t)i * 14 + 7 + (lane - 36)] * Yg[(size_t)i * 14 + 13];
    }
    for (int t = inc_off[i]; t < inc_off[i + 1]; ++t) {
        const int e = inc[t] >> 1, side = inc[t] & 1;
        const double* Y = Ye + (size_t)e * 78;
        if (!is_const) {
            double sm = 0.0;
            if (lane < 36) {
#pragma unroll
                for (int k = 0; k < 6; ++k) sm += Y[k * 13 + 6 * side + r] * Y[k * 13 + 6 * side + c];
            } else if (lane < 42) {
#pragma unroll
                for (int k = 0; k < 6; ++k) sm += Y[k * 13 + 6 * side + (lane - 36)] * Y[k * 13 + 12];
            }
            acc += sm;
        }
        if (side == 0 && lane < 36) {   // H[index1, index2](r, c) = sum_k J1[k][r] J2[k][c]; zero against the constant key frame
            const int j = eidx[e * 2 + 1];
            double o = 0.0;
            if (!is_const && j != const_pose) {
#pragma unroll
                for (int k = 0; k < 6; ++k) o += Y[k * 13 + r] * Y[k * 13 + 6 + c];
            }
            if (e < n_seq) Os[(size_t)i * 36 + lane] = o;
            else Lb[(size_t)(e - n_seq) * 36 + lane] = o;
        }
    }
    if (is_const && lane < 36) acc = r == c ? 1.0 : 0.0;
    if (lane < 36) Dd[(size_t)i * 36 + lane] = acc;
    else if (lane < 42) gd[i * 6 + (lane - 36)] = acc;
}

// Schur terms of one segment (doubles): aa[36] (on H[a,a]), ba[36] (H[b,a], rows b), bb[36], ga[6], gb[6]
constexpr int PG_SEG = 36 * 3 + 12;
__global__ __launch_bounds__(64) void k_pg_seg_elim(int nseg, const int* segs, const double* Dd, const double* gd, const double* Os, const double* scale,
                                                    const double* dgn, double radius, double* Sred, double* rec, int* status) {
    const int sg = blockIdx.x, lane = threadIdx.x & 63;
    if (sg >= nseg) return;
    const int a = segs[2 * sg], b = segs[2 * sg + 1];
    double* out = Sred + (size_t)sg * PG_SEG;
    // lane roles: j < 6 column j of the pivot block, 6..11 columns of H[f, f+1], 12..17 columns of H[f, a], 18 the gradient
    const int role = lane < 6 ? 0 : (lane < 12 ? 1 : (lane < 18 ? 2 : (lane == 18 ? 3 : 4)));
    const int cj = lane < 18 ? lane % 6 : 0;
    double cD[6] = {0, 0, 0, 0, 0, 0}, cB[6] = {0, 0, 0, 0, 0, 0}, cg[6] = {0, 0, 0, 0, 0, 0}, aAA[6] = {0, 0, 0, 0, 0, 0}, aGa[6] = {0, 0, 0, 0, 0, 0};
    bool ok = true;
    for (int f = a + 1; f < b; ++f) {
        const double* sf = scale + (size_t)f * 6;
        double col[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double v = 0.0;
            if (role == 0) v = Dd[(size_t)f * 36 + r * 6 + cj] * sf[r] * sf[cj] + cD[r] + (r == cj ? dgn[(size_t)f * 6 + cj] / radius : 0.0);
            else if (role == 1) v = Os[(size_t)f * 36 + r * 6 + cj] * sf[r] * scale[(size_t)(f + 1) * 6 + cj];
            else if (role == 2) v = (f == a + 1 ? Os[(size_t)a * 36 + cj * 6 + r] * scale[(size_t)a * 6 + cj] * sf[r] : 0.0) + cB[r];
            else if (role == 3) v = gd[(size_t)f * 6 + r] * sf[r] + cg[r];
            col[r] = v;
        }
        // right-looking Cholesky of the 6x6 pivot fused with the forward substitution of the other columns
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            const double piv = rdl64(col[k], k);
            if (!(piv > 0.0) || !isfinite(piv)) ok = false;
            const double wk = col[k] * rsqrt(piv);
            col[k] = wk;
#pragma unroll
            for (int r = k + 1; r < 6; ++r) col[r] -= rdl64(wk, r) * wk;
        }
        // record [Yo | Yb | yz] = L^-T [Wo | Wb | z]: y_f = yz - Yo y_{f+1} - Yb y_a
        {
            double xs[6];
#pragma unroll
            for (int k = 5; k >= 0; --k) {
                double t = col[k];
#pragma unroll
                for (int r = k + 1; r < 6; ++r) t -= rdl64(col[k], r) * xs[r];      // L[r][k] lives in lane r, register k
                xs[k] = t / rdl64(col[k], k);
            }
            if (lane >= 6 && lane < 19) {
#pragma unroll
                for (int k = 0; k < 6; ++k) rec[(size_t)f * 78 + k * 13 + (lane - 6)] = xs[k];
            }
        }
        // Schur terms: P(p, own) = W_p . W_own for the columns p of [Wo | Wb]
        double P[12];
#pragma unroll
        for (int p_ = 0; p_ < 12; ++p_) {
            double d = 0.0;
#pragma unroll
            for (int k = 0; k < 6; ++k) d += rdl64(col[k], 6 + p_) * col[k];
            P[p_] = d;
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const double fromO = __shfl(P[q], (lane + 6) & 63, 64);   // lane j < 6 <- P(Wo_q, Wo_j) held by lane 6 + j
            if (role == 0) cD[q] = -fromO;
            if (role == 2) { cB[q] = -P[q]; aAA[q] -= P[6 + q]; }
            if (role == 3) { cg[q] = -P[q]; aGa[q] -= P[6 + q]; }
        }
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) {
        if (role == 0) out[72 + r * 6 + cj] = cD[r];                                 // bb
        if (role == 2) { out[36 + r * 6 + cj] = cB[r]; out[r * 6 + cj] = aAA[r]; }   // ba (rows b), aa
        if (role == 3) { out[108 + r] = aGa[r]; out[114 + r] = cg[r]; }              // ga, gb
    }
    if (!ok && lane == 0) atomicOr(status, 1);
}

// dense system of the separators (lower triangle): one wave per separator t (frame sep[t])
__global__ __launch_bounds__(64) void k_pg_reduced(int ns, int nr, int ld, const int* sep, const double* Dd, const double* gd, const double* Os, const double* scale,
                                                   const double* dgn, double radius, int const_pose, const double* Sred, double* Hr, double* rhs) {
    const int t = blockIdx.x, lane = threadIdx.x & 63;
    if (t >= ns) {   // padding rows of the blocked factorisation: identity
        for (int k = nr + lane; k < ld; k += 64) Hr[(size_t)k * ld + k] = 1.0;
        return;
    }
    const int f = sep[t], r = lane / 6, c = lane % 6;
    const double* sf = scale + (size_t)f * 6;
    const bool is_const = f == const_pose;
    const bool left = t > 0 && sep[t] - sep[t - 1] > 1, right = t + 1 < ns && sep[t + 1] - sep[t] > 1;   // segments with interior frames
    if (lane < 36) {
        double v = is_const ? (r == c ? 1.0 : 0.0) : Dd[(size_t)f * 36 + lane] * sf[r] * sf[c] + (r == c ? dgn[(size_t)f * 6 + c] / radius : 0.0);
        if (!is_const) {
            if (right) v += Sred[(size_t)t * PG_SEG + lane];
            if (left) v += Sred[(size_t)(t - 1) * PG_SEG + 72 + lane];
        }
        if (r >= c) Hr[(size_t)(t * 6 + r) * ld + t * 6 + c] = v;
        if (t > 0) {   // block (t, t-1): direct edge when the separators are chain neighbours, else the segment's fill
            const int fa = sep[t - 1];
            const double o = left ? Sred[(size_t)(t - 1) * PG_SEG + 36 + lane] : Os[(size_t)fa * 36 + c * 6 + r] * scale[(size_t)fa * 6 + c] * sf[r];
            Hr[(size_t)(t * 6 + r) * ld + (t - 1) * 6 + c] = (is_const || fa == const_pose) ? 0.0 : o;
        }
    } else if (lane < 42) {
        const int k = lane - 36;
        double v = is_const ? 0.0 : gd[(size_t)f * 6 + k] * sf[k];
        if (!is_const) {
            if (right) v += Sred[(size_t)t * PG_SEG + 108 + k];
            if (left) v += Sred[(size_t)(t - 1) * PG_SEG + 114 + k];
        }
        rhs[t * 6 + k] = v;
    }
}
// loop edges, in edge order by ONE wave (several edges may share a block: fixed order keeps the sums reproducible)
__global__ __launch_bounds__(64) void k_pg_reduced_loops(int nl, const int* lidx, const int* sepidx, const double* Lb, const double* scale, int const_pose, int ld, double* Hr) {
    const int lane = threadIdx.x & 63, r = lane / 6, c = lane % 6;
    if (lane >= 36) return;
    for (int e = 0; e < nl; ++e) {
        const int i1 = lidx[2 * e], i2 = lidx[2 * e + 1];
        if (i1 == const_pose || i2 == const_pose) continue;
        const int t1 = sepidx[i1], t2 = sepidx[i2];
        const double v = Lb[(size_t)e * 36 + lane] * scale[(size_t)i1 * 6 + r] * scale[(size_t)i2 * 6 + c];   // H[i1, i2](r, c)
        if (t1 > t2) Hr[(size_t)(t1 * 6 + r) * ld + t2 * 6 + c] += v;
        else Hr[(size_t)(t2 * 6 + c) * ld + t1 * 6 + r] += v;
    }
}
__global__ void k_pg_scatter_sep(int ns, const int* sep, const double* ysep, double* y) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ns * 6) y[(size_t)sep[t / 6] * 6 + t % 6] = ysep[t];
}
__global__ __launch_bounds__(64) void k_pg_seg_backsub(int nseg, const int* segs, const double* rec, double* y) {
    const int sg = blockIdx.x, lane = threadIdx.x & 63;
    if (sg >= nseg) return;
    const int a = segs[2 * sg], b = segs[2 * sg + 1];
    const int k = lane < 6 ? lane : 0;
    double ya[6], yn[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) { ya[q] = y[(size_t)a * 6 + q]; yn[q] = y[(size_t)b * 6 + q]; }
    for (int f = b - 1; f > a; --f) {
        const double* R = rec + (size_t)f * 78 + k * 13;
        double t = R[12];
#pragma unroll
        for (int q = 0; q < 6; ++q) t -= R[q] * yn[q] + R[6 + q] * ya[q];
#pragma unroll
        for (int q = 0; q < 6; ++q) yn[q] = rdl64(t, q);
        if (lane < 6) y[(size_t)f * 6 + lane] = t;
    }
}
}  // namespace liw
extern "C" {
using namespace liw;

struct PgCall {
    liw_ctx* c; const liw_pg_params* pg; int N; double* poses; int n_seq; const int* seq_idx; const double* seq_tf12; int n_loop; const int* loop_idx;
    const double* loop_tf12; int max_iters; liw_summary* summary; double* H_out; double* g_out; double* cost_out;
};
static int pg_run(const PgCall& q);

int liw_posegraph_solve(liw_ctx* c, const liw_pg_params* pg, int N, double* poses, int n_seq, const int* seq_idx, const double* seq_tf12, int n_loop,
                        const int* loop_idx, const double* loop_tf12, int max_iters, liw_summary* summary) {
    return pg_run(PgCall{c, pg, N, poses, n_seq, seq_idx, seq_tf12, n_loop, loop_idx, loop_tf12, max_iters, summary, nullptr, nullptr, nullptr});
}
int liw_posegraph_linearize(liw_ctx* c, const liw_pg_params* pg, int N, const double* poses, int n_seq, const int* seq_idx, const double* seq_tf12, int n_loop,
                            const int* loop_idx, const double* loop_tf12, double* H, double* g, double* cost) {
    return pg_run(PgCall{c, pg, N, const_cast<double*>(poses), n_seq, seq_idx, seq_tf12, n_loop, loop_idx, loop_tf12, 0, nullptr, H, g, cost ? cost : (double*)nullptr});
}

static int pg_run(const PgCall& q) {
    liw_ctx* c = q.c; const liw_pg_params* pg = q.pg; const int N = q.N; double* poses = q.poses; const int n_seq = q.n_seq; const int* seq_idx = q.seq_idx;
    const double* seq_tf12 = q.seq_tf12; const int n_loop = q.n_loop; const int* loop_idx = q.loop_idx; const double* loop_tf12 = q.loop_tf12;
    const int max_iters = q.max_iters; liw_summary* summary = q.summary;
    const bool linearize_only = q.H_out || q.g_out || q.cost_out;
    if (!liw_ctx_has_device(c)) return liw_ctx_fail(c, LIW_ENODEV, "no usable gfx950 device (this library has no CPU fallback)");
    if (!pg || N < 2 || !poses || n_seq < 1 || !seq_idx || !seq_tf12 || n_loop < 0 || (n_loop && (!loop_idx || !loop_tf12)))
        return liw_ctx_fail(c, LIW_EINVAL, "liw_posegraph_solve: bad argument");
    const int E = n_seq + n_loop;
    std::vector<int> eidx((size_t)E * 2);
    std::vector<double> etf((size_t)E * 12), ew(E);
    for (int e = 0; e < E; ++e) {
        const int* id = e < n_seq ? seq_idx + 2 * e : loop_idx + 2 * (e - n_seq);
        if (id[0] < 0 || id[0] >= N || id[1] < 0 || id[1] >= N || id[0] == id[1]) return liw_ctx_fail(c, LIW_EINVAL, "liw_posegraph_solve: edge index out of range");
        eidx[2 * e] = id[0]; eidx[2 * e + 1] = id[1];
        std::memcpy(&etf[(size_t)e * 12], e < n_seq ? seq_tf12 + 12 * (size_t)e : loop_tf12 + 12 * (size_t)(e - n_seq), sizeof(double) * 12);
        ew[e] = e < n_seq ? 1.0 : pg->loop_edge_k;
    }
    const int const_pose = seq_idx[0];   // keyframe_manager.cpp:745-749
    // incidence CSR
    std::vector<int> inc_off(N + 1, 0), inc((size_t)E * 2);
    for (int e = 0; e < E; ++e) { ++inc_off[eidx[2 * e] + 1]; ++inc_off[eidx[2 * e + 1] + 1]; }
    for (int i = 0; i < N; ++i) inc_off[i + 1] += inc_off[i];
    {
        std::vector<int> fill(inc_off.begin(), inc_off.end() - 1);
        for (int e = 0; e < E; ++e) { inc[fill[eidx[2 * e]]++] = e * 2; inc[fill[eidx[2 * e + 1]]++] = e * 2 + 1; }
    }
    // sparse path: the sequential edges must be the chain k -> k+1 (what keyframe_manager builds, :444-453)
    bool sparse = !linearize_only && n_seq == N - 1 && N >= 4 && !std::getenv("LIW_PG_DENSE");
    for (int e = 0; e < n_seq && sparse; ++e) if (eidx[2 * e] != e || eidx[2 * e + 1] != e + 1) sparse = false;
    std::vector<int> sep, sepidx(N, -1), segs;
    if (sparse) {
        std::vector<char> is_sep(N, 0);
        is_sep[0] = is_sep[N - 1] = is_sep[const_pose] = 1;
        for (int e = n_seq; e < E; ++e) { is_sep[eidx[2 * e]] = 1; is_sep[eidx[2 * e + 1]] = 1; }
        for (int i = 0; i < N; i += PG_STRIDE) is_sep[i] = 1;
        for (int i = 0; i < N; ++i) if (is_sep[i]) { sepidx[i] = (int)sep.size(); sep.push_back(i); }
        for (size_t t = 0; t + 1 < sep.size(); ++t) { segs.push_back(sep[t]); segs.push_back(sep[t + 1]); }
    }
    const int ns = (int)sep.size(), nseg = ns > 0 ? ns - 1 : 0, nr = 6 * ns, npr = (nr + 63) / 64 * 64;
    PgNoise noise{};
    for (int k = 0; k < 36; ++k) noise.J[k] = (k % 7 == 0) ? 1.0 : 0.0;   // edge_noise (edge_factor.h:15-25), J(1,2) as written there
    noise.J[0] = 1.0 / pg->loop_sigma_p[0]; noise.J[1 * 6 + 2] = 1.0 / pg->loop_sigma_p[1]; noise.J[2 * 6 + 2] = 1.0 / pg->loop_sigma_p[2];
    noise.J[3 * 6 + 3] = 1.0 / pg->loop_sigma_q[0]; noise.J[4 * 6 + 4] = 1.0 / pg->loop_sigma_q[1]; noise.J[5 * 6 + 5] = 1.0 / pg->loop_sigma_q[2];
    noise.ground_on_p = pg->use_ground_p_factor ? 1.0 : 0.0;
    noise.ground_on_q = pg->use_ground_q_factor ? 1.0 : 0.0;

    (void)hipSetDevice(liw_ctx_device(c));
    hipStream_t s = liw_ctx_stream(c);
    const DevParams P = *liw_ctx_devparams(c);
    const int n = 6 * N, np = (n + 63) / 64 * 64;
    DBuf dx, dxc, deidx, detf, dew, dinc_off, dinc, dYe, dYg, dH, dA, dg, dscale, ddgn, drhs, dpart, dst;
    const int cost_blocks = 64;
    if (dx.alloc(sizeof(double) * n) || dxc.alloc(sizeof(double) * n) || deidx.alloc(sizeof(int) * 2 * (size_t)E) || detf.alloc(sizeof(double) * 12 * (size_t)E) ||
        dew.alloc(sizeof(double) * E) || dinc_off.alloc(sizeof(int) * (N + 1)) || dinc.alloc(sizeof(int) * 2 * (size_t)E) || dYe.alloc(sizeof(double) * 78 * (size_t)E) ||
        dYg.alloc(sizeof(double) * 14 * (size_t)N) || dH.alloc(sparse ? 8 : sizeof(double) * (size_t)np * np) || dA.alloc(sparse ? 8 : sizeof(double) * (size_t)np * np) ||
        dg.alloc(sizeof(double) * np) || dscale.alloc(sizeof(double) * np) || ddgn.alloc(sizeof(double) * np) || drhs.alloc(sizeof(double) * np) ||
        dpart.alloc(sizeof(double) * cost_blocks) || dst.alloc(sizeof(int) * (2 + 2 * (size_t)(np / 64))))
        return liw_ctx_fail(c, LIW_ENOMEM, "hipMalloc");
    DBuf dDd, dgd, dOs, dLb, dsep, dsepidx, dsegs, dSred, drec, dHr, drhsr, dlidx;
    if (sparse && (dDd.alloc(sizeof(double) * 36 * (size_t)N) || dgd.alloc(sizeof(double) * 6 * (size_t)N) || dOs.alloc(sizeof(double) * 36 * (size_t)N) ||
                   dLb.alloc(sizeof(double) * 36 * (size_t)std::max(n_loop, 1)) || dsep.alloc(sizeof(int) * ns) || dsepidx.alloc(sizeof(int) * N) ||
                   dsegs.alloc(sizeof(int) * 2 * (size_t)std::max(nseg, 1)) || dSred.alloc(sizeof(double) * PG_SEG * (size_t)std::max(nseg, 1)) ||
                   drec.alloc(sizeof(double) * 78 * (size_t)N) || dHr.alloc(sizeof(double) * (size_t)npr * npr) || drhsr.alloc(sizeof(double) * npr) ||
                   dlidx.alloc(sizeof(int) * 2 * (size_t)std::max(n_loop, 1))))
        return liw_ctx_fail(c, LIW_ENOMEM, "hipMalloc");
    auto up = [&](DBuf& d, const void* src, size_t bytes) { (void)hipMemcpyAsync(d.p, src, bytes, hipMemcpyHostToDevice, s); };
    up(deidx, eidx.data(), sizeof(int) * eidx.size()); up(detf, etf.data(), sizeof(double) * etf.size()); up(dew, ew.data(), sizeof(double) * E);
    up(dinc_off, inc_off.data(), sizeof(int) * (N + 1)); up(dinc, inc.data(), sizeof(int) * inc.size());
    std::vector<double> Ddh;
    if (sparse) {
        up(dsep, sep.data(), sizeof(int) * ns); up(dsepidx, sepidx.data(), sizeof(int) * N); up(dsegs, segs.data(), sizeof(int) * segs.size());
        if (n_loop) up(dlidx, eidx.data() + 2 * (size_t)n_seq, sizeof(int) * 2 * (size_t)n_loop);
        (void)hipMemsetAsync(dOs.p, 0, sizeof(double) * 36 * (size_t)N, s);
        Ddh.resize((size_t)N * 36);
    }

    std::vector<double> x(poses, poses + n), cand(n), g(n), scale(n, 1.0), dgn(n, 0.0), gs(n), y(np), Hdiag(n), part(cost_blocks);
    const unsigned lin_blocks = (unsigned)((E + 3) / 4 + (N + 7) / 8);
    auto evaluate = [&](const std::vector<double>& xv, DBuf& dxx, bool with_jac, double* cost) -> int {
        up(dxx, xv.data(), sizeof(double) * n);
        hipLaunchKernelGGL(k_pg_linearize, dim3(lin_blocks), dim3(64), 0, s, N, E, dxx.as<double>(), deidx.as<int>(), detf.as<double>(), dew.as<double>(), noise, P,
                           dYe.as<double>(), dYg.as<double>(), with_jac ? 1 : 0);
        hipLaunchKernelGGL(k_pg_cost, dim3(cost_blocks), dim3(256), 0, s, N, E, dYe.as<double>(), dYg.as<double>(), const_pose, dpart.as<double>());
        if (with_jac && sparse) {
            hipLaunchKernelGGL(k_pg_assemble_blocks, dim3(N), dim3(64), 0, s, N, n_seq, dinc_off.as<int>(), dinc.as<int>(), deidx.as<int>(), dYe.as<double>(),
                               dYg.as<double>(), const_pose, dDd.as<double>(), dgd.as<double>(), dOs.as<double>(), dLb.as<double>());
            (void)hipMemcpyAsync(g.data(), dgd.p, sizeof(double) * n, hipMemcpyDeviceToHost, s);
            (void)hipMemcpyAsync(Ddh.data(), dDd.p, sizeof(double) * 36 * (size_t)N, hipMemcpyDeviceToHost, s);
        } else if (with_jac) {
            (void)hipMemsetAsync(dH.p, 0, sizeof(double) * (size_t)np * np, s);
            hipLaunchKernelGGL(k_pg_assemble, dim3(N), dim3(64), 0, s, N, np, dinc_off.as<int>(), dinc.as<int>(), deidx.as<int>(), dYe.as<double>(), dYg.as<double>(),
                               const_pose, dH.as<double>(), dg.as<double>());
            (void)hipMemcpyAsync(g.data(), dg.p, sizeof(double) * n, hipMemcpyDeviceToHost, s);
            hipLaunchKernelGGL(k_pg_diag, dim3((n + 255) / 256), dim3(256), 0, s, n, np, dH.as<double>(), drhs.as<double>());
            (void)hipMemcpyAsync(Hdiag.data(), drhs.p, sizeof(double) * n, hipMemcpyDeviceToHost, s);
        }
        (void)hipMemcpyAsync(part.data(), dpart.p, sizeof(double) * cost_blocks, hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) return liw_ctx_fail(c, LIW_EHIP, "pose graph evaluate");
        if (with_jac && sparse)
            for (int i = 0; i < n; ++i) Hdiag[i] = Ddh[(size_t)(i / 6) * 36 + (i % 6) * 7];
        double t = 0.0;
        for (double v : part) t += v;
        *cost = 0.5 * t;
        return LIW_OK;
    };
    auto is_const = [&](int i) { return i / 6 == const_pose; };
    auto plus = [&](const std::vector<double>& xin, const std::vector<double>& delta, std::vector<double>& xout) {
        xout = xin;
        for (int i = 0; i < N; ++i) {
            if (i == const_pose) continue;
            for (int k = 0; k < 3; ++k) xout[i * 6 + k] = xin[i * 6 + k] + delta[i * 6 + k];
            so3_plus_host(&xin[i * 6 + 3], &delta[i * 6 + 3], &xout[i * 6 + 3]);
        }
    };
    auto gradient_max_norm = [&]() {
        std::vector<double> ng(n), xp;
        for (int i = 0; i < n; ++i) ng[i] = -g[i];
        plus(x, ng, xp);
        double m = 0.0;
        for (int i = 0; i < n; ++i) if (!is_const(i)) m = std::max(m, std::fabs(x[i] - xp[i]));
        return m;
    };

    // ---- Ceres trust-region Levenberg-Marquardt (defaults)
    const int K = max_iters > 0 ? max_iters : 50;
    const double kMinDiagH = 1e-6, kMaxDiagH = 1e32, kMinRelDecH = 1e-3, kFuncTolH = 1e-6, kGradTolH = 1e-10, kParamTolH = 1e-8;
    double radius = 1e4, decrease_factor = 2.0, x_cost = 0.0;
    bool reuse_diagonal = false;
    liw_summary sum{};
    if (int r = evaluate(x, dx, true, &x_cost)) return r;
    if (linearize_only) {   // dense tangent-space normal equations (lower triangle mirrored), constant key frame = identity block
        if (q.cost_out) *q.cost_out = x_cost;
        if (q.g_out) std::memcpy(q.g_out, g.data(), sizeof(double) * n);
        if (q.H_out) {
            std::vector<double> Hp((size_t)np * np);
            if (hipMemcpy(Hp.data(), dH.p, sizeof(double) * (size_t)np * np, hipMemcpyDeviceToHost) != hipSuccess) return liw_ctx_fail(c, LIW_EHIP, "hipMemcpy");
            for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) q.H_out[(size_t)i * n + j] = q.H_out[(size_t)j * n + i] = Hp[(size_t)i * np + j];
        }
        return LIW_OK;
    }
    sum.initial_cost = x_cost;
    for (int i = 0; i < n; ++i) scale[i] = is_const(i) ? 1.0 : 1.0 / (1.0 + std::sqrt(Hdiag[i]));
    double x_norm = 0.0;
    for (int i = 0; i < n; ++i) if (!is_const(i)) x_norm += x[i] * x[i];
    x_norm = std::sqrt(x_norm);
    double gmax = gradient_max_norm();
    int iteration = 0, invalid_steps = 0, termination = 0, successful = 0;
    bool last_successful = true;
    if (gmax <= kGradTolH) termination = 1;
    while (!termination) {
        if (iteration >= K) { termination = 4; break; }
        if (iteration > 0 && last_successful && gmax <= kGradTolH) { termination = 1; break; }
        if (!(radius > 1e-32)) { termination = 5; break; }
        ++iteration;
        for (int i = 0; i < n; ++i) gs[i] = is_const(i) ? 0.0 : g[i] * scale[i];
        if (!reuse_diagonal)
            for (int i = 0; i < n; ++i) dgn[i] = is_const(i) ? 0.0 : std::min(std::max(Hdiag[i] * scale[i] * scale[i], kMinDiagH), kMaxDiagH);
        reuse_diagonal = true;
        up(dscale, scale.data(), sizeof(double) * n); up(ddgn, dgn.data(), sizeof(double) * n);
        std::fill(y.begin(), y.end(), 0.0);
        std::memcpy(y.data(), gs.data(), sizeof(double) * n);
        up(drhs, y.data(), sizeof(double) * np);
        int st = 0, st2 = 0;
        if (sparse) {
            int* seg_status = dst.as<int>() + 1 + 2 * (np / 64);
            (void)hipMemsetAsync(seg_status, 0, sizeof(int), s);
            (void)hipMemsetAsync(dHr.p, 0, sizeof(double) * (size_t)npr * npr, s);
            (void)hipMemsetAsync(drhsr.p, 0, sizeof(double) * npr, s);
            hipLaunchKernelGGL(k_pg_seg_elim, dim3(nseg), dim3(64), 0, s, nseg, dsegs.as<int>(), dDd.as<double>(), dgd.as<double>(), dOs.as<double>(), dscale.as<double>(),
                               ddgn.as<double>(), radius, dSred.as<double>(), drec.as<double>(), seg_status);
            hipLaunchKernelGGL(k_pg_reduced, dim3(ns + 1), dim3(64), 0, s, ns, nr, npr, dsep.as<int>(), dDd.as<double>(), dgd.as<double>(), dOs.as<double>(),
                               dscale.as<double>(), ddgn.as<double>(), radius, const_pose, dSred.as<double>(), dHr.as<double>(), drhsr.as<double>());
            if (n_loop) hipLaunchKernelGGL(k_pg_reduced_loops, dim3(1), dim3(64), 0, s, n_loop, dlidx.as<int>(), dsepidx.as<int>(), dLb.as<double>(), dscale.as<double>(),
                                           const_pose, npr, dHr.as<double>());
            if (int r = dense_cholesky_solve(c, dHr.as<double>(), npr, drhsr.as<double>(), dst.as<int>(), s)) return r;
            hipLaunchKernelGGL(k_pg_scatter_sep, dim3((nr + 255) / 256), dim3(256), 0, s, ns, dsep.as<int>(), drhsr.as<double>(), drhs.as<double>());
            hipLaunchKernelGGL(k_pg_seg_backsub, dim3(nseg), dim3(64), 0, s, nseg, dsegs.as<int>(), drec.as<double>(), drhs.as<double>());
            (void)hipMemcpyAsync(&st2, seg_status, sizeof(int), hipMemcpyDeviceToHost, s);
        } else {
            hipLaunchKernelGGL(k_pg_scale_damp, dim3((np + 255) / 256, np), dim3(256), 0, s, n, np, np, dH.as<double>(), dscale.as<double>(), ddgn.as<double>(), radius,
                               dA.as<double>());
            if (int r = dense_cholesky_solve(c, dA.as<double>(), np, drhs.as<double>(), dst.as<int>(), s)) return r;
        }
        (void)hipMemcpyAsync(y.data(), drhs.p, sizeof(double) * np, hipMemcpyDeviceToHost, s);
        (void)hipMemcpyAsync(&st, dst.p, sizeof(int), hipMemcpyDeviceToHost, s);
        if (hipStreamSynchronize(s) != hipSuccess) return liw_ctx_fail(c, LIW_EHIP, "pose graph solve");
        bool solved = st == 0 && st2 == 0;
        for (int i = 0; i < n && solved; ++i) if (!std::isfinite(y[i])) solved = false;
        // model cost change with (A + D^2) y = g_s, step = -y:  (y'g_s + y'D^2 y) / 2
        double model_cost_change = 0.0;
        bool valid = false;
        if (solved) {
            double ytg = 0.0, dsum = 0.0;
            for (int i = 0; i < n; ++i) if (!is_const(i)) { ytg += y[i] * gs[i]; dsum += dgn[i] / radius * y[i] * y[i]; }
            model_cost_change = 0.5 * (ytg + dsum);
            valid = model_cost_change > 0.0 && std::isfinite(model_cost_change);
        }
        if (!valid) {
            if (++invalid_steps >= 5) { termination = 6; --iteration; break; }
            radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            last_successful = false;
            continue;
        }
        invalid_steps = 0;
        std::vector<double> delta(n, 0.0);
        for (int i = 0; i < n; ++i) if (!is_const(i)) delta[i] = -y[i] * scale[i];
        plus(x, delta, cand);
        double candidate_cost = 0.0;
        if (int r = evaluate(cand, dxc, false, &candidate_cost)) return r;
        if (!std::isfinite(candidate_cost)) candidate_cost = 1.7976931348623157e308;
        double step_norm = 0.0;
        for (int i = 0; i < n; ++i) if (!is_const(i)) step_norm += (x[i] - cand[i]) * (x[i] - cand[i]);
        step_norm = std::sqrt(step_norm);
        if (step_norm <= kParamTolH * (x_norm + kParamTolH)) { termination = 3; break; }
        if (std::fabs(x_cost - candidate_cost) <= kFuncTolH * x_cost) { termination = 2; break; }
        const double rho = (x_cost - candidate_cost) / model_cost_change;
        if (rho > kMinRelDecH) {
            x = cand;
            if (int r = evaluate(x, dx, true, &x_cost)) return r;
            x_norm = 0.0;
            for (int i = 0; i < n; ++i) if (!is_const(i)) x_norm += x[i] * x[i];
            x_norm = std::sqrt(x_norm);
            gmax = gradient_max_norm();
            radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3.0)));
            decrease_factor = 2.0; reuse_diagonal = false;
            ++successful;
            last_successful = true;
        } else {
            radius /= decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
            last_successful = false;
        }
    }
    std::memcpy(poses, x.data(), sizeof(double) * n);
    sum.iterations = iteration; sum.successful_steps = successful; sum.termination = termination; sum.final_cost = x_cost;
    if (summary) *summary = sum;
    return LIW_OK;
}

}  // extern "C"
