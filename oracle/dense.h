// ORACLE — TEST INFRASTRUCTURE ONLY (see jet.h header).  PARITY UNPINNED.
//
// Dense linear algebra the reference delegates to Eigen 3.3.x (not vendored): MatrixXd products
// (src/factor/solver.cpp:12-13,36-37), MatrixXd::inverse() = partial-pivot LU (solver.cpp:21,
// imu_preintegraption.h:149), LLT (imu_preintegraption.h:149, wheel_odom_preintegration.h:123),
// SelfAdjointEigenSolver (solver.cpp:392).  Restated from their textbook definitions; row-major
// storage.  The eigen-solver is a cyclic Jacobi sweep (Eigen uses tridiagonal QL; eigen-vector
// signs are arbitrary in both, so only sign-invariant quantities are compared — DESIGN.md).
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

namespace oracle {

struct DMat {
    int rows = 0, cols = 0;
    std::vector<double> d;  // row-major
    DMat() {}
    DMat(int r, int c) : rows(r), cols(c), d(size_t(r) * c, 0.0) {}
    double& operator()(int i, int j) { return d[size_t(i) * cols + j]; }
    const double& operator()(int i, int j) const { return d[size_t(i) * cols + j]; }
    double* row(int i) { return d.data() + size_t(i) * cols; }
    const double* row(int i) const { return d.data() + size_t(i) * cols; }
};

inline int& dense_threads() { static int t = 1; return t; }   // OpenMP team of the dense kernels (oracle_set_threads; 1 = the reference's serial Eigen)

// H = J^T J (all of it, like Eigen's general product — the reference does not exploit symmetry),
// g = -J^T R.   J is rows x cols row-major.  Blocked over rows so that the update of a
// (cols x cols) panel streams J once; inner loops are written for the auto-vectoriser.
inline void gemm_JtJ(const DMat& J, DMat& H) {
    const int m = J.rows, n = J.cols;
    H = DMat(n, n);
    constexpr int RB = 4;  // rows of J consumed per sweep
    int r = 0;
    for (; r + RB <= m; r += RB) {
        const double* j0 = J.row(r);
        const double* j1 = J.row(r + 1);
        const double* j2 = J.row(r + 2);
        const double* j3 = J.row(r + 3);
        for (int a = 0; a < n; ++a) {
            const double a0 = j0[a], a1 = j1[a], a2 = j2[a], a3 = j3[a];
            if (a0 == 0.0 && a1 == 0.0 && a2 == 0.0 && a3 == 0.0) continue;  // exact zeros add exactly 0
            double* h = H.row(a);
            for (int b = 0; b < n; ++b) h[b] += a0 * j0[b] + a1 * j1[b] + a2 * j2[b] + a3 * j3[b];
        }
    }
    for (; r < m; ++r) {
        const double* j0 = J.row(r);
        for (int a = 0; a < n; ++a) {
            const double a0 = j0[a];
            if (a0 == 0.0) continue;
            double* h = H.row(a);
            for (int b = 0; b < n; ++b) h[b] += a0 * j0[b];
        }
    }
}
// The same product with NO zero skipping: the arithmetic the reference's dense Eigen GEMM
// performs (2*rows*cols^2 flops).  Used by the timed cpu_baseline so the comparator is not
// flattered by sparsity tricks the reference does not have.
inline void gemm_JtJ_dense(const DMat& J, DMat& H) {
    const int m = J.rows, n = J.cols;
    H = DMat(n, n);
    constexpr int RB = 8;
    if (dense_threads() > 1) {   // every thread owns a band of rows of H and streams J once: same sums in the same order per entry
        const int T = dense_threads();
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) num_threads(T)
#endif
        for (int band = 0; band < T; ++band) {
            const int a0 = int((long)n * band / T), a1 = int((long)n * (band + 1) / T);
            int r = 0;
            for (; r + RB <= m; r += RB) {
                const double* j[RB];
                for (int k = 0; k < RB; ++k) j[k] = J.row(r + k);
                for (int a = a0; a < a1; ++a) {
                    double* h = H.row(a);
                    double c[RB];
                    for (int k = 0; k < RB; ++k) c[k] = j[k][a];
                    for (int b = 0; b < n; ++b) {
                        double s = h[b];
                        for (int k = 0; k < RB; ++k) s += c[k] * j[k][b];
                        h[b] = s;
                    }
                }
            }
            for (; r < m; ++r) {
                const double* j0 = J.row(r);
                for (int a = a0; a < a1; ++a) {
                    double* h = H.row(a);
                    const double av = j0[a];
                    for (int b = 0; b < n; ++b) h[b] += av * j0[b];
                }
            }
        }
        return;
    }
    int r = 0;
    for (; r + RB <= m; r += RB) {
        const double* j[RB];
        for (int k = 0; k < RB; ++k) j[k] = J.row(r + k);
        for (int a = 0; a < n; ++a) {
            double* h = H.row(a);
            double c[RB];
            for (int k = 0; k < RB; ++k) c[k] = j[k][a];
            for (int b = 0; b < n; ++b) {
                double s = h[b];
                for (int k = 0; k < RB; ++k) s += c[k] * j[k][b];
                h[b] = s;
            }
        }
    }
    for (; r < m; ++r) {
        const double* j0 = J.row(r);
        for (int a = 0; a < n; ++a) {
            double* h = H.row(a);
            const double a0 = j0[a];
            for (int b = 0; b < n; ++b) h[b] += a0 * j0[b];
        }
    }
}
inline void gemv_mJtR(const DMat& J, const std::vector<double>& R, std::vector<double>& g) {
    const int m = J.rows, n = J.cols;
    g.assign(n, 0.0);
    for (int r = 0; r < m; ++r) {
        const double* j0 = J.row(r);
        const double rr = R[r];
        for (int a = 0; a < n; ++a) g[a] += j0[a] * rr;
    }
    for (int a = 0; a < n; ++a) g[a] = -g[a];
}
inline DMat matmul(const DMat& A, const DMat& B) {
    DMat C(A.rows, B.cols);
    for (int i = 0; i < A.rows; ++i)
        for (int k = 0; k < A.cols; ++k) {
            const double a = A(i, k);
            const double* b = B.row(k);
            double* c = C.row(i);
            for (int j = 0; j < B.cols; ++j) c[j] += a * b[j];
        }
    return C;
}

// inverse by LU with partial pivoting (Eigen PartialPivLU::inverse()).  Returns false on a zero pivot.
inline bool lu_inverse(const DMat& A, DMat& Ainv) {
    const int n = A.rows;
    DMat lu = A;
    std::vector<int> perm(n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int k = 0; k < n; ++k) {
        int piv = k;
        double best = std::fabs(lu(k, k));
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(lu(i, k)) > best) { best = std::fabs(lu(i, k)); piv = i; }
        if (best == 0.0) return false;
        if (piv != k) {
            for (int j = 0; j < n; ++j) std::swap(lu(k, j), lu(piv, j));
            std::swap(perm[k], perm[piv]);
        }
        const double inv = 1.0 / lu(k, k);
        for (int i = k + 1; i < n; ++i) {
            const double f = lu(i, k) * inv;
            lu(i, k) = f;
            if (f == 0.0) continue;
            double* ri = lu.row(i);
            const double* rk = lu.row(k);
            for (int j = k + 1; j < n; ++j) ri[j] -= f * rk[j];
        }
    }
    // solve LU X = P I, all right-hand sides at once (row-major X)
    Ainv = DMat(n, n);
    for (int i = 0; i < n; ++i) Ainv(i, perm[i]) = 1.0;
    for (int i = 0; i < n; ++i) {           // forward, unit lower
        double* xi = Ainv.row(i);
        for (int k = 0; k < i; ++k) {
            const double f = lu(i, k);
            if (f == 0.0) continue;
            const double* xk = Ainv.row(k);
            for (int j = 0; j < n; ++j) xi[j] -= f * xk[j];
        }
    }
    for (int i = n - 1; i >= 0; --i) {      // backward
        double* xi = Ainv.row(i);
        for (int k = i + 1; k < n; ++k) {
            const double f = lu(i, k);
            if (f == 0.0) continue;
            const double* xk = Ainv.row(k);
            for (int j = 0; j < n; ++j) xi[j] -= f * xk[j];
        }
        const double inv = 1.0 / lu(i, i);
        for (int j = 0; j < n; ++j) xi[j] *= inv;
    }
    return true;
}

// A = L L^T (lower).  Returns false if a pivot is <= 0.
inline bool llt_lower(const DMat& A, DMat& L) {
    const int n = A.rows;
    L = DMat(n, n);
    for (int j = 0; j < n; ++j) {
        double s = A(j, j);
        for (int k = 0; k < j; ++k) s -= L(j, k) * L(j, k);
        if (!(s > 0.0)) return false;
        const double d = std::sqrt(s);
        L(j, j) = d;
        const int T = dense_threads();   // (rows below the pivot are independent dot products: same sums, same order, any team size)
#if defined(_OPENMP)
#pragma omp parallel for schedule(static) num_threads(T) if (T > 1 && n - j > 64)
#endif
        for (int i = j + 1; i < n; ++i) {
            double t = A(i, j);
            const double* li = L.row(i);
            const double* lj = L.row(j);
            for (int k = 0; k < j; ++k) t -= li[k] * lj[k];
            L(i, j) = t / d;
        }
    }
    return true;
}
// solve A x = b given A = L L^T
inline void llt_solve(const DMat& L, std::vector<double>& b) {
    const int n = L.rows;
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        const double* li = L.row(i);
        for (int k = 0; k < i; ++k) s -= li[k] * b[k];
        b[i] = s / li[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L(k, i) * b[k];
        b[i] = s / L(i, i);
    }
}

// symmetric eigen-decomposition, cyclic Jacobi.  A = V diag(w) V^T, w ascending, V columns.
// Sign convention: each eigen-vector's largest-magnitude component is made positive.
inline void jacobi_eigh(const DMat& Ain, std::vector<double>& w, DMat& V) {
    const int n = Ain.rows;
    DMat A = Ain;
    V = DMat(n, n);
    for (int i = 0; i < n; ++i) V(i, i) = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < n; ++i) {
            diag += A(i, i) * A(i, i);
            for (int j = i + 1; j < n; ++j) off += A(i, j) * A(i, j);
        }
        if (off <= 1e-60 * diag || off == 0.0) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A(p, q);
                if (apq == 0.0) continue;
                const double app = A(p, p), aqq = A(q, q);
                const double tau = (aqq - app) / (2.0 * apq);
                const double t = (tau >= 0.0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
                const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = A(k, p), akq = A(k, q);
                    A(k, p) = c * akp - s * akq;
                    A(k, q) = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = A(p, k), aqk = A(q, k);
                    A(p, k) = c * apk - s * aqk;
                    A(q, k) = s * apk + c * aqk;
                }
                for (int k = 0; k < n; ++k) {
                    const double vkp = V(k, p), vkq = V(k, q);
                    V(k, p) = c * vkp - s * vkq;
                    V(k, q) = s * vkp + c * vkq;
                }
            }
    }
    w.resize(n);
    for (int i = 0; i < n; ++i) w[i] = A(i, i);
    // sort ascending (selection sort, n is 15)
    for (int i = 0; i < n; ++i) {
        int m = i;
        for (int j = i + 1; j < n; ++j) if (w[j] < w[m]) m = j;
        if (m != i) {
            std::swap(w[i], w[m]);
            for (int k = 0; k < n; ++k) std::swap(V(k, i), V(k, m));
        }
    }
    for (int j = 0; j < n; ++j) {
        int m = 0;
        for (int k = 1; k < n; ++k) if (std::fabs(V(k, j)) > std::fabs(V(m, j))) m = k;
        if (V(m, j) < 0.0) for (int k = 0; k < n; ++k) V(k, j) = -V(k, j);
    }
}

}  // namespace oracle
