// smallmat.hpp — l x l (l <= 128) dense symmetric algebra on the host, f64, row-major.
// Used by the randomized-PCA driver for the Rayleigh–Ritz step (eigen-decomposition of the
// l x l projected matrix) and the CholeskyQR orthonormalisation of the k x l block.  These
// are O(l^3) = ~1e6 flop problems: not worth a kernel launch, far from the hot path.
// Pure C++ (no HIP) so the CPU test-suite can exercise it.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

namespace srx {
namespace smallmat {

// Householder reduction of the symmetric matrix held in V (n x n, row-major) to tridiagonal
// form; on return V holds the accumulated orthogonal transform, d the diagonal, e the
// sub-diagonal (e[0] = 0).  Classic EISPACK tred2 scheme.
inline void tridiagonalize(int n, double* V, double* d, double* e) {
    // A(i, j) is stored at V[j*n + i] (column-major): every inner loop below runs over the FIRST
    // index, i.e. over contiguous memory, and vectorises
    auto A = [&](int i, int j) -> double& { return V[(size_t)j * n + i]; };
    for (int j = 0; j < n; ++j) d[j] = A(n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) {
                d[j] = A(i - 1, j);
                A(i, j) = 0.0;
                A(j, i) = 0.0;
            }
        } else {
            for (int k = 0; k < i; ++k) {
                d[k] /= scale;
                h += d[k] * d[k];
            }
            double f = d[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                A(j, i) = f;
                g = e[j] + A(j, j) * f;
                for (int k = j + 1; k <= i - 1; ++k) {
                    g += A(k, j) * d[k];
                    e[k] += A(k, j) * f;
                }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) {
                e[j] /= h;
                f += e[j] * d[j];
            }
            double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j];
                g = e[j];
                for (int k = j; k <= i - 1; ++k) A(k, j) -= (f * e[k] + g * d[k]);
                d[j] = A(i - 1, j);
                A(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) {
        A(n - 1, i) = A(i, i);
        A(i, i) = 1.0;
        double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = A(k, i + 1) / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += A(k, i + 1) * A(k, j);
                for (int k = 0; k <= i; ++k) A(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) A(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) {
        d[j] = A(n - 1, j);
        A(n - 1, j) = 0.0;
    }
    A(n - 1, n - 1) = 1.0;
    e[0] = 0.0;
}

// Implicit-shift QL on the tridiagonal (d, e), rotating the columns of V along.
// Returns false if an eigenvalue fails to converge in 60 sweeps.
inline bool tridiagonal_ql(int n, double* V, double* d, double* e) {
    auto A = [&](int i, int j) -> double& { return V[(size_t)j * n + i]; };   // column-major, as above
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = 2.220446049250313e-16;
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) {
            if (std::fabs(e[m]) <= eps * tst1) break;
            ++m;
        }
        if (m >= n) m = n - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 60) return false;
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c;
                double el1 = e[l + 1];
                double s = 0.0, s2 = 0.0;
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2;
                    c2 = c;
                    s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = std::hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; ++k) {
                        h = A(k, i + 1);
                        A(k, i + 1) = s * A(k, i) + c * h;
                        A(k, i) = c * A(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        d[l] += f;
        e[l] = 0.0;
    }
    return true;
}

// Eigen-decomposition of a symmetric n x n matrix (row-major, only read).  On return
// evals is DESCENDING and evecs (n x n row-major) holds eigenvector i in COLUMN i.
inline bool sym_eig_desc(int n, const double* Ain, double* evals, double* evecs) {
    std::vector<double> V(Ain, Ain + (size_t)n * n), d(n), e(n);
    // symmetrise against round-off in the input
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            double s = 0.5 * (V[(size_t)i * n + j] + V[(size_t)j * n + i]);
            V[(size_t)i * n + j] = s;
            V[(size_t)j * n + i] = s;
        }
    tridiagonalize(n, V.data(), d.data(), e.data());
    if (!tridiagonal_ql(n, V.data(), d.data(), e.data())) return false;
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] > d[b]; });
    for (int c = 0; c < n; ++c) {
        evals[c] = d[order[c]];
        for (int r = 0; r < n; ++r) evecs[(size_t)r * n + c] = V[(size_t)order[c] * n + r];
    }
    return true;
}

// Upper Cholesky factor R (G = R^T R) of the leading n x n block of a symmetric positive
// definite matrix stored with leading dimension ld; returns false when a pivot is not
// positive.  Rinv receives R^{-1} (upper triangular, ld x ld, zero elsewhere).
inline bool chol_upper_inverse(int n, int ld, const double* G, double* Rinv) {
    std::vector<double> R((size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double s = G[(size_t)j * ld + j];
        for (int k = 0; k < j; ++k) s -= R[(size_t)k * n + j] * R[(size_t)k * n + j];
        if (!(s > 0.0)) return false;
        double rjj = std::sqrt(s);
        R[(size_t)j * n + j] = rjj;
        for (int i = j + 1; i < n; ++i) {
            double t = G[(size_t)j * ld + i];
            for (int k = 0; k < j; ++k) t -= R[(size_t)k * n + j] * R[(size_t)k * n + i];
            R[(size_t)j * n + i] = t / rjj;
        }
    }
    for (size_t t = 0; t < (size_t)ld * ld; ++t) Rinv[t] = 0.0;
    // back-substitution column by column: R * X = I
    for (int c = 0; c < n; ++c) {
        for (int i = c; i >= 0; --i) {
            double s = (i == c) ? 1.0 : 0.0;
            for (int k = i + 1; k <= c; ++k) s -= R[(size_t)i * n + k] * Rinv[(size_t)k * ld + c];
            Rinv[(size_t)i * ld + c] = s / R[(size_t)i * n + i];
        }
    }
    return true;
}

}  // namespace smallmat
}  // namespace srx
