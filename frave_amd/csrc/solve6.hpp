// solve6.hpp -- the 6 x 6 solves behind the context-model fit, ONE source for the host (fri_hip_solve6, fri_hip_fit_*_params) and for the
// device (fit_solve_kernel, k4_fit.hip): the same sequence of IEEE f64 operations on both sides (+, -, *, / and, on the rare route,
// sqrt; the library is built with -ffp-contract=off), so the device-side solve of the asynchronous encode chain returns bit for bit
// the parameters the host functions return for the same sums (tests/test_gpu_fit.py).
//
// The reference fits with an SVD least squares over n x 6 f32 design matrices (lstsq, context_modeling.rs:144-202, third-party
// arithmetic: parity unpinned). Here: normal equations M x = y from exact integer sums.
//   * M safely positive definite (every pivot of the LDL^T factorisation above 1e-8 of the largest diagonal entry - any image with
//     texture in the layer group): one solution, found without a square root and with six divisions (the pivots' reciprocals) - on the
//     device this runs on one lane between two kernels of the chain, where a dependent f64 division costs ~0.1 us.
//   * otherwise (rank deficient or nearly so: flat regions, a feature that is zero everywhere): the minimum-norm solution through a cyclic
//     Jacobi eigen-decomposition with lstsq's relative cut-off - what the SVD returns, up to rounding.
#pragma once
#include <cmath>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define FRI_HD __host__ __device__
#else
#define FRI_HD
#endif

namespace fri {

FRI_HD inline void solve6(const double (&m)[6][6], const double (&y)[6], double (&x)[6]) {
    {
        // M = L D L^T, L unit lower triangular. inv[j] = 1 / D[j].
        double l[6][6], d[6], inv[6], dmax = 0.0;
        for (int i = 0; i < 6; i++) dmax = m[i][i] > dmax ? m[i][i] : dmax;
        bool ok = dmax > 0.0 && dmax < 1.0e300; // (a NaN or an infinity fails the comparison or the bound)
        for (int j = 0; j < 6 && ok; j++) {
            double dj = m[j][j];
            for (int k = 0; k < j; k++) dj -= l[j][k] * l[j][k] * d[k];
            if (!(dj > 1e-8 * dmax)) {
                ok = false;
                break;
            }
            d[j] = dj;
            inv[j] = 1.0 / dj;
            for (int i = j + 1; i < 6; i++) {
                double t = m[i][j];
                for (int k = 0; k < j; k++) t -= l[i][k] * l[j][k] * d[k];
                l[i][j] = t * inv[j];
            }
        }
        if (ok) {
            double z[6];
            for (int i = 0; i < 6; i++) { // L z = y
                double t = y[i];
                for (int k = 0; k < i; k++) t -= l[i][k] * z[k];
                z[i] = t;
            }
            for (int i = 5; i >= 0; i--) { // D L^T x = z
                double t = z[i] * inv[i];
                for (int k = i + 1; k < 6; k++) t -= l[k][i] * x[k];
                x[i] = t;
            }
            return;
        }
    }
    // cyclic Jacobi: a = V diag(lam) V^T; x = sum over the eigen-directions above the cut-off of v (v . y) / lam
    double a[6][6], v[6][6];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            a[i][j] = m[i][j];
            v[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (int i = 0; i < 6; i++)
            for (int j = i + 1; j < 6; j++) off += a[i][j] * a[i][j];
        if (!(off >= 1e-300)) break; // converged (or not a number: nothing to iterate on)
        for (int pp = 0; pp < 6; pp++)
            for (int q = pp + 1; q < 6; q++) {
                const double apq = a[pp][q];
                if ((apq < 0 ? -apq : apq) < 1e-300) continue;
                const double theta = (a[q][q] - a[pp][pp]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / ((theta < 0 ? -theta : theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 6; k++) {
                    const double akp = a[k][pp], akq = a[k][q];
                    a[k][pp] = c * akp - sn * akq;
                    a[k][q] = sn * akp + c * akq;
                }
                for (int k = 0; k < 6; k++) {
                    const double apk = a[pp][k], aqk = a[q][k];
                    a[pp][k] = c * apk - sn * aqk;
                    a[q][k] = sn * apk + c * aqk;
                }
                for (int k = 0; k < 6; k++) {
                    const double vkp = v[k][pp], vkq = v[k][q];
                    v[k][pp] = c * vkp - sn * vkq;
                    v[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    double lmax = 0.0;
    for (int i = 0; i < 6; i++) lmax = a[i][i] > lmax ? a[i][i] : lmax;
    for (int k = 0; k < 6; k++) x[k] = 0.0;
    for (int i = 0; i < 6; i++) {
        if (!(a[i][i] > 1e-12 * lmax)) continue; // rank-deficient direction: the minimum-norm solution leaves it at 0
        double proj = 0.0;
        for (int k = 0; k < 6; k++) proj += v[k][i] * y[k];
        for (int k = 0; k < 6; k++) x[k] += v[k][i] * proj / a[i][i];
    }
}

FRI_HD inline int tri_index(int i, int j, int n) { // index into the upper triangle (row major)
    if (i > j) {
        const int t = i;
        i = j;
        j = t;
    }
    return i * n - i * (i - 1) / 2 + (j - i);
}

// optimize_value_prediction (context_modeling.rs:175-202) from one layer group's Gram sums gram[28] (upper triangle of sum u u^T,
// u = [v0..v5, value]): A^T A = rows / columns 0..5, A^T b = column 6.
FRI_HD inline void fit_value_group(const long long *gram, float *out /* [6] */) {
    double m[6][6], y[6], x[6];
    for (int i = 0; i < 6; i++) {
        y[i] = (double)gram[tri_index(i, 6, 7)];
        for (int j = 0; j < 6; j++) m[i][j] = (double)gram[tri_index(i, j, 7)];
    }
    solve6(m, y, x);
    for (int k = 0; k < 6; k++) out[k] = (float)x[k];
}

// optimize_width_prediction (context_modeling.rs:144-173) from wtw[21], wtr[6] over the Some rows; `rows` = height of the reference's
// matrix: its all-zero rows carry the constant feature 1 with residual 0.
FRI_HD inline void fit_width_group(const long long *wtw, const double *wtr, unsigned long long rows, float *out /* [6] */) {
    double m[6][6], y[6], x[6];
    for (int i = 0; i < 6; i++) {
        y[i] = wtr[i];
        for (int j = 0; j < 6; j++) m[i][j] = (double)wtw[tri_index(i, j, 6)];
    }
    m[0][0] += (double)rows - (double)wtw[0];
    solve6(m, y, x);
    for (int k = 0; k < 6; k++) out[k] = (float)x[k];
}

} // namespace fri
