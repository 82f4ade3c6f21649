// solve6.hpp -- the 6 x 6 solves behind the context-model fit, ONE source for the host (fri_hip_solve6, fri_hip_fit_*_params) and for the
// device (the tail of fit_accumulate_kernel2 and fit_solve_kernel, k4_fit.hip): the same sequence of IEEE f64 operations on both sides (+, -, *, / and, on the rare route,
// sqrt; the library is built with -ffp-contract=off), so the device-side solve of the asynchronous encode chain returns bit for bit
// the parameters the host functions return for the same sums (tests/test_gpu_fit.py).
//
// The reference fits with an SVD least squares over n x 6 f32 design matrices (lstsq, context_modeling.rs:144-202, third-party
// arithmetic: parity unpinned). Here: normal equations M x = y from exact integer sums.
//   * M safely positive definite (every pivot of the LDL^T factorisation above 1e-8 of the largest diagonal entry - any image with
//     texture in the layer group): one solution, found without a square root and with six divisions (the pivots' reciprocals) - on the
//     device this runs on one lane at the end of a kernel of the chain, where a dependent f64 division costs ~0.1 us.
//   * otherwise (rank deficient or nearly so: flat regions, a feature that is zero everywhere): the minimum-norm solution through a cyclic
//     Jacobi eigen-decomposition with lstsq's relative cut-off - what the SVD returns, up to rounding.
#pragma once
#include <cmath>

#if defined(__HIPCC__)
#define FRI_HD __host__ __device__
#else
#define FRI_HD
#endif

namespace fri {

// Every array of a solve lives in a workspace the caller provides: on the host a local variable, on the device a piece of LDS (the last workgroup of
// the sums kernel solves in its tail, k4_fit.hip; as local arrays they sat in scratch memory, one dependent ~0.5 us round trip after the other: the
// solve kernels of round 3's first chain took 12 us each).
struct Solve6Work {
    double m[6][6], y[6], x[6];                   // in: M, y; out: x
    double l[6][6], d[6], inv[6], z[6];           // LDL^T route
    double a[6][6], v[6][6];                      // eigen-decomposition route
};

#if defined(__HIP_DEVICE_COMPILE__)
#define FRI_SOLVE_STAGE() asm volatile("" ::: "memory") /* values go back to the workspace between stages instead of piling up in registers */
#else
#define FRI_SOLVE_STAGE() ((void)0)
#endif

FRI_HD inline void solve6(Solve6Work &w) {
    {
        // M = L D L^T, L unit lower triangular. inv[j] = 1 / D[j]. No early exit (the loops unroll to straight-line code with static offsets): a failed pivot
        // only clears `ok`, and what the later columns then compute - possibly infinities - is never used.
        double dmax = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) dmax = w.m[i][i] > dmax ? w.m[i][i] : dmax;
        bool ok = dmax > 0.0 && dmax < 1.0e300; // (a NaN or an infinity fails the comparison or the bound)
        FRI_SOLVE_STAGE();
#pragma unroll
        for (int j = 0; j < 6; j++) {
            double dj = w.m[j][j];
#pragma unroll
            for (int k = 0; k < j; k++) dj -= w.l[j][k] * w.l[j][k] * w.d[k];
            ok = ok && dj > 1e-8 * dmax;
            w.d[j] = dj;
            w.inv[j] = 1.0 / dj;
#pragma unroll
            for (int i = j + 1; i < 6; i++) {
                double t = w.m[i][j];
#pragma unroll
                for (int k = 0; k < j; k++) t -= w.l[i][k] * w.l[j][k] * w.d[k];
                w.l[i][j] = t * w.inv[j];
            }
            FRI_SOLVE_STAGE();
        }
        if (ok) {
#pragma unroll
            for (int i = 0; i < 6; i++) { // L z = y
                double t = w.y[i];
#pragma unroll
                for (int k = 0; k < i; k++) t -= w.l[i][k] * w.z[k];
                w.z[i] = t;
            }
            FRI_SOLVE_STAGE();
#pragma unroll
            for (int i = 5; i >= 0; i--) { // D L^T x = z
                double t = w.z[i] * w.inv[i];
#pragma unroll
                for (int k = i + 1; k < 6; k++) t -= w.l[k][i] * w.x[k];
                w.x[i] = t;
            }
            return;
        }
    }
    // (rolled loops on purpose: unrolled, the 72 doubles of a and v travel in registers - more than the sums kernel, whose tail this is, may use)
    // cyclic Jacobi: a = V diag(lam) V^T; x = sum over the eigen-directions above the cut-off of v (v . y) / lam
#pragma unroll 1
    for (int i = 0; i < 6; i++)
#pragma unroll 1
        for (int j = 0; j < 6; j++) {
            w.a[i][j] = w.m[i][j];
            w.v[i][j] = i == j ? 1.0 : 0.0;
        }
#pragma unroll 1
    for (int sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
#pragma unroll 1
        for (int i = 0; i < 6; i++)
#pragma unroll 1
            for (int j = i + 1; j < 6; j++) off += w.a[i][j] * w.a[i][j];
        if (!(off >= 1e-300)) break; // converged (or not a number: nothing to iterate on)
#pragma unroll 1
        for (int pp = 0; pp < 6; pp++)
#pragma unroll 1
            for (int q = pp + 1; q < 6; q++) {
                const double apq = w.a[pp][q];
                if ((apq < 0 ? -apq : apq) < 1e-300) continue;
                const double theta = (w.a[q][q] - w.a[pp][pp]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / ((theta < 0 ? -theta : theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll 1
                for (int k = 0; k < 6; k++) {
                    const double akp = w.a[k][pp], akq = w.a[k][q];
                    w.a[k][pp] = c * akp - sn * akq;
                    w.a[k][q] = sn * akp + c * akq;
                }
#pragma unroll 1
                for (int k = 0; k < 6; k++) {
                    const double apk = w.a[pp][k], aqk = w.a[q][k];
                    w.a[pp][k] = c * apk - sn * aqk;
                    w.a[q][k] = sn * apk + c * aqk;
                }
#pragma unroll 1
                for (int k = 0; k < 6; k++) {
                    const double vkp = w.v[k][pp], vkq = w.v[k][q];
                    w.v[k][pp] = c * vkp - sn * vkq;
                    w.v[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    double lmax = 0.0;
#pragma unroll 1
    for (int i = 0; i < 6; i++) lmax = w.a[i][i] > lmax ? w.a[i][i] : lmax;
#pragma unroll 1
    for (int k = 0; k < 6; k++) w.x[k] = 0.0;
#pragma unroll 1
    for (int i = 0; i < 6; i++) {
        if (!(w.a[i][i] > 1e-12 * lmax)) continue; // rank-deficient direction: the minimum-norm solution leaves it at 0
        double proj = 0.0;
#pragma unroll 1
        for (int k = 0; k < 6; k++) proj += w.v[k][i] * w.y[k];
#pragma unroll 1
        for (int k = 0; k < 6; k++) w.x[k] += w.v[k][i] * proj / w.a[i][i];
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
FRI_HD inline void fit_value_group(const long long *gram, float *out /* [6] */, Solve6Work &w) {
    for (int i = 0; i < 6; i++) {
        w.y[i] = (double)gram[tri_index(i, 6, 7)];
        for (int j = 0; j < 6; j++) w.m[i][j] = (double)gram[tri_index(i, j, 7)];
    }
    solve6(w);
    for (int k = 0; k < 6; k++) out[k] = (float)w.x[k];
}

// optimize_width_prediction (context_modeling.rs:144-173) from wtw[21], wtr[6] over the Some rows; `rows` = height of the reference's
// matrix: its all-zero rows carry the constant feature 1 with residual 0.
FRI_HD inline void fit_width_group(const long long *wtw, const double *wtr, unsigned long long rows, float *out /* [6] */, Solve6Work &w) {
    for (int i = 0; i < 6; i++) {
        w.y[i] = wtr[i];
        for (int j = 0; j < 6; j++) w.m[i][j] = (double)wtw[tri_index(i, j, 6)];
    }
    w.m[0][0] += (double)rows - (double)wtw[0];
    solve6(w);
    for (int k = 0; k < 6; k++) out[k] = (float)w.x[k];
}

} // namespace fri
