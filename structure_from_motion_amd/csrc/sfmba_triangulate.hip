// Batched two-view linear triangulation and reprojection errors (gfx950, fp64).
//
// Takes the place of the reference's triangulate_points (utils.py:176-211): cv.triangulatePoints
// builds, per correspondence, the 4x4 matrix
//     [ x1 P1[2] - P1[0] ;  y1 P1[2] - P1[1] ;  x2 P2[2] - P2[0] ;  y2 P2[2] - P2[1] ]
// and returns the right singular vector of its smallest singular value; the reference then
// de-homogenises (cv.convertPointsFromHomogeneous: a zero last coordinate divides by 1) and
// measures |x - dehom(K (R X + t))| in both views with calculate_reprojection_error
// (utils.py:337-352).  One thread per correspondence: the 4x4 singular vectors come from a
// one-sided Jacobi iteration held in registers (no normal equations, so the small singular value
// is resolved to working precision), both projection matrices are wave-uniform.
#include "sfmba_internal.h"

namespace sfmba {

struct TwoViews { double P1[12]; double P2[12]; };

__device__ __forceinline__ void rotate_pair(double (&G)[4][4], double (&V)[4][4], int p, int q, bool& rotated) {
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) { alpha += G[r][p] * G[r][p]; beta += G[r][q] * G[r][q]; gamma += G[r][p] * G[r][q]; }
    if (!(fabs(gamma) > 2.3e-16 * sqrt(alpha * beta))) return;
    rotated = true;
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double gp = G[r][p], gq = G[r][q];
        G[r][p] = c * gp - s * gq;
        G[r][q] = s * gp + c * gq;
        const double vp = V[r][p], vq = V[r][q];
        V[r][p] = c * vp - s * vq;
        V[r][q] = s * vp + c * vq;
    }
}

__device__ __forceinline__ double reprojection_error(const double* P, double X0, double X1, double X2, double2 x) {
    const double u = P[0] * X0 + P[1] * X1 + P[2] * X2 + P[3];
    const double v = P[4] * X0 + P[5] * X1 + P[6] * X2 + P[7];
    const double w = P[8] * X0 + P[9] * X1 + P[10] * X2 + P[11];
    const double sc = w != 0.0 ? 1.0 / w : 1.0;
    const double du = x.x - u * sc, dv = x.y - v * sc;
    return sqrt(du * du + dv * dv);
}

__global__ __launch_bounds__(kBlock) void k_triangulate(TwoViews tv, int n, const double2* __restrict__ x1,
                                                        const double2* __restrict__ x2, double* __restrict__ X,
                                                        double* __restrict__ err1, double* __restrict__ err2) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double2 a = x1[i], b = x2[i];
    double G[4][4], V[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        G[0][k] = a.x * tv.P1[8 + k] - tv.P1[k];
        G[1][k] = a.y * tv.P1[8 + k] - tv.P1[4 + k];
        G[2][k] = b.x * tv.P2[8 + k] - tv.P2[k];
        G[3][k] = b.y * tv.P2[8 + k] - tv.P2[4 + k];
#pragma unroll
        for (int r = 0; r < 4; ++r) V[r][k] = r == k ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        rotate_pair(G, V, 0, 1, rotated); rotate_pair(G, V, 0, 2, rotated); rotate_pair(G, V, 0, 3, rotated);
        rotate_pair(G, V, 1, 2, rotated); rotate_pair(G, V, 1, 3, rotated); rotate_pair(G, V, 2, 3, rotated);
        if (!rotated) break;
    }
    // the column of smallest norm belongs to the smallest singular value
    double best = 0.0, h[4] = {0.0, 0.0, 0.0, 1.0};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        double nk = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) nk += G[r][k] * G[r][k];
        if (k == 0 || nk < best) {
            best = nk;
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = V[r][k];
        }
    }
    const double sc = h[3] != 0.0 ? 1.0 / h[3] : 1.0;
    const double X0 = h[0] * sc, X1 = h[1] * sc, X2 = h[2] * sc;
    X[3 * (size_t)i] = X0; X[3 * (size_t)i + 1] = X1; X[3 * (size_t)i + 2] = X2;
    err1[i] = reprojection_error(tv.P1, X0, X1, X2, a);
    err2[i] = reprojection_error(tv.P2, X0, X1, X2, b);
}

// Reprojection errors of given points in one view (calculate_reprojection_error, batched).
__global__ __launch_bounds__(kBlock) void k_reprojection_errors(TwoViews tv, int n, const double* __restrict__ X,
                                                                const double2* __restrict__ x, double* __restrict__ err) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    err[i] = reprojection_error(tv.P1, X[3 * (size_t)i], X[3 * (size_t)i + 1], X[3 * (size_t)i + 2], x[i]);
}

void launch_triangulate(const double* P1, const double* P2, int n, const double2* x1, const double2* x2,
                        double* X, double* err1, double* err2, hipStream_t st) {
    TwoViews tv;
    for (int k = 0; k < 12; ++k) { tv.P1[k] = P1[k]; tv.P2[k] = P2[k]; }
    if (n > 0) hipLaunchKernelGGL(k_triangulate, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, tv, n, x1, x2, X, err1, err2);
}

void launch_reprojection_errors(const double* P, int n, const double* X, const double2* x, double* err, hipStream_t st) {
    TwoViews tv;
    for (int k = 0; k < 12; ++k) { tv.P1[k] = P[k]; tv.P2[k] = 0.0; }
    if (n > 0) hipLaunchKernelGGL(k_reprojection_errors, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, tv, n, X, x, err);
}

}  // namespace sfmba
