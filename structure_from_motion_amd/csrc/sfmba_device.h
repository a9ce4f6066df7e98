// Device-side building blocks shared by every kernel of libsfmba: the projection model of
// /root/reference/bundle_adjustment.py:10-31 (cv2.projectPoints with a fixed K and no
// distortion) and that of /root/reference/ba_test.py:62-89 (rotation vector, translation, focal
// length, two radial coefficients) as two camera models with one interface, their analytic
// derivatives, tiny dense factorizations held in registers, and wave/block reductions for
// 64-lane wavefronts.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sfmba {

struct Intr { double fx, fy, cx, cy; };

// One pinhole camera as stored in HBM: 12 doubles = row-major rotation Q (9) then translation
// t (3), 96 bytes, 16-byte aligned, so a gather is six 16-byte loads.
struct Cam { double q[9]; double t[3]; };

__device__ __forceinline__ Cam load_cam(const double2* __restrict__ cams, int c) {
    const double2* p = cams + 6 * (size_t)c;
    double2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5];
    Cam k;
    k.q[0] = a0.x; k.q[1] = a0.y; k.q[2] = a1.x; k.q[3] = a1.y; k.q[4] = a2.x; k.q[5] = a2.y;
    k.q[6] = a3.x; k.q[7] = a3.y; k.q[8] = a4.x; k.t[0] = a4.y; k.t[1] = a5.x; k.t[2] = a5.y;
    return k;
}

// A value that is the same in every lane, moved into scalar registers.
__device__ __forceinline__ double wave_uniform(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// 1 / y by the hardware estimate and two Newton steps (five instructions against the fourteen of
// an IEEE division; last-bit accuracy, no scaling: |y| must lie well inside the normal range).  For
// derivative blocks only - residuals and costs use the division (in the observation passes, which
// wait on memory more than on the vector unit, the division costs nothing measurable).
__device__ __forceinline__ double newton_rcp(double y) {
    double r = __builtin_amdgcn_rcp(y);
    r = fma(r, fma(-y, r, 1.0), r);
    r = fma(r, fma(-y, r, 1.0), r);
    return r;
}

// Normalised image coordinates of a point in a camera: x = Y0/Z, y = Y1/Z, iz = 1/Z with the
// cv2.projectPoints guard (Z == 0 is treated as Z == 1).
struct Seed { double x, y, iz; };

__device__ __forceinline__ Seed project_seed(const Cam& k, double X0, double X1, double X2) {
    double Y0 = k.q[0] * X0 + k.q[1] * X1 + k.q[2] * X2 + k.t[0];
    double Y1 = k.q[3] * X0 + k.q[4] * X1 + k.q[5] * X2 + k.t[1];
    double Y2 = k.q[6] * X0 + k.q[7] * X1 + k.q[8] * X2 + k.t[2];
    Seed s;
    s.iz = (Y2 != 0.0) ? 1.0 / Y2 : 1.0;
    s.x = Y0 * s.iz;
    s.y = Y1 * s.iz;
    return s;
}

// project_seed with newton_rcp (same guard).
__device__ __forceinline__ Seed project_seed_jac(const Cam& k, double X0, double X1, double X2) {
    double Y0 = k.q[0] * X0 + k.q[1] * X1 + k.q[2] * X2 + k.t[0];
    double Y1 = k.q[3] * X0 + k.q[4] * X1 + k.q[5] * X2 + k.t[1];
    double Y2 = k.q[6] * X0 + k.q[7] * X1 + k.q[8] * X2 + k.t[2];
    Seed s;
    s.iz = (Y2 != 0.0) ? newton_rcp(Y2) : 1.0;
    s.x = Y0 * s.iz;
    s.y = Y1 * s.iz;
    return s;
}

// Residual (reprojection minus observation), bundle_adjustment.py:50.
__device__ __forceinline__ void residual(const Seed& s, const Intr& K, double u, double v,
                                         double& ru, double& rv) {
    ru = (K.fx * s.x + K.cx) - u;
    rv = (K.fy * s.y + K.cy) - v;
}

// d residual / d (dw, dt) for the left perturbation Y' = exp([dw]x) Y + dt.  Depends on the
// seed only.  jc[0..5] is the u row, jc[6..11] the v row.
__device__ __forceinline__ void jac_cam(const Seed& s, const Intr& K, double* jc) {
    const double fx = K.fx, fy = K.fy;
    jc[0] = -fx * s.x * s.y;
    jc[1] = fx * (1.0 + s.x * s.x);
    jc[2] = -fx * s.y;
    jc[3] = fx * s.iz;
    jc[4] = 0.0;
    jc[5] = -fx * s.x * s.iz;
    jc[6] = -fy * (1.0 + s.y * s.y);
    jc[7] = fy * s.x * s.y;
    jc[8] = fy * s.x;
    jc[9] = 0.0;
    jc[10] = fy * s.iz;
    jc[11] = -fy * s.y * s.iz;
}

// d residual / d X: rows a (u) and b (v).
__device__ __forceinline__ void jac_pt(const Seed& s, const Intr& K, const double* q,
                                       double* a, double* b) {
    const double fu = K.fx * s.iz, fv = K.fy * s.iz;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        a[k] = fu * (q[k] - s.x * q[6 + k]);
        b[k] = fv * (q[3 + k] - s.y * q[6 + k]);
    }
}

// w = Jc * d for a camera tangent d (6): two numbers.
__device__ __forceinline__ void jc_times(const Seed& s, const Intr& K, const double* d,
                                         double& w0, double& w1) {
    // v = dw x (x, y, 1) + dt * iz ; w0 = fx (v0 - x v2), w1 = fy (v1 - y v2)
    double v0 = d[1] - d[2] * s.y + d[3] * s.iz;
    double v1 = d[2] * s.x - d[0] + d[4] * s.iz;
    double v2 = d[0] * s.y - d[1] * s.x + d[5] * s.iz;
    w0 = K.fx * (v0 - s.x * v2);
    w1 = K.fy * (v1 - s.y * v2);
}

// out(3) += Jp^T * (w0, w1) = Q^T e with e = (fu w0, fv w1, -(x fu w0 + y fv w1)).
__device__ __forceinline__ void jpT_times_acc(const Seed& s, const Intr& K, const double* q,
                                              double w0, double w1, double* out) {
    const double e0 = K.fx * s.iz * w0, e1 = K.fy * s.iz * w1;
    const double e2 = -(s.x * e0 + s.y * e1);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] += q[k] * e0 + q[3 + k] * e1 + q[6 + k] * e2;
}

// (w0, w1) = Jp * d for a point step d (3).
__device__ __forceinline__ void jp_times(const Seed& s, const Intr& K, const double* q,
                                         const double* d, double& w0, double& w1) {
    const double y0 = q[0] * d[0] + q[1] * d[1] + q[2] * d[2];
    const double y1 = q[3] * d[0] + q[4] * d[1] + q[5] * d[2];
    const double y2 = q[6] * d[0] + q[7] * d[1] + q[8] * d[2];
    w0 = K.fx * s.iz * (y0 - s.x * y2);
    w1 = K.fy * s.iz * (y1 - s.y * y2);
}

// out(6) += Jc^T * (w0, w1).
__device__ __forceinline__ void jcT_times_acc(const Seed& s, const Intr& K, double w0, double w1,
                                              double* out) {
    const double a = K.fx * w0, b = K.fy * w1;
    out[0] += -a * s.x * s.y - b * (1.0 + s.y * s.y);
    out[1] += a * (1.0 + s.x * s.x) + b * s.x * s.y;
    out[2] += -a * s.y + b * s.x;
    out[3] += a * s.iz;
    out[4] += b * s.iz;
    out[5] += -(a * s.x + b * s.y) * s.iz;
}

// Packed upper triangle of a symmetric N x N matrix, row-major: (i, j), i <= j.
__host__ __device__ constexpr int upidx(int n, int i, int j) { return i * n - (i * (i - 1)) / 2 + (j - i); }

// Inverse of a symmetric positive definite 3x3 given as packed upper (00 01 02 11 12 22).
__device__ __forceinline__ void spd3_inverse(const double* a, double* inv) {
    const double tiny = 1e-300;
    double l00 = sqrt(fmax(a[0], tiny));
    double i00 = 1.0 / l00;
    double l10 = a[1] * i00, l20 = a[2] * i00;
    double l11 = sqrt(fmax(a[3] - l10 * l10, tiny));
    double i11 = 1.0 / l11;
    double l21 = (a[4] - l20 * l10) * i11;
    double l22 = sqrt(fmax(a[5] - l20 * l20 - l21 * l21, tiny));
    double i22 = 1.0 / l22;
    double i10 = -l10 * i00 * i11;
    double i21 = -l21 * i11 * i22;
    double i20 = -(l20 * i00 + l21 * i10) * i22;
    inv[0] = i00 * i00 + i10 * i10 + i20 * i20;
    inv[1] = i10 * i11 + i20 * i21;
    inv[2] = i20 * i22;
    inv[3] = i11 * i11 + i21 * i21;
    inv[4] = i21 * i22;
    inv[5] = i22 * i22;
}

// In-register Cholesky of a symmetric positive definite N x N (full storage, row-major): A = L L'.
// Returns false when a pivot is not positive; A is destroyed, the lower triangle of Li receives
// L^-1 (the strict upper triangle is left untouched).
template <int N>
__device__ __forceinline__ bool spd_chol_linv(double (&A)[N][N], double (&Li)[N][N]) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
        if (!(d > 0.0)) { ok = false; d = 1.0; }
        double l = sqrt(d), il = 1.0 / l;
        A[j][j] = l;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double v = A[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v -= A[i][k] * A[j][k];
            A[i][j] = v * il;
        }
    }
    // Linv (lower) by forward substitution
#pragma unroll
    for (int j = 0; j < N; ++j) {
        Li[j][j] = 1.0 / A[j][j];
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) v -= A[i][k] * Li[k][j];
            Li[i][j] = v / A[i][i];
        }
    }
    return ok;
}

// Inverse of a symmetric positive definite N x N through its Cholesky factor: Ainv = Li' Li.
// Returns false when a pivot is not positive; A is destroyed.
template <int N>
__device__ __forceinline__ bool spd_inverse(double (&A)[N][N], double (&Ainv)[N][N], double (&Li)[N][N]) {
    const bool ok = spd_chol_linv(A, Li);
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = i; j < N; ++j) {
            double v = 0.0;
#pragma unroll
            for (int k = j; k < N; ++k) v += Li[k][i] * Li[k][j];
            Ainv[i][j] = v;
            Ainv[j][i] = v;
        }
    return ok;
}
template <int N>
__device__ __forceinline__ bool spd_inverse(double (&A)[N][N], double (&Ainv)[N][N]) {
    double Li[N][N];
    return spd_inverse(A, Ainv, Li);
}
__device__ __forceinline__ bool spd6_chol_linv(double (&A)[6][6], double (&Li)[6][6]) { return spd_chol_linv(A, Li); }
__device__ __forceinline__ bool spd6_inverse(double (&A)[6][6], double (&Ainv)[6][6], double (&Li)[6][6]) { return spd_inverse(A, Ainv, Li); }
__device__ __forceinline__ bool spd6_inverse(double (&A)[6][6], double (&Ainv)[6][6]) { return spd_inverse(A, Ainv); }

// exp([w]x) as a row-major 3x3.
__device__ __forceinline__ void so3_exp(const double* w, double* E) {
    const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    double a, b;
    if (t2 < 1e-8) {
        a = 1.0 - t2 / 6.0;
        b = 0.5 - t2 / 24.0;
    } else {
        const double t = sqrt(t2);
        a = sin(t) / t;
        b = (1.0 - cos(t)) / t2;
    }
    // K = [w]x ; K^2 = w w^T - t2 I
    E[0] = 1.0 + b * (w[0] * w[0] - t2);
    E[1] = -a * w[2] + b * w[0] * w[1];
    E[2] = a * w[1] + b * w[0] * w[2];
    E[3] = a * w[2] + b * w[0] * w[1];
    E[4] = 1.0 + b * (w[1] * w[1] - t2);
    E[5] = -a * w[0] + b * w[1] * w[2];
    E[6] = -a * w[1] + b * w[0] * w[2];
    E[7] = a * w[0] + b * w[1] * w[2];
    E[8] = 1.0 + b * (w[2] * w[2] - t2);
}

// ---- camera models ---------------------------------------------------------------------------
// A model names the camera record in HBM, the size of a camera's tangent space and the per-
// observation arithmetic.  Every model perturbs the pose from the left, Y' = exp([dw]x) Y + dt,
// so that the pose columns of a camera block depend on the seed (x, y, 1/Z) alone; what differs
// is the map from the normalised image point to pixels and any further per-camera parameters.
enum CameraModel { MODEL_PINHOLE = 0, MODEL_BAL = 1 };

// The reference's bundle adjustment (/root/reference/bundle_adjustment.py:10-31): one shared K,
// no distortion, tangent (dw, dt).
struct Pinhole {
    static constexpr int ND = 6;              // tangent dimensions of a camera
    static constexpr int NU = 21;             // packed upper triangle of an ND x ND block
    static constexpr int NV = 27;             // NU + ND: what a chunk accumulates per camera
    static constexpr int CW = 6;              // 16-byte words of a camera record
    static constexpr bool kDiagonalPixelMap = true;    // d pixel / d (x, y) is diagonal: jc[4] = jc[ND + 3] = 0
    using Cam = sfmba::Cam;
    struct Obs { Seed s; Intr K; };
    static __device__ __forceinline__ Cam load(const double2* __restrict__ cams, int c) { return load_cam(cams, c); }
    // rotation only (the translation is left unset)
    static __device__ __forceinline__ Cam load_rot(const double2* __restrict__ cams, int c) {
        const double2* p = cams + 6 * (size_t)c;
        const double2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4];
        Cam k;
        k.q[0] = a0.x; k.q[1] = a0.y; k.q[2] = a1.x; k.q[3] = a1.y; k.q[4] = a2.x; k.q[5] = a2.y;
        k.q[6] = a3.x; k.q[7] = a3.y; k.q[8] = a4.x; k.t[0] = k.t[1] = k.t[2] = 0.0;
        return k;
    }
    static __device__ __forceinline__ void store(double2* __restrict__ cams, int c, const Cam& k) {
        double2* dst = cams + 6 * (size_t)c;
        dst[0] = make_double2(k.q[0], k.q[1]); dst[1] = make_double2(k.q[2], k.q[3]); dst[2] = make_double2(k.q[4], k.q[5]);
        dst[3] = make_double2(k.q[6], k.q[7]); dst[4] = make_double2(k.q[8], k.t[0]); dst[5] = make_double2(k.t[1], k.t[2]);
    }
    // a camera that is the same in every lane, moved into scalar registers (rotation only)
    static __device__ __forceinline__ Cam uniform_rot(const Cam& k) {
        Cam u;
#pragma unroll
        for (int i = 0; i < 9; ++i) u.q[i] = wave_uniform(k.q[i]);
        u.t[0] = u.t[1] = u.t[2] = 0.0;
        return u;
    }
    // the whole camera in scalar registers
    static __device__ __forceinline__ Cam uniform(const Cam& k) {
        Cam u = uniform_rot(k);
#pragma unroll
        for (int i = 0; i < 3; ++i) u.t[i] = wave_uniform(k.t[i]);
        return u;
    }
    static __device__ __forceinline__ Seed seed(const Cam& k, double X0, double X1, double X2) { return project_seed(k, X0, X1, X2); }
    static __device__ __forceinline__ Seed seed_jac(const Cam& k, double X0, double X1, double X2) { return project_seed_jac(k, X0, X1, X2); }
    static __device__ __forceinline__ Obs obs(const Seed& s, const Cam&, const Intr& K) { return Obs{s, K}; }
    static __device__ __forceinline__ void residual(const Obs& o, double u, double v, double& ru, double& rv) { sfmba::residual(o.s, o.K, u, v, ru, rv); }
    static __device__ __forceinline__ void jac_cam(const Obs& o, double* jc) { sfmba::jac_cam(o.s, o.K, jc); }
    static __device__ __forceinline__ void jac_pt(const Obs& o, const double* q, double* a, double* b) { sfmba::jac_pt(o.s, o.K, q, a, b); }
    static __device__ __forceinline__ void jc_times(const Obs& o, const double* d, double& w0, double& w1) { sfmba::jc_times(o.s, o.K, d, w0, w1); }
    static __device__ __forceinline__ void jpT_times_acc(const Obs& o, const double* q, double w0, double w1, double* out) { sfmba::jpT_times_acc(o.s, o.K, q, w0, w1, out); }
    static __device__ __forceinline__ void jp_times(const Obs& o, const double* q, const double* d, double& w0, double& w1) { sfmba::jp_times(o.s, o.K, q, d, w0, w1); }
    static __device__ __forceinline__ void jcT_times_acc(const Obs& o, double w0, double w1, double* out) { sfmba::jcT_times_acc(o.s, o.K, w0, w1, out); }
    // the camera moved by the tangent step w; returns the camera's share of |x|^2 (the raw
    // 12-number camera: 3 for the rotation entries, then the translation)
    static __device__ __forceinline__ Cam retract(const Cam& k, const double* w, double& x2) {
        double E[9];
        so3_exp(w, E);
        Cam o;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                o.q[3 * i + j] = E[3 * i] * k.q[j] + E[3 * i + 1] * k.q[3 + j] + E[3 * i + 2] * k.q[6 + j];
            o.t[i] = E[3 * i] * k.t[0] + E[3 * i + 1] * k.t[1] + E[3 * i + 2] * k.t[2] + w[3 + i];
        }
        x2 = 3.0 + k.t[0] * k.t[0] + k.t[1] * k.t[1] + k.t[2] * k.t[2];
        return o;
    }
};

// The camera of the "Bundle Adjustment in the Large" files (/root/reference/ba_test.py:78-89):
// P = R X + t, p = -P[:2] / P[2], pixel = f (1 + k1 |p|^2 + k2 |p|^4) p, with f, k1, k2 per
// camera and estimated: tangent (dw, dt, df, dk1, dk2).  The rotation is kept as a matrix in HBM
// (the rotation vector of the file is converted on the host).
struct Bal {
    static constexpr int ND = 9;
    static constexpr int NU = 45;
    static constexpr int NV = 54;
    static constexpr int CW = 8;              // q[9] t[3] f k1 k2 pad
    static constexpr bool kDiagonalPixelMap = false;
    struct Cam { double q[9]; double t[3]; double f, k1, k2; };
    // a = d pixel / d (x, y) (symmetric 2x2), g = (r, f n, f n^2): d pixel / d (f, k1, k2) = -(x, y) g
    struct Obs { Seed s; double a00, a01, a11, g0, g1, g2, fr; };
    static __device__ __forceinline__ Cam load(const double2* __restrict__ cams, int c) {
        const double2* p = cams + 8 * (size_t)c;
        const double2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5], a6 = p[6], a7 = p[7];
        Cam k;
        k.q[0] = a0.x; k.q[1] = a0.y; k.q[2] = a1.x; k.q[3] = a1.y; k.q[4] = a2.x; k.q[5] = a2.y;
        k.q[6] = a3.x; k.q[7] = a3.y; k.q[8] = a4.x; k.t[0] = a4.y; k.t[1] = a5.x; k.t[2] = a5.y;
        k.f = a6.x; k.k1 = a6.y; k.k2 = a7.x;
        return k;
    }
    static __device__ __forceinline__ Cam load_rot(const double2* __restrict__ cams, int c) {
        const double2* p = cams + 8 * (size_t)c;
        const double2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a6 = p[6], a7 = p[7];
        Cam k;
        k.q[0] = a0.x; k.q[1] = a0.y; k.q[2] = a1.x; k.q[3] = a1.y; k.q[4] = a2.x; k.q[5] = a2.y;
        k.q[6] = a3.x; k.q[7] = a3.y; k.q[8] = a4.x; k.t[0] = k.t[1] = k.t[2] = 0.0;
        k.f = a6.x; k.k1 = a6.y; k.k2 = a7.x;
        return k;
    }
    static __device__ __forceinline__ void store(double2* __restrict__ cams, int c, const Cam& k) {
        double2* dst = cams + 8 * (size_t)c;
        dst[0] = make_double2(k.q[0], k.q[1]); dst[1] = make_double2(k.q[2], k.q[3]); dst[2] = make_double2(k.q[4], k.q[5]);
        dst[3] = make_double2(k.q[6], k.q[7]); dst[4] = make_double2(k.q[8], k.t[0]); dst[5] = make_double2(k.t[1], k.t[2]);
        dst[6] = make_double2(k.f, k.k1); dst[7] = make_double2(k.k2, 0.0);
    }
    static __device__ __forceinline__ Cam uniform_rot(const Cam& k) {
        Cam u;
#pragma unroll
        for (int i = 0; i < 9; ++i) u.q[i] = wave_uniform(k.q[i]);
        u.t[0] = u.t[1] = u.t[2] = 0.0;
        u.f = wave_uniform(k.f); u.k1 = wave_uniform(k.k1); u.k2 = wave_uniform(k.k2);
        return u;
    }
    static __device__ __forceinline__ Cam uniform(const Cam& k) {
        Cam u = uniform_rot(k);
#pragma unroll
        for (int i = 0; i < 3; ++i) u.t[i] = wave_uniform(k.t[i]);
        return u;
    }
    // no guard against Z == 0: the reference divides (ba_test.py:82) and SciPy then refuses the
    // non-finite residual
    static __device__ __forceinline__ Seed seed(const Cam& k, double X0, double X1, double X2) {
        const double Y0 = k.q[0] * X0 + k.q[1] * X1 + k.q[2] * X2 + k.t[0];
        const double Y1 = k.q[3] * X0 + k.q[4] * X1 + k.q[5] * X2 + k.t[1];
        const double Y2 = k.q[6] * X0 + k.q[7] * X1 + k.q[8] * X2 + k.t[2];
        Seed s;
        s.iz = 1.0 / Y2;
        s.x = Y0 * s.iz;
        s.y = Y1 * s.iz;
        return s;
    }
    static __device__ __forceinline__ Seed seed_jac(const Cam& k, double X0, double X1, double X2) {
        const double Y0 = k.q[0] * X0 + k.q[1] * X1 + k.q[2] * X2 + k.t[0];
        const double Y1 = k.q[3] * X0 + k.q[4] * X1 + k.q[5] * X2 + k.t[1];
        const double Y2 = k.q[6] * X0 + k.q[7] * X1 + k.q[8] * X2 + k.t[2];
        Seed s;
        s.iz = newton_rcp(Y2);
        s.x = Y0 * s.iz;
        s.y = Y1 * s.iz;
        return s;
    }
    static __device__ __forceinline__ Obs obs(const Seed& s, const Cam& k, const Intr&) {
        const double n = s.x * s.x + s.y * s.y;
        const double r = 1.0 + n * (k.k1 + k.k2 * n);
        const double rp2 = 2.0 * (k.k1 + 2.0 * k.k2 * n);         // 2 dr/dn
        Obs o;
        o.s = s;
        o.a00 = -k.f * (r + rp2 * s.x * s.x);
        o.a01 = -k.f * rp2 * s.x * s.y;
        o.a11 = -k.f * (r + rp2 * s.y * s.y);
        o.g0 = r; o.g1 = k.f * n; o.g2 = k.f * n * n; o.fr = k.f * r;
        return o;
    }
    static __device__ __forceinline__ void residual(const Obs& o, double u, double v, double& ru, double& rv) {
        ru = -o.fr * o.s.x - u;
        rv = -o.fr * o.s.y - v;
    }
    // rows of d (x, y) / d (dw, dt): the pinhole rows with unit focal lengths
    static __device__ __forceinline__ void normalised_rows(const Seed& s, double* nx, double* ny) {
        nx[0] = -s.x * s.y; nx[1] = 1.0 + s.x * s.x; nx[2] = -s.y; nx[3] = s.iz; nx[4] = 0.0; nx[5] = -s.x * s.iz;
        ny[0] = -(1.0 + s.y * s.y); ny[1] = s.x * s.y; ny[2] = s.x; ny[3] = 0.0; ny[4] = s.iz; ny[5] = -s.y * s.iz;
    }
    // jc[0..8] is the u row, jc[9..17] the v row
    static __device__ __forceinline__ void jac_cam(const Obs& o, double* jc) {
        double nx[6], ny[6];
        normalised_rows(o.s, nx, ny);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            jc[i] = o.a00 * nx[i] + o.a01 * ny[i];
            jc[9 + i] = o.a01 * nx[i] + o.a11 * ny[i];
        }
        jc[6] = -o.s.x * o.g0; jc[7] = -o.s.x * o.g1; jc[8] = -o.s.x * o.g2;
        jc[15] = -o.s.y * o.g0; jc[16] = -o.s.y * o.g1; jc[17] = -o.s.y * o.g2;
    }
    static __device__ __forceinline__ void jac_pt(const Obs& o, const double* q, double* a, double* b) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double an = o.s.iz * (q[k] - o.s.x * q[6 + k]);
            const double bn = o.s.iz * (q[3 + k] - o.s.y * q[6 + k]);
            a[k] = o.a00 * an + o.a01 * bn;
            b[k] = o.a01 * an + o.a11 * bn;
        }
    }
    static __device__ __forceinline__ void jc_times(const Obs& o, const double* d, double& w0, double& w1) {
        const Seed& s = o.s;
        const double v0 = d[1] - d[2] * s.y + d[3] * s.iz;
        const double v1 = d[2] * s.x - d[0] + d[4] * s.iz;
        const double v2 = d[0] * s.y - d[1] * s.x + d[5] * s.iz;
        const double n0 = v0 - s.x * v2, n1 = v1 - s.y * v2;
        const double e = o.g0 * d[6] + o.g1 * d[7] + o.g2 * d[8];
        w0 = o.a00 * n0 + o.a01 * n1 - s.x * e;
        w1 = o.a01 * n0 + o.a11 * n1 - s.y * e;
    }
    static __device__ __forceinline__ void jpT_times_acc(const Obs& o, const double* q, double w0, double w1, double* out) {
        const double e0 = o.s.iz * (o.a00 * w0 + o.a01 * w1), e1 = o.s.iz * (o.a01 * w0 + o.a11 * w1);
        const double e2 = -(o.s.x * e0 + o.s.y * e1);
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] += q[k] * e0 + q[3 + k] * e1 + q[6 + k] * e2;
    }
    static __device__ __forceinline__ void jp_times(const Obs& o, const double* q, const double* d, double& w0, double& w1) {
        const double y0 = q[0] * d[0] + q[1] * d[1] + q[2] * d[2];
        const double y1 = q[3] * d[0] + q[4] * d[1] + q[5] * d[2];
        const double y2 = q[6] * d[0] + q[7] * d[1] + q[8] * d[2];
        const double n0 = o.s.iz * (y0 - o.s.x * y2), n1 = o.s.iz * (y1 - o.s.y * y2);
        w0 = o.a00 * n0 + o.a01 * n1;
        w1 = o.a01 * n0 + o.a11 * n1;
    }
    static __device__ __forceinline__ void jcT_times_acc(const Obs& o, double w0, double w1, double* out) {
        const Seed& s = o.s;
        const double a = o.a00 * w0 + o.a01 * w1, b = o.a01 * w0 + o.a11 * w1;
        out[0] += -a * s.x * s.y - b * (1.0 + s.y * s.y);
        out[1] += a * (1.0 + s.x * s.x) + b * s.x * s.y;
        out[2] += -a * s.y + b * s.x;
        out[3] += a * s.iz;
        out[4] += b * s.iz;
        out[5] += -(a * s.x + b * s.y) * s.iz;
        const double e = -(s.x * w0 + s.y * w1);
        out[6] += o.g0 * e; out[7] += o.g1 * e; out[8] += o.g2 * e;
    }
    // x2: the camera's share of |x|^2 in the file's parameters (rotation vector, t, f, k1, k2)
    static __device__ __forceinline__ Cam retract(const Cam& k, const double* w, double& x2) {
        double E[9];
        so3_exp(w, E);
        Cam o;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                o.q[3 * i + j] = E[3 * i] * k.q[j] + E[3 * i + 1] * k.q[3 + j] + E[3 * i + 2] * k.q[6 + j];
            o.t[i] = E[3 * i] * k.t[0] + E[3 * i + 1] * k.t[1] + E[3 * i + 2] * k.t[2] + w[3 + i];
        }
        o.f = k.f + w[6]; o.k1 = k.k1 + w[7]; o.k2 = k.k2 + w[8];
        const double rx = k.q[7] - k.q[5], ry = k.q[2] - k.q[6], rz = k.q[3] - k.q[1];
        const double sn = 0.5 * sqrt(rx * rx + ry * ry + rz * rz), cs = 0.5 * (k.q[0] + k.q[4] + k.q[8] - 1.0);
        const double ang = atan2(sn, cs);
        x2 = ang * ang + k.t[0] * k.t[0] + k.t[1] * k.t[1] + k.t[2] * k.t[2] + k.f * k.f + k.k1 * k.k1 + k.k2 * k.k2;
        return o;
    }
};

// Workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2).  This maps the
// hardware block id to a logical one so that each XCD works on one contiguous range of the
// grid: neighbouring logical blocks touch neighbouring cameras / points, whose gathered rows
// then stay in that XCD's L2.  Bijective for every grid size; affects speed only.
__device__ __forceinline__ int xcd_block(int b, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = b & 7, i = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// ---- reductions over a 64-lane wavefront / a workgroup ---------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Sum over the workgroup (blockDim.x a multiple of 64, at most 1024); the result is valid in
// thread 0.  scratch must hold 16 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}
__device__ __forceinline__ double block_max(double v, double* scratch) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < nw; ++i) r = fmax(r, scratch[i]);
    return r;
}

// Reduce N per-thread values over the workgroup (4 wavefronts) with ONE workgroup barrier and no
// shuffle chains: every lane stores its values as a row of its wavefront's LDS tile, 8 N lanes
// add 8 rows each for one value, N lanes add the 8 parts, and after the barrier thread i < N adds
// the four wavefront results of value i.  Bit i of max_mask selects a maximum (of non-negative
// values) instead of a sum for value i.  The order of the operations is fixed.  `scratch` holds
// 4 * (64 N + 8 N) + 4 N doubles; thread i < N returns the result of value i, others 0.
template <int N>
__device__ __forceinline__ double block_reduce_n(const double (&v)[N], unsigned max_mask, double* scratch) {
    static_assert(8 * N <= 64, "one lane per (value, part)");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double* tile = scratch + w * (72 * N);
    double* parts = tile + 64 * N;
    double* red = scratch + 4 * (72 * N);
#pragma unroll
    for (int i = 0; i < N; ++i) tile[lane * N + i] = v[i];
    __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    if (lane < 8 * N) {
        const int k = lane % N, part = lane / N;
        const bool is_max = (max_mask >> k) & 1u;
        double a = 0.0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const double x = tile[(8 * part + r) * N + k];
            a = is_max ? fmax(a, x) : a + x;
        }
        parts[lane] = a;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    if (lane < N) {
        const bool is_max = (max_mask >> lane) & 1u;
        double a = 0.0;
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) {
            const double x = parts[p8 * N + lane];
            a = is_max ? fmax(a, x) : a + x;
        }
        red[w * N + lane] = a;
    }
    __syncthreads();
    double out = 0.0;
    if (threadIdx.x < N) {
        const bool is_max = (max_mask >> threadIdx.x) & 1u;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) {
            const double x = red[ww * N + threadIdx.x];
            out = is_max ? fmax(out, x) : out + x;
        }
    }
    return out;
}
template <int N> constexpr int block_reduce_scratch() { return 4 * (72 * N) + 4 * N; }

// Sum N per-lane values over the 64 lanes through a per-wave LDS tile instead of log-step
// shuffles: in passes of H values every lane stores its H values as one row (row stride H+2
// doubles, 16-byte aligned); then P = 64 / H lanes share a column - lane e + k H adds rows k, k + P,
// k + 2 P, ... of column e - and lane e < H collects the P partial sums.  After the call
// lane e (< H) holds the totals of values e, H+e, 2H+e, ... in out[0], out[1], ...
// (One lane per column walked all 64 rows with 64 - H lanes idle: per pass 64 LDS reads and 64
// additions of a wavefront against 22 + 22 + 6 now, a tenth of the instructions of a 512-observation
// chunk.)
// buf: 64*(H+2) doubles owned by this wave.  LDS operations of one wave complete in order, so a
// wave-level barrier for the compiler plus a wait for outstanding LDS traffic is all the
// synchronisation the hand-over between lanes needs.
template <int N, int H>
__device__ __forceinline__ void wave_sum_lds(const double (&acc)[N], double* buf, int lane,
                                             double (&out)[(N + H - 1) / H]) {
    static_assert(H % 2 == 0, "rows are written as 16-byte pairs");
    constexpr int LD = H + 2;
    constexpr int P = 64 / H;                    // lanes per column
    constexpr int R = (64 + P - 1) / P;          // rows per lane (the last one only where it exists)
    const int part = lane / H, e = lane - part * H;
#pragma unroll
    for (int p = 0; p < (N + H - 1) / H; ++p) {
        double2* row = reinterpret_cast<double2*>(buf + lane * LD);
#pragma unroll
        for (int i = 0; i < H; i += 2) {
            const int e0 = p * H + i, e1 = e0 + 1;
            row[i / 2] = make_double2(e0 < N ? acc[e0] : 0.0, e1 < N ? acc[e1] : 0.0);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        if (part < P) {
            const double* col = buf + part * LD + e;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if ((R - 1) * P + P - 1 < 64 || r < R - 1 || part + P * r < 64) {
                    const double v = col[r * P * LD];
                    if (r % 4 == 0) s0 += v; else if (r % 4 == 1) s1 += v; else if (r % 4 == 2) s2 += v; else s3 += v;
                }
            }
        }
        double s = (s0 + s1) + (s2 + s3);
        double tot = s;
#pragma unroll
        for (int k = 1; k < P; ++k) tot += __shfl(s, lane + k * H, 64);       // (meaningful in lanes < H)
        out[p] = tot;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- camera-sized dense helpers --------------------------------------------------------------
// y = (U + lambda clamp(diag U)) x with U packed upper (N (N + 1) / 2); y = M x with M full
// row-major N x N.
template <int N>
__device__ __forceinline__ void ustar_times(const double* U, double lambda, const double (&x)[N], double (&y)[N]) {
    double A[N][N];
    int n = 0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = i; j < N; ++j) { A[i][j] = U[n]; A[j][i] = U[n]; ++n; }
#pragma unroll
    for (int i = 0; i < N; ++i) A[i][i] += lambda * fmin(fmax(A[i][i], 1e-6), 1e32);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < N; ++j) v += A[i][j] * x[j];
        y[i] = v;
    }
}

__device__ __forceinline__ void mat6_times(const double* M, const double* x, double* y) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double v = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) v += M[6 * i + j] * x[j];
        y[i] = v;
    }
}

// ---- segments inside a wavefront ----------------------------------------------------------
// Lanes carry a key; a segment is a run of consecutive lanes with the same key (the
// observations of one point).  Inclusive segmented sums by log-step shuffles; the loop stops as
// soon as no lane has a same-segment partner at the current distance.
template <int N>
__device__ __forceinline__ void seg_scan(double (&v)[N], int key, int lane) {
    for (int off = 1; off < 64; off <<= 1) {
        const int ku = __shfl_up(key, off, 64);
        const bool take = lane >= off && ku == key;
        if (!__any(take)) break;
        // (one fused multiply-add per value instead of an add and two selects: v + 1 u and v + 0 u
        // are exact for finite u, and a non-finite value anywhere in a linearisation fails the
        // solve whichever point it reaches)
        const double m = take ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double u = __shfl_up(v[i], off, 64);
            v[i] = fma(u, m, v[i]);
        }
    }
}
// Lane index of the last lane of this lane's segment.
__device__ __forceinline__ int seg_last(int key, int lane) {
    const int kn = __shfl_down(key, 1, 64);
    const bool is_last = lane == 63 || kn != key;
    const unsigned long long m = __ballot(is_last);
    return lane + __builtin_ctzll(m >> lane);
}

}  // namespace sfmba
