// HIP kernels of libsfmba for gfx950 (MI355X).  fp64 throughout.
//
// Two observation orders are kept in HBM.  Point-major ("pm"): the observations of one point
// are contiguous, one thread owns one point and keeps its 3x3 block and 3-vector sums in
// registers.  Camera-major ("cm"): the observations of one camera are contiguous and cut into
// chunks of at most kChunk; one 64-lane wavefront owns one chunk, the camera is wave-uniform,
// per-observation 6-vector / 6x6 contributions are summed in registers, reduced across the
// wavefront with cross-lane shuffles and written as one partial row per chunk; a per-camera
// pass adds the chunk rows in a fixed order, so every sum is bitwise reproducible.
//
// What the kernels replace in the reference: the per-observation residual loop
// (/root/reference/bundle_adjustment.py:22-30,49-50), SciPy's finite-difference Jacobian over
// the jac_sparsity pattern (bundle_adjustment.py:53-74,139-141) and the LSMR trust-region step
// (scipy/optimize/_lsq/trf.py:471-499).
#include "sfmba_internal.h"

namespace sfmba {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static int blocks_chunk(const Dev& d) { return cdiv(d.n_chunks, kBlock / 64); }

// Store one block partial of a scalar slot.
__device__ __forceinline__ void put_partial(const Dev& d, int slot, double v) {
    d.scal_part[(size_t)slot * kMaxScalBlocks + blockIdx.x] = v;
}

// ------------------------------------------------------------------------------------------
// fun() parity: residuals in the caller's observation order.
__global__ __launch_bounds__(kBlock) void k_residuals(Dev d, double* __restrict__ out) {
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= d.n_obs) return;
    double4 X = d.X[d.pm_pt[j]];
    Cam k = load_cam(d.cams, d.pm_cam[j]);
    Seed s = project_seed(k, X.x, X.y, X.z);
    double2 uv = d.pm_uv[j];
    double ru, rv;
    residual(s, d.K, uv.x, uv.y, ru, rv);
    size_t o = 2 * (size_t)d.pm_orig[j];
    out[o] = ru;
    out[o + 1] = rv;
}

__global__ __launch_bounds__(kBlock) void k_jacobian_blocks(Dev d, double* __restrict__ jc_out,
                                                            double* __restrict__ jp_out) {
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= d.n_obs) return;
    double4 X = d.X[d.pm_pt[j]];
    Cam k = load_cam(d.cams, d.pm_cam[j]);
    Seed s = project_seed(k, X.x, X.y, X.z);
    double jc[12], a[3], b[3];
    jac_cam(s, d.K, jc);
    jac_pt(s, d.K, k.q, a, b);
    size_t o = (size_t)d.pm_orig[j];
#pragma unroll
    for (int i = 0; i < 12; ++i) jc_out[12 * o + i] = jc[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { jp_out[6 * o + i] = a[i]; jp_out[6 * o + 3 + i] = b[i]; }
}

// ------------------------------------------------------------------------------------------
// A wave tile: a range of point-major observations made of whole points, at most 64 of them
// unless it is one single point with more (then the wave strides over it).  Lane l works on
// observation beg + l, so every stream is read and written fully coalesced, and per-point sums
// are segmented reductions over the lanes of a point.
struct Tile { int beg, end; bool valid, is_long; };
__device__ __forceinline__ Tile wave_tile(const Dev& d) {
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6);
    Tile t;
    t.valid = q < d.n_wtiles;
    t.beg = t.valid ? __builtin_amdgcn_readfirstlane(d.wtile_beg[q]) : 0;
    t.end = t.valid ? __builtin_amdgcn_readfirstlane(d.wtile_beg[q + 1]) : 0;
    t.is_long = t.end - t.beg > 64;
    return t;
}

// Damping of one point: Vinv = (V + lambda * clamp(diag V))^-1 and s0 = Vinv g_p.
__device__ __forceinline__ void damp_point(const Dev& d, int p, const double* Vin, const double* g, double lambda) {
    double V[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) V[i] = Vin[i];
    V[0] += lambda * fmin(fmax(V[0], 1e-6), 1e32);
    V[3] += lambda * fmin(fmax(V[3], 1e-6), 1e32);
    V[5] += lambda * fmin(fmax(V[5], 1e-6), 1e32);
    double inv[6];
    spd3_inverse(V, inv);
    double* o = d.Vinv + 6 * (size_t)p;
#pragma unroll
    for (int i = 0; i < 6; ++i) o[i] = inv[i];
    double4 s;
    s.x = inv[0] * g[0] + inv[1] * g[1] + inv[2] * g[2];
    s.y = inv[1] * g[0] + inv[3] * g[1] + inv[4] * g[2];
    s.z = inv[2] * g[0] + inv[4] * g[1] + inv[5] * g[2];
    s.w = 0.0;
    d.s0[p] = s;
}

// Linearisation, point-major half: V_p and g_p per point, cost and gradient maximum; with kStore
// also the seeds + residuals per observation, which only the matrix-free Schur product reads
// (every other consumer rebuilds them from X and the camera it gathers anyway).
// With lambda > 0 the writer lane also damps its point: Vinv = (V + lambda clamp(diag V))^-1 and
// s0 = Vinv g_p, what k_damp_pm would otherwise read V and g_p back for.
template <bool kStore>
__global__ __launch_bounds__(kBlock) void k_lin_pm(Dev d, double lambda) {
    __shared__ double red[block_reduce_scratch<2>()];
    const int lane = threadIdx.x & 63;
    const Tile t = wave_tile(d);
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double cost = 0.0, gmax = 0.0;
    int key = -1 - lane;
    for (int j = t.beg + lane; j < t.end; j += 64) {
        const int p = d.pm_pt[j];
        key = p;
        const double4 X = d.X[p];
        const Cam k = load_cam(d.cams, d.pm_cam[j]);
        const Seed s = project_seed(k, X.x, X.y, X.z);
        const double2 uv = d.pm_uv[j];
        double ru, rv;
        residual(s, d.K, uv.x, uv.y, ru, rv);
        if (kStore) {
            d.pm_seed[j] = make_double4(s.x, s.y, s.iz, ru);
            d.pm_rv[j] = rv;
        }
        double a[3], b[3];
        jac_pt(s, d.K, k.q, a, b);
        acc[0] += a[0] * a[0] + b[0] * b[0];
        acc[1] += a[0] * a[1] + b[0] * b[1];
        acc[2] += a[0] * a[2] + b[0] * b[2];
        acc[3] += a[1] * a[1] + b[1] * b[1];
        acc[4] += a[1] * a[2] + b[1] * b[2];
        acc[5] += a[2] * a[2] + b[2] * b[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[6 + i] += a[i] * ru + b[i] * rv;
        cost += ru * ru + rv * rv;
    }
    bool writer;
    if (t.is_long) {
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = wave_sum(acc[i]);
        writer = lane == 0;
    } else {
        seg_scan(acc, key, lane);
        const int last = seg_last(key, lane);      // every lane takes part in the shuffles
        writer = key >= 0 && last == lane;
    }
    if (writer) {
        double* Vo = d.V + 6 * (size_t)key;
#pragma unroll
        for (int i = 0; i < 6; ++i) Vo[i] = acc[i];
        double* go = d.gp + 4 * (size_t)key;
        go[0] = acc[6]; go[1] = acc[7]; go[2] = acc[8]; go[3] = 0.0;
        gmax = fmax(fabs(acc[6]), fmax(fabs(acc[7]), fabs(acc[8])));
        if (lambda > 0.0) damp_point(d, key, acc, acc + 6, lambda);
    }
    static_assert(SLOT_GMAX_PT == SLOT_COST + 1, "slot order");
    const double vals[2] = {cost, gmax};
    const double r = block_reduce_n<2>(vals, 2u, red);
    if (threadIdx.x < 2) put_partial(d, SLOT_COST + threadIdx.x, r);
}

// Linearisation, camera-major half: seeds per observation (camera-major copy) and the chunk
// partials of U_c (21 packed upper) and g_c (6).
__global__ __launch_bounds__(kBlock) void k_lin_cm(Dev d) {
    __shared__ __attribute__((aligned(16))) double lds[(kBlock / 64) * 64 * 16];
    const int lane = threadIdx.x & 63;
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6);
    if (q >= d.n_chunks) return;
    const int c = __builtin_amdgcn_readfirstlane(d.chunk_cam[q]);
    const int k0 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q]);
    const int k1 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q + 1]);
    Cam cam = load_cam(d.cams, c);
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    for (int k = k0 + lane; k < k1; k += 64) {
        double4 X = d.X[d.cm_pt[k]];
        Seed s = project_seed(cam, X.x, X.y, X.z);
        double2 uv = d.cm_uv[k];
        double ru, rv;
        residual(s, d.K, uv.x, uv.y, ru, rv);
        d.cm_sx[k] = s.x; d.cm_sy[k] = s.y; d.cm_siz[k] = s.iz;
        double jc[12];
        jac_cam(s, d.K, jc);
        int n = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) acc[n++] += jc[i] * jc[j] + jc[6 + i] * jc[6 + j];
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[21 + i] += jc[i] * ru + jc[6 + i] * rv;
    }
    double tot[2];
    wave_sum_lds<27, 14>(acc, lds + (threadIdx.x >> 6) * (64 * 16), lane, tot);
    if (lane < 14) {
        double* o = d.chunk_part + 27 * (size_t)q;
        o[lane] = tot[0];
        if (lane < 13) o[14 + lane] = tot[1];
    }
}

// out[c][v] = sum over the chunks of camera c of chunk_part[q][v], fixed order.
__global__ __launch_bounds__(kBlock) void k_chunk_reduce(Dev d, int nvals, double* __restrict__ out) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= d.n_cam * nvals) return;
    int c = i / nvals, v = i - c * nvals;
    double s = 0.0;
    for (int q = d.cam_chunk_ptr[c]; q < d.cam_chunk_ptr[c + 1]; ++q)
        s += d.chunk_part[(size_t)q * nvals + v];
    out[i] = s;
}

// ------------------------------------------------------------------------------------------
// Damping, per point: Vinv = (V + lambda * clamp(diag V))^-1 and s0 = Vinv g_p.
__global__ __launch_bounds__(kBlock) void k_damp_pm(Dev d, double lambda) {
    int p = xcd_block(blockIdx.x, gridDim.x) * kBlock + threadIdx.x;
    if (p >= d.n_pts) return;
    damp_point(d, p, d.V + 6 * (size_t)p, d.gp + 4 * (size_t)p, lambda);
}

// Camera-major: chunk partials of T_c = sum W Vinv W' (21 packed upper) and sum W s0 (6),
// W = Jc' Jp (6x3), formed as Jc' (Jp Vinv Jp') Jc without the 6x3 intermediates.
__global__ __launch_bounds__(kBlock) void k_sdiag_cm(Dev d) {
    __shared__ __attribute__((aligned(16))) double lds[(kBlock / 64) * 64 * 16];
    const int lane = threadIdx.x & 63;
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6);
    if (q >= d.n_chunks) return;
    const int c = __builtin_amdgcn_readfirstlane(d.chunk_cam[q]);
    const int k0 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q]);
    const int k1 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q + 1]);
    Cam cam = load_cam(d.cams, c);
    double acc[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) acc[i] = 0.0;
    for (int k = k0 + lane; k < k1; k += 64) {
        const int p = d.cm_pt[k];
        Seed s; s.x = d.cm_sx[k]; s.y = d.cm_sy[k]; s.iz = d.cm_siz[k];
        const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)p);
        double2 v01 = vi[0], v23 = vi[1], v45 = vi[2];
        double4 s0 = d.s0[p];
        double jc[12], a[3], b[3];
        jac_cam(s, d.K, jc);
        jac_pt(s, d.K, cam.q, a, b);
        // M = Jp Vinv Jp' (2x2 symmetric), then Jc' M Jc (upper triangle)
        const double ta0 = v01.x * a[0] + v01.y * a[1] + v23.x * a[2];
        const double ta1 = v01.y * a[0] + v23.y * a[1] + v45.x * a[2];
        const double ta2 = v23.x * a[0] + v45.x * a[1] + v45.y * a[2];
        const double tb0 = v01.x * b[0] + v01.y * b[1] + v23.x * b[2];
        const double tb1 = v01.y * b[0] + v23.y * b[1] + v45.x * b[2];
        const double tb2 = v23.x * b[0] + v45.x * b[1] + v45.y * b[2];
        const double m00 = a[0] * ta0 + a[1] * ta1 + a[2] * ta2;
        const double m01 = a[0] * tb0 + a[1] * tb1 + a[2] * tb2;
        const double m11 = b[0] * tb0 + b[1] * tb1 + b[2] * tb2;
        int n = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const double l0 = jc[i] * m00 + jc[6 + i] * m01;
            const double l1 = jc[i] * m01 + jc[6 + i] * m11;
#pragma unroll
            for (int j = i; j < 6; ++j) acc[n++] += l0 * jc[j] + l1 * jc[6 + j];
        }
        // Jc' (Jp (Vinv g_p)): the right-hand-side contribution
        const double w0 = a[0] * s0.x + a[1] * s0.y + a[2] * s0.z;
        const double w1 = b[0] * s0.x + b[1] * s0.y + b[2] * s0.z;
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[21 + i] += jc[i] * w0 + jc[6 + i] * w1;
    }
    double tot[2];
    wave_sum_lds<27, 14>(acc, lds + (threadIdx.x >> 6) * (64 * 16), lane, tot);
    if (lane < 14) {
        double* o = d.chunk_part + 27 * (size_t)q;
        o[lane] = tot[0];
        if (lane < 13) o[14 + lane] = tot[1];
    }
}

// Per camera: Minv = (U + lambda D - T)^-1, rhs = -g_c + T[21..26]; gradient max.
__global__ __launch_bounds__(kBlock) void k_precond(Dev d, double lambda) {
    __shared__ double red[16];
    int c = blockIdx.x * kBlock + threadIdx.x;
    double gmax = 0.0;
    if (c < d.n_cam) {
        const double* U = d.UG + 27 * (size_t)c;
        const double* T = d.T + 27 * (size_t)c;
        double A[6][6], Ai[6][6], Ud[6];
        int n = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = i; j < 6; ++j) {
                double v = U[n] - T[n];
                if (i == j) { Ud[i] = U[n] + lambda * fmin(fmax(U[n], 1e-6), 1e32); v = Ud[i] - T[n]; }
                A[i][j] = v; A[j][i] = v;
                ++n;
            }
        bool ok = spd6_inverse(A, Ai);
        double* o = d.Minv + 36 * (size_t)c;
        if (ok) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) o[6 * i + j] = Ai[i][j];
        } else {   // fall back to the damped diagonal of U
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) o[6 * i + j] = (i == j) ? 1.0 / Ud[i] : 0.0;
        }
        const double* g = d.UG + 27 * (size_t)c + 21;
        double* r = d.rhs + 6 * (size_t)c;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            r[i] = -g[i] + T[21 + i];
            gmax = fmax(gmax, fabs(g[i]));
        }
    }
    (void)gmax;
    (void)red;
}

// ------------------------------------------------------------------------------------------
// Schur matvec, point-major half: s_p = Vinv_p * sum_o Jp' (Jc x_c).
__global__ __launch_bounds__(kBlock) void k_spmv_pm(Dev d, const double* __restrict__ x, const PcgState* __restrict__ state) {
    if (state && state->done) return;
    const int lane = threadIdx.x & 63;
    const Tile t = wave_tile(d);
    double e[3] = {0, 0, 0};
    int key = -1 - lane;
    for (int j = t.beg + lane; j < t.end; j += 64) {
        key = d.pm_pt[j];
        const int c = d.pm_cam[j];
        const double4 sd = d.pm_seed[j];
        Seed s; s.x = sd.x; s.y = sd.y; s.iz = sd.z;
        const double2* xc = reinterpret_cast<const double2*>(x + 6 * (size_t)c);
        const double2 x01 = xc[0], x23 = xc[1], x45 = xc[2];
        const double dd[6] = {x01.x, x01.y, x23.x, x23.y, x45.x, x45.y};
        double w0, w1;
        jc_times(s, d.K, dd, w0, w1);
        const double2* cq = d.cams + 6 * (size_t)c;
        const double2 q01 = cq[0], q23 = cq[1], q45 = cq[2], q67 = cq[3], q8 = cq[4];
        const double Q[9] = {q01.x, q01.y, q23.x, q23.y, q45.x, q45.y, q67.x, q67.y, q8.x};
        jpT_times_acc(s, d.K, Q, w0, w1, e);
    }
    bool writer;
    if (t.is_long) {
#pragma unroll
        for (int i = 0; i < 3; ++i) e[i] = wave_sum(e[i]);
        writer = lane == 0;
    } else {
        seg_scan(e, key, lane);
        const int last = seg_last(key, lane);      // every lane takes part in the shuffles
        writer = key >= 0 && last == lane;
    }
    if (writer) {
        const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)key);
        const double2 v01 = vi[0], v23 = vi[1], v45 = vi[2];
        double4 sp;
        sp.x = v01.x * e[0] + v01.y * e[1] + v23.x * e[2];
        sp.y = v01.y * e[0] + v23.y * e[1] + v45.x * e[2];
        sp.z = v23.x * e[0] + v45.x * e[1] + v45.y * e[2];
        sp.w = 0.0;
        d.s[key] = sp;
    }
}

// Schur matvec, camera-major half: chunk partials of sum_o Jc' (Jp s_p)  (6 values).
__global__ __launch_bounds__(kBlock) void k_spmv_cm(Dev d, const double4* __restrict__ svec, const PcgState* __restrict__ state) {
    if (state && state->done) return;
    const int lane = threadIdx.x & 63;
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6);
    if (q >= d.n_chunks) return;
    const int c = __builtin_amdgcn_readfirstlane(d.chunk_cam[q]);
    const int k0 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q]);
    const int k1 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q + 1]);
    Cam cam = load_cam(d.cams, c);
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int k = k0 + lane; k < k1; k += 64) {
        Seed s; s.x = d.cm_sx[k]; s.y = d.cm_sy[k]; s.iz = d.cm_siz[k];
        double4 sp = svec[d.cm_pt[k]];
        double dd[3] = {sp.x, sp.y, sp.z};
        double w0, w1;
        jp_times(s, d.K, cam.q, dd, w0, w1);
        jcT_times_acc(s, d.K, w0, w1, acc);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
        double* o = d.chunk_part + 6 * (size_t)q;
#pragma unroll
        for (int i = 0; i < 6; ++i) o[i] = acc[i];
    }
}

// ------------------------------------------------------------------------------------------
// Small dense helpers for the camera-sized vectors (the PCG kernels live in sfmba_pcg.hip).

// y = U* x - qS  (diagnostic: one application of the reduced operator).
__global__ __launch_bounds__(kBlock) void k_schur_apply(Dev d, const double* __restrict__ x,
                                                        double lambda, double* __restrict__ y) {
    int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= d.n_cam) return;
    double p[6], q[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) p[i] = x[6 * (size_t)c + i];
    ustar_times(d.UG + 27 * (size_t)c, lambda, p, q);
#pragma unroll
    for (int i = 0; i < 6; ++i) y[6 * (size_t)c + i] = q[i] - d.qS[6 * (size_t)c + i];
}

// ------------------------------------------------------------------------------------------
// Back-substitution per point: dp = -Vinv (g_p + sum Jp' Jc dc); X_trial = X + dp; the cost of
// the linear model at the step; |dp|^2 and |X|^2; and, in the same pass, the cost at the trial
// point (X_trial with the retracted cameras, which k_cam_update must have written): every lane
// holds its observation, its point's step and gathers its camera anyway.  Seeds and residuals at
// the linearisation point are rebuilt from X and the camera rather than read back.
struct BackObs { Seed s; double ru, rv, w0, w1; Cam k; double2 uv; int c; };
__device__ __forceinline__ BackObs back_load(const Dev& d, const double* __restrict__ dc, int j, const double4& X) {
    BackObs o;
    o.c = d.pm_cam[j];
    o.uv = d.pm_uv[j];
    o.k = load_cam(d.cams, o.c);
    o.s = project_seed(o.k, X.x, X.y, X.z);
    residual(o.s, d.K, o.uv.x, o.uv.y, o.ru, o.rv);
    const double2* xc = reinterpret_cast<const double2*>(dc + 6 * (size_t)o.c);
    const double2 x01 = xc[0], x23 = xc[1], x45 = xc[2];
    const double dd[6] = {x01.x, x01.y, x23.x, x23.y, x45.x, x45.y};
    jc_times(o.s, d.K, dd, o.w0, o.w1);
    return o;
}
// squared residual of observation o at the trial point; non-finite values are counted instead
__device__ __forceinline__ void trial_cost(const Dev& d, const BackObs& o, const Cam& kt, const double4& X, const double* dp,
                                           double& cost, double& bad) {
    const Seed st = project_seed(kt, X.x + dp[0], X.y + dp[1], X.z + dp[2]);
    double ru, rv;
    residual(st, d.K, o.uv.x, o.uv.y, ru, rv);
    const double e = ru * ru + rv * rv;
    if (isfinite(e)) cost += e; else bad += 1.0;
}

__global__ __launch_bounds__(kBlock) void k_back_pm(Dev d, const double* __restrict__ dc,
                                                    double* __restrict__ dp_out) {
    __shared__ double red[block_reduce_scratch<5>()];
    const int lane = threadIdx.x & 63;
    const Tile t = wave_tile(d);
    double model = 0.0, step2 = 0.0, x2 = 0.0, trial = 0.0, bad = 0.0;
    double e[3] = {0, 0, 0};
    int key = -1 - lane;
    BackObs o{};
    double4 X = make_double4(0.0, 0.0, 0.0, 0.0);
    // everything the lane will need after the segmented sum is requested up front, so that the
    // scan does not sit between two rounds of memory latency
    double4 g = make_double4(0.0, 0.0, 0.0, 0.0);
    double2 v01 = make_double2(0.0, 0.0), v23 = v01, v45 = v01;
    Cam kt{};
    for (int j = t.beg + lane; j < t.end; j += 64) {
        key = d.pm_pt[j];
        X = d.X[key];
        if (!t.is_long) {
            g = *reinterpret_cast<const double4*>(d.gp + 4 * (size_t)key);
            const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)key);
            v01 = vi[0]; v23 = vi[1]; v45 = vi[2];
        }
        o = back_load(d, dc, j, X);
        if (!t.is_long) kt = load_cam(d.cams_trial, o.c);
        jpT_times_acc(o.s, d.K, o.k.q, o.w0, o.w1, e);
    }
    bool writer;
    if (t.is_long) {
#pragma unroll
        for (int i = 0; i < 3; ++i) e[i] = wave_sum(e[i]);
        key = __shfl(key, 0, 64);
        writer = lane == 0;
        if (key >= 0) {
            X = d.X[key];
            g = *reinterpret_cast<const double4*>(d.gp + 4 * (size_t)key);
            const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)key);
            v01 = vi[0]; v23 = vi[1]; v45 = vi[2];
        }
    } else {
        seg_scan(e, key, lane);
        const int last = seg_last(key, lane);
#pragma unroll
        for (int i = 0; i < 3; ++i) e[i] = __shfl(e[i], last, 64);
        writer = key >= 0 && last == lane;
    }
    double dp[3] = {0, 0, 0};
    if (key >= 0) {
        const double t0 = e[0] + g.x, t1 = e[1] + g.y, t2 = e[2] + g.z;
        dp[0] = -(v01.x * t0 + v01.y * t1 + v23.x * t2);
        dp[1] = -(v01.y * t0 + v23.y * t1 + v45.x * t2);
        dp[2] = -(v23.x * t0 + v45.x * t1 + v45.y * t2);
    }
    if (writer) {
        d.X_trial[key] = make_double4(X.x + dp[0], X.y + dp[1], X.z + dp[2], 0.0);
        if (dp_out) { dp_out[3 * (size_t)key] = dp[0]; dp_out[3 * (size_t)key + 1] = dp[1]; dp_out[3 * (size_t)key + 2] = dp[2]; }
        step2 = dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
        x2 = X.x * X.x + X.y * X.y + X.z * X.z;
    }
    if (t.is_long) {
        for (int j = t.beg + lane; j < t.end; j += 64) {
            const BackObs b = back_load(d, dc, j, X);
            const Cam kt = load_cam(d.cams_trial, b.c);
            double u0, u1;
            jp_times(b.s, d.K, b.k.q, dp, u0, u1);
            const double l0 = b.ru + b.w0 + u0, l1 = b.rv + b.w1 + u1;
            model += l0 * l0 + l1 * l1;
            trial_cost(d, b, kt, X, dp, trial, bad);
        }
    } else if (key >= 0) {
        double u0, u1;
        jp_times(o.s, d.K, o.k.q, dp, u0, u1);
        const double l0 = o.ru + o.w0 + u0, l1 = o.rv + o.w1 + u1;
        model = l0 * l0 + l1 * l1;
        trial_cost(d, o, kt, X, dp, trial, bad);
    }
    static_assert(SLOT_STEP2_PT == SLOT_MODEL + 1 && SLOT_X2_PT == SLOT_MODEL + 2 && SLOT_TRIAL == SLOT_MODEL + 3 &&
                  SLOT_NONFINITE == SLOT_MODEL + 4, "slot order");
    const double vals[5] = {model, step2, x2, trial, bad};
    const double r = block_reduce_n<5>(vals, 0u, red);
    if (threadIdx.x < 5) put_partial(d, SLOT_MODEL + threadIdx.x, r);
}

// Camera retraction: Q' = exp([dw]x) Q, t' = exp([dw]x) t + dt.
__global__ __launch_bounds__(kBlock) void k_cam_update(Dev d, const double* __restrict__ dc) {
    __shared__ double red[16];
    int c = blockIdx.x * kBlock + threadIdx.x;
    double step2 = 0.0, x2 = 0.0;
    if (c < d.n_cam) {
        Cam k = load_cam(d.cams, c);
        double w[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) { w[i] = dc[6 * (size_t)c + i]; step2 += w[i] * w[i]; }
        double E[9];
        so3_exp(w, E);
        double o[12];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                o[3 * i + j] = E[3 * i] * k.q[j] + E[3 * i + 1] * k.q[3 + j] + E[3 * i + 2] * k.q[6 + j];
            o[9 + i] = E[3 * i] * k.t[0] + E[3 * i + 1] * k.t[1] + E[3 * i + 2] * k.t[2] + w[3 + i];
        }
        double2* dst = d.cams_trial + 6 * (size_t)c;
#pragma unroll
        for (int i = 0; i < 6; ++i) dst[i] = make_double2(o[2 * i], o[2 * i + 1]);
        x2 = 3.0 + k.t[0] * k.t[0] + k.t[1] * k.t[1] + k.t[2] * k.t[2];
    }
    double a = block_sum(step2, red);
    double b = block_sum(x2, red);
    if (threadIdx.x == 0) { put_partial(d, SLOT_STEP2_CAM, a); put_partial(d, SLOT_X2_CAM, b); }
}

// scalars[slot] = sum (or max) of the block partials of that slot; one workgroup per slot.
__global__ __launch_bounds__(1024) void k_scal_finalize(Dev d, int first_slot, int nblocks, unsigned max_mask) {
    __shared__ double red[16];
    const int slot = first_slot + blockIdx.x;
    const bool is_max = (max_mask >> blockIdx.x) & 1u;
    const double* part = d.scal_part + (size_t)slot * kMaxScalBlocks;
    double v = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 1024) v = is_max ? fmax(v, part[i]) : v + part[i];
    double r = is_max ? block_max(v, red) : block_sum(v, red);
    if (threadIdx.x == 0) d.scalars[slot] = r;
}

// Per-camera sum of reprojection errors |projection - observation| at the current estimate,
// camera-major: chunk partials (sum, count).  The per-view quantity the reference prints after
// every bundle adjustment (/root/reference/view.py:103-119, utils.py:337-352).
__global__ __launch_bounds__(kBlock) void k_reproj_cm(Dev d) {
    const int lane = threadIdx.x & 63;
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6);
    if (q >= d.n_chunks) return;
    const int c = __builtin_amdgcn_readfirstlane(d.chunk_cam[q]);
    const int k0 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q]);
    const int k1 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q + 1]);
    const Cam cam = load_cam(d.cams, c);
    double sum = 0.0;
    for (int k = k0 + lane; k < k1; k += 64) {
        const double4 X = d.X[d.cm_pt[k]];
        const Seed s = project_seed(cam, X.x, X.y, X.z);
        const double2 uv = d.cm_uv[k];
        double ru, rv;
        residual(s, d.K, uv.x, uv.y, ru, rv);
        sum += sqrt(ru * ru + rv * rv);
    }
    sum = wave_sum(sum);
    if (lane == 0) {
        d.chunk_part[2 * (size_t)q] = sum;
        d.chunk_part[2 * (size_t)q + 1] = (double)(k1 - k0);
    }
}
void launch_reproj_cm(const Dev& d, double* out2, hipStream_t st) {
    if (d.n_chunks > 0) hipLaunchKernelGGL(k_reproj_cm, dim3(blocks_chunk(d)), dim3(kBlock), 0, st, d);
    launch_chunk_reduce(d, 2, out2, st);
}

// max |g_c| over the cameras (after the all-reduce of UG).
__global__ __launch_bounds__(kBlock) void k_gmax_cam(Dev d) {
    __shared__ double red[16];
    const int i = blockIdx.x * kBlock + threadIdx.x;
    double g = 0.0;
    if (i < d.n_cam * 6) g = fabs(d.UG[27 * (size_t)(i / 6) + 21 + (i % 6)]);
    const double m = block_max(g, red);
    if (threadIdx.x == 0) put_partial(d, SLOT_GMAX_CAM, m);
}
void launch_gmax_cam(const Dev& d, hipStream_t st) {
    const int nb = cdiv(d.n_cam * 6, kBlock);
    hipLaunchKernelGGL(k_gmax_cam, dim3(nb), dim3(kBlock), 0, st, d);
    hipLaunchKernelGGL(k_scal_finalize, dim3(1), dim3(1024), 0, st, d, (int)SLOT_GMAX_CAM, nb, 1u);
}

// a += b (in-process all-reduce of several handles)
__global__ __launch_bounds__(kBlock) void k_vec_add(double* __restrict__ a, const double* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) a[i] += b[i];
}
void launch_vec_add(double* a, const double* b, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_vec_add, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a, b, n);
}

// ------------------------------------------------------------------------------------------
int scal_blocks_pt(const Dev& d) { return cdiv(d.n_pts, kBlock); }
static int blocks_wtile(const Dev& d) { return cdiv(d.n_wtiles, kBlock / 64); }
int blocks_cam(const Dev& d) { return cdiv(d.n_cam, kBlock); }


void launch_residuals(const Dev& d, double* out, hipStream_t st) {
    if (d.n_obs == 0) return;
    hipLaunchKernelGGL(k_residuals, dim3(cdiv(d.n_obs, kBlock)), dim3(kBlock), 0, st, d, out);
}
void launch_jacobian_blocks(const Dev& d, double* jc, double* jp, hipStream_t st) {
    if (d.n_obs == 0) return;
    hipLaunchKernelGGL(k_jacobian_blocks, dim3(cdiv(d.n_obs, kBlock)), dim3(kBlock), 0, st, d, jc, jp);
}
void launch_lin_pm(const Dev& d, bool store_seeds, double lambda, hipStream_t st) {
    const int nb = blocks_wtile(d);
    if (nb > 0) {
        if (store_seeds) hipLaunchKernelGGL(k_lin_pm<true>, dim3(nb), dim3(kBlock), 0, st, d, lambda);
        else hipLaunchKernelGGL(k_lin_pm<false>, dim3(nb), dim3(kBlock), 0, st, d, lambda);
    }
    hipLaunchKernelGGL(k_scal_finalize, dim3(2), dim3(1024), 0, st, d, (int)SLOT_COST, nb, 2u);
}
void launch_lin_cm(const Dev& d, hipStream_t st) {
    if (d.n_chunks == 0) return;
    hipLaunchKernelGGL(k_lin_cm, dim3(blocks_chunk(d)), dim3(kBlock), 0, st, d);
}
void launch_chunk_reduce(const Dev& d, int nvals, double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_chunk_reduce, dim3(cdiv(d.n_cam * nvals, kBlock)), dim3(kBlock), 0, st, d, nvals, out);
}
void launch_damp_pm(const Dev& d, double lambda, hipStream_t st) {
    if (d.n_pts == 0) return;
    hipLaunchKernelGGL(k_damp_pm, dim3(scal_blocks_pt(d)), dim3(kBlock), 0, st, d, lambda);
}
void launch_sdiag_cm(const Dev& d, hipStream_t st) {
    if (d.n_chunks == 0) return;
    hipLaunchKernelGGL(k_sdiag_cm, dim3(blocks_chunk(d)), dim3(kBlock), 0, st, d);
}
void launch_precond(const Dev& d, double lambda, hipStream_t st) {
    hipLaunchKernelGGL(k_precond, dim3(blocks_cam(d)), dim3(kBlock), 0, st, d, lambda);
}
void launch_spmv_pm(const Dev& d, const double* x, const PcgState* state, hipStream_t st) {
    if (d.n_wtiles == 0) return;
    hipLaunchKernelGGL(k_spmv_pm, dim3(blocks_wtile(d)), dim3(kBlock), 0, st, d, x, state);
}
void launch_spmv_cm(const Dev& d, const double4* svec, const PcgState* state, hipStream_t st) {
    if (d.n_chunks == 0) return;
    hipLaunchKernelGGL(k_spmv_cm, dim3(blocks_chunk(d)), dim3(kBlock), 0, st, d, svec, state);
}
void launch_schur_apply(const Dev& d, const double* x, double lambda, double* y, hipStream_t st) {
    hipLaunchKernelGGL(k_schur_apply, dim3(blocks_cam(d)), dim3(kBlock), 0, st, d, x, lambda, y);
}
void launch_back_pm(const Dev& d, const double* dc, double* dp_out, hipStream_t st) {
    const int nb = blocks_wtile(d);
    if (nb > 0) hipLaunchKernelGGL(k_back_pm, dim3(nb), dim3(kBlock), 0, st, d, dc, dp_out);
    hipLaunchKernelGGL(k_scal_finalize, dim3(5), dim3(1024), 0, st, d, (int)SLOT_MODEL, nb, 0u);     // MODEL .. NONFINITE
}
void launch_cam_update(const Dev& d, const double* dc, hipStream_t st) {
    hipLaunchKernelGGL(k_cam_update, dim3(blocks_cam(d)), dim3(kBlock), 0, st, d, dc);
    hipLaunchKernelGGL(k_scal_finalize, dim3(2), dim3(1024), 0, st, d, (int)SLOT_STEP2_CAM, blocks_cam(d), 0u);
}

}  // namespace sfmba
