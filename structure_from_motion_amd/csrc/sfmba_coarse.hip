// Coarse space of the two-level PCG preconditioner (gfx950, fp64).
//
// Block-Jacobi alone leaves the error components that move whole groups of cameras together: the
// reduced camera matrix S is nearly singular along the seven gauge directions (translation,
// rotation and scale of the whole scene), and restricted to a group of neighbouring cameras those
// directions are the slowest modes of the iteration.  The two-level preconditioner adds a
// correction on exactly that space,
//
//     M^-1 = M_J^-1 + Z E^-1 Z',        E = Z' S Z,
//
// where Z has seven columns per cluster of consecutive cameras: the change of every camera of the
// cluster under a unit translation (3), rotation (3) and scaling (1) of the world, written in the
// camera tangent (dw, dt, ...) of the left perturbation Y' = exp([dw]x) Y + dt that the Jacobian
// blocks use.  For a camera with rotation Q and translation t:
//     world translation e_a :  dw = 0,          dt = -Q e_a
//     world rotation    e_k :  dw = -Q e_k,     dt = (Q e_k) x t
//     world scaling         :  dw = 0,          dt = t
// (intrinsic entries of the tangent, where the camera model has any, are zero, and so is the whole
// of Z for a camera nobody observes).  E is small
// (7 x clusters) and dense; it is assembled from the block-CSR matrix once per damping, scaled to
// unit diagonal and inverted in place by a blocked Gauss-Jordan sweep, one launch per 32 columns.
// Each PCG iteration then costs one more small launch: the prolongation z += Z E^-1 (Z' r), which
// also takes over the r'z partials from k_pcg_update; the restriction Z' r is formed by k_pcg_update
// itself (per workgroup and cluster) and by a kernel of its own only for the first residual of a solve.
// All sums run in a fixed order, so the result is the same from run to run.
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "sfmba_internal.h"

namespace sfmba {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

constexpr int kVecs = kCoarseVecs;            // coarse vectors per cluster
constexpr int kSlotsOf7 = kBlock / kVecs;     // 36 groups of 7 threads in a workgroup
constexpr int kTile = 32;                     // Gauss-Jordan block size

// Z of every camera, [n_cam][7][ND]; thread = camera.
template <class M>
__global__ __launch_bounds__(kBlock) void k_coarse_basis(Dev d, Coarse k) {
    constexpr int ND = M::ND;
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= d.n_cam) return;
    const auto cam = M::load(d.cams, c);
    double z[kVecs * ND];
#pragma unroll
    for (int i = 0; i < kVecs * ND; ++i) z[i] = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double u0 = cam.q[a], u1 = cam.q[3 + a], u2 = cam.q[6 + a];     // Q e_a
        z[a * ND + 3] = -u0; z[a * ND + 4] = -u1; z[a * ND + 5] = -u2;
        z[(3 + a) * ND + 0] = -u0; z[(3 + a) * ND + 1] = -u1; z[(3 + a) * ND + 2] = -u2;
        z[(3 + a) * ND + 3] = u1 * cam.t[2] - u2 * cam.t[1];
        z[(3 + a) * ND + 4] = u2 * cam.t[0] - u0 * cam.t[2];
        z[(3 + a) * ND + 5] = u0 * cam.t[1] - u1 * cam.t[0];
    }
    z[6 * ND + 3] = cam.t[0]; z[6 * ND + 4] = cam.t[1]; z[6 * ND + 5] = cam.t[2];
    // a camera without observations (on any rank: U is the all-ranks sum) has a zero right-hand
    // side and must not move: it takes no part in the coarse space (the flag is left by the finishing
    // pass of the damping, so that this kernel reads nothing the next linearisation rewrites)
    const bool seen = d.cam_seen[c] != 0;
    double* out = k.Z + kVecs * ND * (size_t)c;
#pragma unroll
    for (int i = 0; i < kVecs * ND; ++i) out[i] = seen ? z[i] : 0.0;
}

// E(I, J) = sum over the blocks (r, c) of S with r in cluster I, c in cluster J of Z_r' S_rc Z_c.
// The blocks of a cluster pair are cut into chunks; one workgroup per chunk: a group of 7 threads
// takes every 36th block of the chunk (thread a of the group forms row a of the 7 x 7 product),
// the 36 partial products are added in order into the chunk's partial, and k_coarse_collect adds
// the chunk partials of every pair in order.
template <int ND>
__global__ __launch_bounds__(kBlock) void k_coarse_assemble(Dev d, Coarse k) {
    constexpr int NN = ND * ND;
    __shared__ double part[kSlotsOf7][kVecs * kVecs];
    const int p = blockIdx.x;
    const int slot = threadIdx.x / kVecs, a = threadIdx.x - kVecs * slot;
    if (slot < kSlotsOf7) {
        double acc[kVecs];
#pragma unroll
        for (int j = 0; j < kVecs; ++j) acc[j] = 0.0;
        const int e1 = k.cchunk_beg[p + 1];
        for (int idx = k.cchunk_beg[p] + slot; idx < e1; idx += kSlotsOf7) {
            const int b = k.cpair_blk[idx], r = k.cpair_row[idx], c = d.bsr_col[b];
            const double* __restrict__ S = d.Sfull + NN * (size_t)b;
            const double* __restrict__ zr = k.Z + (kVecs * (size_t)r + a) * ND;
            const double* __restrict__ zc = k.Z + kVecs * ND * (size_t)c;
            double u[ND];
#pragma unroll
            for (int j = 0; j < ND; ++j) u[j] = 0.0;
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                const double zi = zr[i];
#pragma unroll
                for (int j = 0; j < ND; ++j) u[j] += zi * S[ND * i + j];
            }
#pragma unroll
            for (int a2 = 0; a2 < kVecs; ++a2) {
                double g = 0.0;
#pragma unroll
                for (int j = 0; j < ND; ++j) g += u[j] * zc[a2 * ND + j];
                acc[a2] += g;
            }
        }
#pragma unroll
        for (int a2 = 0; a2 < kVecs; ++a2) part[slot][kVecs * a + a2] = acc[a2];
    }
    __syncthreads();
    if (threadIdx.x < kVecs * kVecs) {
        double s = 0.0;
        for (int q = 0; q < kSlotsOf7; ++q) s += part[q][threadIdx.x];
        k.cpart[kVecs * kVecs * (size_t)p + threadIdx.x] = s;
    }
}

// thread = (cluster pair, entry of its 7 x 7 block)
__global__ __launch_bounds__(kBlock) void k_coarse_collect(Coarse k, double* __restrict__ E) {
    const int t = blockIdx.x * kBlock + threadIdx.x;
    const int p = t / (kVecs * kVecs), e = t - kVecs * kVecs * p;
    if (p >= k.n_cpairs) return;
    double s = 0.0;
    for (int q = k.cpair_cptr[p]; q < k.cpair_cptr[p + 1]; ++q) s += k.cpart[kVecs * kVecs * (size_t)q + e];
    const int ra = e / kVecs, ca = e - kVecs * ra;
    E[(size_t)(kVecs * k.cpair_i[p] + ra) * k.n_pad + kVecs * k.cpair_j[p] + ca] = s;
}

// dst = D src D with D = diag(src)^-1/2 (unit diagonal), identity on the padding and wherever a
// diagonal entry is not positive (that coarse vector then drops out); D goes to k.scale.
__global__ __launch_bounds__(kBlock) void k_coarse_scale(Coarse k, const double* __restrict__ src, double* __restrict__ dst) {
    const size_t idx = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t total = (size_t)k.n_pad * k.n_pad;
    if (idx >= total) return;
    const int i = (int)(idx / k.n_pad), j = (int)(idx - (size_t)i * k.n_pad);
    double si = 0.0, sj = 0.0;
    if (i < k.n) { const double v = src[(size_t)i * k.n_pad + i]; if (v > 0.0 && v < 1e300) si = 1.0 / sqrt(v); }
    if (j < k.n) { const double v = src[(size_t)j * k.n_pad + j]; if (v > 0.0 && v < 1e300) sj = 1.0 / sqrt(v); }
    double v = (si != 0.0 && sj != 0.0) ? src[idx] * si * sj : 0.0;
    if (i == j && si == 0.0) v = 1.0;
    dst[idx] = v;
    if (j == 0) k.scale[i] = si;
}

// One block column of the Gauss-Jordan sweep that turns a matrix into its inverse: with pivot
// block p and P = A_pp^-1,
//     A_pp <- P,   A_pj <- P A_pj,   A_ip <- -A_ip P,   A_ij <- A_ij - A_ip P A_pj    (i, j != p).
// One workgroup per 32 x 32 tile; it reads `src` and writes `dst`, so no tile sees a value of this
// step.  Every workgroup inverts the pivot block itself (in LDS, unpivoted: the matrix is
// symmetric positive definite with unit diagonal), which is cheaper than a launch of its own.
// kThreads = 64 (SFMBA_GJ_WAVES=1) is the same sweep with one wavefront per tile, sixteen entries per
// thread, built to see whether a one-wavefront workgroup finds room beside k_schur_pairs, under which
// a sweep of the four-wavefront form waits for that kernel's whole grid.  It does not: the sweep
// still took the 150 us of the kernel beside it, and 42 instead of 20 us alone (Venice shape 2.29
// against 2.12 ms per solve).  The default stays the four-wavefront form.
template <int kThreads>
__global__ __launch_bounds__(kThreads) void k_coarse_gj(const double* __restrict__ src, double* __restrict__ dst, int n_pad, int p) {
    __shared__ double P[kTile][kTile + 1], A[kTile][kTile + 1], B[kTile][kTile + 1], T[kTile][kTile + 1];
    const int bi = blockIdx.y, bj = blockIdx.x;
    constexpr int kTy = kThreads / kTile, kM = kTile / kTy, kUnroll = kM > 4 ? 2 : 8;                  // rows of 32 threads; entries per thread
    const int tx = threadIdx.x & (kTile - 1), ty = threadIdx.x / kTile;
    const bool row_p = bi == p, col_p = bj == p;
#pragma unroll
    for (int m = 0; m < kM; ++m) {
        const int i = ty + kTy * m;
        if (!row_p) A[i][tx] = src[(size_t)(kTile * bi + i) * n_pad + kTile * p + tx];
        if (!col_p) B[i][tx] = src[(size_t)(kTile * p + i) * n_pad + kTile * bj + tx];
    }
    // the pivot block: every thread keeps its four entries (rows ty + 8 m, column tx) in registers;
    // a step publishes the pivot row and column through LDS buffers that alternate with the
    // step's parity, so one barrier per step is enough
    __shared__ double rowbuf[2][kTile], colbuf[2][kTile];
    double pv[kM];
#pragma unroll
    for (int m = 0; m < kM; ++m) pv[m] = src[(size_t)(kTile * p + ty + kTy * m) * n_pad + kTile * p + tx];
#pragma unroll
    for (int kk = 0; kk < kTile; ++kk) {
        const int par = kk & 1;
        if (ty == kk % kTy) rowbuf[par][tx] = pv[kk / kTy];
        if (tx == kk) {
#pragma unroll
            for (int m = 0; m < kM; ++m) colbuf[par][ty + kTy * m] = pv[m];
        }
        __syncthreads();
        const double pd = rowbuf[par][kk];
        const double piv = (pd > 1e-300 && pd < 1e300) ? 1.0 / pd : 0.0;
        const double rowk = rowbuf[par][tx];
#pragma unroll
        for (int m = 0; m < kM; ++m) {
            const int i = ty + kTy * m;
            const double colv = colbuf[par][i];
            double v;
            if (i == kk) v = tx == kk ? piv : rowk * piv;
            else v = tx == kk ? -colv * piv : pv[m] - colv * piv * rowk;
            pv[m] = v;
        }
    }
#pragma unroll
    for (int m = 0; m < kM; ++m) P[ty + kTy * m][tx] = pv[m];
    __syncthreads();
    double out[kM];
    if (row_p && col_p) {
#pragma unroll
        for (int m = 0; m < kM; ++m) out[m] = P[ty + kTy * m][tx];
    } else if (row_p) {
        double s[kM];
#pragma unroll
            for (int m = 0; m < kM; ++m) s[m] = 0.0;
#pragma unroll kUnroll
        for (int q = 0; q < kTile; ++q) {
            const double b = B[q][tx];
#pragma unroll
            for (int m = 0; m < kM; ++m) s[m] += P[ty + kTy * m][q] * b;
        }
#pragma unroll
        for (int m = 0; m < kM; ++m) out[m] = s[m];
    } else {
        {
            double s[kM];
#pragma unroll
            for (int m = 0; m < kM; ++m) s[m] = 0.0;
#pragma unroll kUnroll
            for (int q = 0; q < kTile; ++q) {
                const double b = P[q][tx];
#pragma unroll
                for (int m = 0; m < kM; ++m) s[m] += A[ty + kTy * m][q] * b;
            }
#pragma unroll
            for (int m = 0; m < kM; ++m) { T[ty + kTy * m][tx] = s[m]; out[m] = -s[m]; }
        }
        if (!col_p) {
            __syncthreads();
            double s[kM];
#pragma unroll
            for (int m = 0; m < kM; ++m) s[m] = src[(size_t)(kTile * bi + ty + kTy * m) * n_pad + kTile * bj + tx];
#pragma unroll kUnroll
            for (int q = 0; q < kTile; ++q) {
                const double b = B[q][tx];
#pragma unroll
                for (int m = 0; m < kM; ++m) s[m] -= T[ty + kTy * m][q] * b;
            }
#pragma unroll
            for (int m = 0; m < kM; ++m) out[m] = s[m];
        }
    }
#pragma unroll
    for (int m = 0; m < kM; ++m) dst[(size_t)(kTile * bi + ty + kTy * m) * n_pad + kTile * bj + tx] = out[m];
}

// rc = D Z' r: one workgroup per cluster; a group of 7 threads takes every 36th camera.
template <int ND>
__global__ __launch_bounds__(kBlock) void k_coarse_restrict(Dev d, Coarse k, int state) {
    __shared__ double part[kSlotsOf7][kVecs];
    if (d.pcg[state].done) return;
    const int I = blockIdx.x;
    const int slot = threadIdx.x / kVecs, a = threadIdx.x - kVecs * slot;
    if (slot < kSlotsOf7) {
        double acc = 0.0;
        const int hi = k.cl_ptr[I + 1];
        // four cameras of a thread at a time, their loads requested together (clamped, masked)
        for (int c = k.cl_ptr[I] + slot; c < hi; c += 4 * kSlotsOf7) {
            double s[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cc = min(c + u * kSlotsOf7, hi - 1);
                const double* __restrict__ z = k.Z + (kVecs * (size_t)cc + a) * ND;
                const double* __restrict__ r = d.pr + ND * (size_t)cc;
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < ND; ++i) t += z[i] * r[i];
                s[u] = t;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += c + u * kSlotsOf7 < hi ? s[u] : 0.0;
        }
        part[slot][a] = acc;
    }
    __syncthreads();
    if (threadIdx.x < kVecs) {
        double s = 0.0;
        for (int q = 0; q < kSlotsOf7; ++q) s += part[q][threadIdx.x];
        const int g = kVecs * I + threadIdx.x;
        k.rc[g] = k.scale[g] * s;
    }
}

// z += Z D Einv rc and the r'z partials; thread = (camera, row) as in k_pcg_update.  A workgroup's
// cameras lie in at most two clusters (a cluster is never smaller than a workgroup's share of
// cameras): it forms the up to 14 entries of y = D Einv rc it needs itself, one wavefront per row.
// kFromParts: rc is put together here from what k_pcg_update left per workgroup and cluster slot
// (a cluster's cameras span a few consecutive workgroups of that kernel; it is the second cluster of
// the first of them unless it begins where that workgroup does, and the first of all the others).
template <int ND, bool kFromParts>
__global__ __launch_bounds__(kBlock) void k_coarse_prolong(Dev d, Coarse k, int state) {
    constexpr int kCams = kBlock / ND;
    __shared__ double red[16];
    __shared__ double y[2 * kVecs];
    if (d.pcg[state].done) return;
    // neighbouring workgroups need the same rows of Einv: keep them on one die's L2
    const int blk = xcd_block(blockIdx.x, gridDim.x);
    const int c0 = blk * kCams;
    const int c1 = min(c0 + kCams, d.n_cam) - 1;
    const int I0 = k.cam_cl[c0];
    const int nrows = min(kVecs * (k.cam_cl[c1] - I0 + 1), 2 * kVecs);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    {
        // y = D Einv rc for the workgroup's (up to 14) rows: thread t takes columns t, t + 256, ...
        // of every row, so that all loads of a round are independent and requested together; the
        // sums of a row meet in a wavefront reduction and four LDS words
        constexpr int kRows = 2 * kVecs;
        __shared__ double ypart[kBlock / 64][kRows];
        const double* __restrict__ e0 = k.Einv + (size_t)(kVecs * I0) * k.n_pad;
        double s[kRows];
#pragma unroll
        for (int q = 0; q < kRows; ++q) s[q] = 0.0;
        for (int j = threadIdx.x; j < k.n_pad; j += 2 * kBlock) {
            // (n_pad is a multiple of 32, not of the workgroup: the second column is clamped and masked)
            const int j2 = min(j + kBlock, k.n_pad - 1);
            double rj, rj2;
            if constexpr (kFromParts) {
                auto restricted = [&](int col) {
                    if (col >= k.n) return 0.0;
                    const int I = col / kVecs, a = col - kVecs * I;
                    const int lo = k.cl_ptr[I], hi = k.cl_ptr[I + 1];
                    const int w0 = lo / kCams, w1 = (hi - 1) / kCams;
                    double t = k.rpart[2 * kVecs * (size_t)w0 + (lo - kCams * w0 != 0 ? kVecs : 0) + a];
                    for (int w = w0 + 1; w <= w1; ++w) t += k.rpart[2 * kVecs * (size_t)w + a];
                    return k.scale[col] * t;
                };
                rj = restricted(j);
                rj2 = j + kBlock < k.n_pad ? restricted(j2) : 0.0;
            } else {
                rj = k.rc[j];
                rj2 = j + kBlock < k.n_pad ? k.rc[j2] : 0.0;
            }
            double e[kRows], e2[kRows];
#pragma unroll
            for (int q = 0; q < kRows; ++q) {
                const double* __restrict__ row = e0 + (size_t)(q < nrows ? q : 0) * k.n_pad;
                e[q] = row[j];
                e2[q] = row[j2];
            }
#pragma unroll
            for (int q = 0; q < kRows; ++q) s[q] += e[q] * rj + e2[q] * rj2;
        }
#pragma unroll
        for (int q = 0; q < kRows; ++q) {
            const double t = wave_sum(s[q]);
            if (lane == 0) ypart[w][q] = t;
        }
        __syncthreads();
        if (threadIdx.x < nrows) {
            double t = ypart[0][threadIdx.x];
#pragma unroll
            for (int v = 1; v < kBlock / 64; ++v) t += ypart[v][threadIdx.x];
            y[threadIdx.x] = k.scale[kVecs * I0 + threadIdx.x] * t;
        }
    }
    __syncthreads();
    const int local = threadIdx.x / ND, kk = threadIdx.x - ND * local;
    const int c = c0 + local;
    double dot = 0.0;
    if (local < kCams && c < d.n_cam) {
        const size_t o = ND * (size_t)c + kk;
        const double* __restrict__ zc = k.Z + kVecs * ND * (size_t)c + kk;
        const double* yy = y + kVecs * (k.cam_cl[c] - I0);
        double z = d.pz[o];
#pragma unroll
        for (int a = 0; a < kVecs; ++a) z += zc[a * ND] * yy[a];
        d.pz[o] = z;
        dot = d.pr[o] * z;
    }
    const double s = block_sum(dot, red);
    if (threadIdx.x == 0) d.dot_part[d.dot_off + blk] = s;
}

// ---- launch wrappers --------------------------------------------------------------------

// Builds Z, E = Z'SZ and its inverse; returns the buffer that holds D^-1 E^-1 D^-1 (k.scale = D).
// `assembled`, when given, is recorded once nothing of the problem (cameras, UG, the matrix) is
// read any more: what follows works on the coarse matrices alone.
// `sum_over_ranks`, when given, is called on E once it is assembled (on `st`): with the reduced matrix
// split over the ranks E = Z'SZ is the sum of the ranks' Z'S_k Z.
const double* launch_coarse_setup(const Dev& d, const Coarse& k, hipStream_t st, hipEvent_t assembled,
                                  void (*sum_over_ranks)(void*, double*, size_t), void* ctx) {
    const dim3 blk(kBlock);
    static const bool trace = getenv("SFMBA_TRACE_COARSE_HOST") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    const auto h0 = now();
    if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_coarse_basis<Bal>), dim3(cdiv(d.n_cam, kBlock)), blk, 0, st, d, k);
    else hipLaunchKernelGGL((k_coarse_basis<Pinhole>), dim3(cdiv(d.n_cam, kBlock)), blk, 0, st, d, k);
    const size_t total = (size_t)k.n_pad * k.n_pad;
    const auto h1 = now();
    (void)hipMemsetAsync(k.E[0], 0, total * sizeof(double), st);
    const auto h2 = now();
    if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_coarse_assemble<Bal::ND>), dim3(k.n_cchunks), blk, 0, st, d, k);
    else hipLaunchKernelGGL((k_coarse_assemble<Pinhole::ND>), dim3(k.n_cchunks), blk, 0, st, d, k);
    hipLaunchKernelGGL(k_coarse_collect, dim3(cdiv(k.n_cpairs * kVecs * kVecs, kBlock)), blk, 0, st, k, k.E[0]);
    if (sum_over_ranks) sum_over_ranks(ctx, k.E[0], total);
    const auto h3 = now();
    if (assembled) (void)hipEventRecord(assembled, st);
    const auto h4 = now();
    hipLaunchKernelGGL(k_coarse_scale, dim3((unsigned)((total + kBlock - 1) / kBlock)), blk, 0, st, k, k.E[0], k.E[1]);
    int src = 1;
    const int nt = k.n_pad / kTile;
    static const int gj_waves = [] { const char* e = getenv("SFMBA_GJ_WAVES"); return e ? atoi(e) : 4; }();     // tuning: 1 = one wavefront per tile
    for (int p = 0; p < nt; ++p) {
        if (gj_waves == 1) hipLaunchKernelGGL((k_coarse_gj<64>), dim3(nt, nt), dim3(64), 0, st, k.E[src], k.E[src ^ 1], k.n_pad, p);
        else hipLaunchKernelGGL((k_coarse_gj<kBlock>), dim3(nt, nt), blk, 0, st, k.E[src], k.E[src ^ 1], k.n_pad, p);
        src ^= 1;
    }
    if (trace) {
        auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        fprintf(stderr, "coarse set-up, host time: basis %.1f, memset %.1f, assemble + collect %.1f, event %.1f, scale + %d sweeps %.1f us\n",
                us(h0, h1), us(h1, h2), us(h2, h3), us(h3, h4), nt, us(h4, now()));
    }
    return k.E[src];
}

// z = M_J^-1 r is in pz and r in pr: adds the coarse correction and rewrites the r'z partials.
// `state` names the PcgState whose `done` flag gates the work.
void launch_coarse_apply(const Dev& d, const Coarse& k, int state, hipStream_t st, bool from_parts) {
    const dim3 blk(kBlock), grid(pcg_update_blocks(d));
    if (from_parts) {
        if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_coarse_prolong<Bal::ND, true>), grid, blk, 0, st, d, k, state);
        else hipLaunchKernelGGL((k_coarse_prolong<Pinhole::ND, true>), grid, blk, 0, st, d, k, state);
    } else if (d.model == MODEL_BAL) {
        hipLaunchKernelGGL((k_coarse_restrict<Bal::ND>), dim3(k.n_cl), blk, 0, st, d, k, state);
        hipLaunchKernelGGL((k_coarse_prolong<Bal::ND, false>), grid, blk, 0, st, d, k, state);
    } else {
        hipLaunchKernelGGL((k_coarse_restrict<Pinhole::ND>), dim3(k.n_cl), blk, 0, st, d, k, state);
        hipLaunchKernelGGL((k_coarse_prolong<Pinhole::ND, false>), grid, blk, 0, st, d, k, state);
    }
}

}  // namespace sfmba
