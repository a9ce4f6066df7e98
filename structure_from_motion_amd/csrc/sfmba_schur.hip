// Explicit block-sparse Schur complement of libsfmba (gfx950, fp64).
//
// S = (U + lambda D_c) - sum_p W_p (V_p + lambda D_p)^-1 W_p' has an ND x ND block (ND = 6 or 9
// tangent numbers per camera, by camera model) for every pair of
// cameras that share a point.  The set-up lists, per upper block (r <= c), the point of every pair of distinct
// observations (a, b) of one point with camera(a) = r and camera(b) = c (an observation paired
// with itself belongs to the diagonal block and is summed by the camera-major pass k_sdiag_cm), sorted by block and cut
// into chunks; one wavefront owns one chunk, both cameras are wave-uniform, every lane forms
//     Jc_a' (Jp_a Vinv_p Jp_b') Jc_b          (ND x 2 . 2x2 . 2 x ND)
// for its pairs from the point and the two cameras, sums in registers, and the wavefront reduces the ND^2 sums
// through an LDS tile into one partial row per chunk.  A per-block pass adds the chunk rows
// in fixed order (bitwise reproducible, no atomics), a finishing pass mirrors the upper blocks
// into a full block-CSR matrix and adds the damped U blocks on the diagonal.  The reduced system
// is then small (Venice shape: 41 k blocks, 12 MB, L2/MALL resident) and PCG runs on it with a
// block-CSR matvec; no observation is touched inside the PCG loop.
//
// This is the camera-camera accumulation that replaces the LSMR products J v / J' u of the
// reference's solver (scipy/optimize/_lsq/trf.py:479-480, scipy/sparse/linalg/_isolve/lsmr.py).
#include <algorithm>

#include "sfmba_internal.h"

namespace sfmba {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// A load that writes LDS directly: lane l's kBytes land at dst + l * kBytes (dst the same in every lane).
// Rows [I0, I1) of the block sums of pair chunk q.
//
// A pair's operands are its point and the point's damped inverse block; both observations'
// seeds are rebuilt from the point and the two wave-uniform cameras (derivatives only: the
// reciprocal of the depth by two Newton steps).  Reading the seeds the camera-major pass stored
// instead (two more indices per pair, six 8-byte gathers) kept the vector caches busier than the
// arithmetic this adds keeps the vector unit: 195 -> 155 us per launch at the Venice shape.
template <class M, int I0, int I1>
__device__ __forceinline__ void schur_pairs_rows(const Dev& d, int q, int lane, double* lds) {
    constexpr int ND = M::ND, NN = (I1 - I0) * ND;
    const int4 rec = d.pchunk_rec[q];
    const int k0 = __builtin_amdgcn_readfirstlane(rec.x);
    const int k1 = __builtin_amdgcn_readfirstlane(rec.y);
    const int r = __builtin_amdgcn_readfirstlane(rec.z);
    const int c = __builtin_amdgcn_readfirstlane(rec.w);
    // both cameras are the same for the whole wavefront: they live in scalar registers
    const typename M::Cam camr = M::uniform(M::load(d.cams, r));
    const typename M::Cam camc = M::uniform(M::load(d.cams, c));
    double acc[NN];
#pragma unroll
    for (int i = 0; i < NN; ++i) acc[i] = 0.0;
    struct Gathered { double2 x01, x2, v01, v23, v45; };
    auto gather = [&](int p) {
        Gathered g;
        const double2* xp = reinterpret_cast<const double2*>(d.X + p);
        const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)p);
        g.x01 = xp[0]; g.x2 = xp[1];
        g.v01 = vi[0]; g.v23 = vi[1]; g.v45 = vi[2];
        return g;
    };
    auto accumulate = [&](const Gathered& g) {
        const Seed sa = M::seed_jac(camr, g.x01.x, g.x01.y, g.x2.x), sb = M::seed_jac(camc, g.x01.x, g.x01.y, g.x2.x);
        const double2 v01 = g.v01, v23 = g.v23, v45 = g.v45;
        const typename M::Obs oa = M::obs(sa, camr, d.K), ob = M::obs(sb, camc, d.K);
        double aa[3], ab[3], ba[3], bb[3];
        M::jac_pt(oa, camr.q, aa, ab);     // rows of Jp_a
        M::jac_pt(ob, camc.q, ba, bb);     // rows of Jp_b
        // t0 = Vinv * ba, t1 = Vinv * bb  (Vinv packed upper: 00 01 02 11 12 22)
        const double t00 = v01.x * ba[0] + v01.y * ba[1] + v23.x * ba[2];
        const double t01 = v01.y * ba[0] + v23.y * ba[1] + v45.x * ba[2];
        const double t02 = v23.x * ba[0] + v45.x * ba[1] + v45.y * ba[2];
        const double t10 = v01.x * bb[0] + v01.y * bb[1] + v23.x * bb[2];
        const double t11 = v01.y * bb[0] + v23.y * bb[1] + v45.x * bb[2];
        const double t12 = v23.x * bb[0] + v45.x * bb[1] + v45.y * bb[2];
        // M = Jp_a Vinv Jp_b'  (2x2)
        const double m00 = aa[0] * t00 + aa[1] * t01 + aa[2] * t02;
        const double m01 = aa[0] * t10 + aa[1] * t11 + aa[2] * t12;
        const double m10 = ab[0] * t00 + ab[1] * t01 + ab[2] * t02;
        const double m11 = ab[0] * t10 + ab[1] * t11 + ab[2] * t12;
        double jca[2 * ND], jcb[2 * ND];
        M::jac_cam(oa, jca);
        M::jac_cam(ob, jcb);
        // (with the pinhole model column 3 of a camera block, the x translation, has no v entry and
        // column 4 no u entry: products with those zeros are left out)
        constexpr bool kZ = M::kDiagonalPixelMap;
#pragma unroll
        for (int i = I0; i < I1; ++i) {
            const double l0 = kZ && i == 3 ? jca[i] * m00 : kZ && i == 4 ? jca[ND + i] * m10 : jca[i] * m00 + jca[ND + i] * m10;
            const double l1 = kZ && i == 3 ? jca[i] * m01 : kZ && i == 4 ? jca[ND + i] * m11 : jca[i] * m01 + jca[ND + i] * m11;
#pragma unroll
            for (int j = 0; j < ND; ++j)
                acc[ND * (i - I0) + j] += kZ && j == 3 ? l0 * jcb[j] : kZ && j == 4 ? l1 * jcb[ND + j] : l0 * jcb[j] + l1 * jcb[ND + j];
        }
    };
    // The point index is requested a round ahead (6 x 6 blocks, three wavefronts per SIMD) or two
    // rounds ahead with the operands one round ahead (9 x 9 blocks: this round's arithmetic
    // overlaps the next round's gathers, 530 -> 440 us per launch at the Venice shape; with 6 x 6
    // blocks the registers of that pipeline cost the third wavefront and 8 %).
    constexpr bool kAhead = ND > 6;
    const int kf = k0 + lane;
    if constexpr (kAhead) {
        int pn = 0;
        Gathered gn{};
        if (kf < k1) gn = gather(d.pair_p[kf]);
        if (kf + 64 < k1) pn = d.pair_p[kf + 64];
        for (int k = kf; k < k1; k += 64) {
            const Gathered g = gn;
            const int p = pn;
            if (k + 128 < k1) pn = d.pair_p[k + 128];
            if (k + 64 < k1) gn = gather(p);
            accumulate(g);
        }
    } else {
        int pn = kf < k1 ? d.pair_p[kf] : 0;
        for (int k = kf; k < k1; k += 64) {
            const int p = pn;
            if (k + 64 < k1) pn = d.pair_p[k + 64];
            accumulate(gather(p));
        }
    }
    constexpr int kPasses = (NN + 17) / 18;
    double tot[kPasses];
    wave_sum_lds<NN, 18>(acc, lds, lane, tot);
    if (lane < 18) {
        double* o = d.pair_part + ND * ND * (size_t)q + ND * I0;
#pragma unroll
        for (int p = 0; p < kPasses; ++p)
            if (18 * p + lane < NN) o[18 * p + lane] = tot[p];
    }
}

// kSplit = 2: blockIdx.y picks the upper or lower half of the block's rows, so that a lane carries
// half of the running sums (the 81 sums of a 9 x 9 block leave one wavefront per SIMD, and the
// kernel waits on its gathers) at the price of forming the pair's 2 x 2 core twice: two wavefronts
// per SIMD, 570 -> 530 us per launch at the Venice shape.  (Three thirds still need 236 registers
// and ran at 750 us; a 168-register budget spills 69 of them.)
//
// kThreads: a workgroup's wavefronts own chunks of very different lengths and its registers and LDS
// return to the compute unit only when the last one has finished, so small workgroups keep more
// wavefronts resident (6 x 6: 2.0 -> 2.4 per SIMD on average with one wavefront per workgroup,
// 155 -> 145 us; 9 x 9: two per workgroup, 384 -> 362 us).
template <class M, int kSplit, int kThreads>
__global__ __launch_bounds__(kThreads) void k_schur_pairs(Dev d) {
    constexpr int ND = M::ND;
    __shared__ __attribute__((aligned(16))) double lds_all[(kThreads / 64) * 64 * 20];
    double* lds = lds_all + (threadIdx.x >> 6) * (64 * 20);
    const int lane = threadIdx.x & 63;
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kThreads / 64) + (threadIdx.x >> 6);
    if (q >= d.n_pchunks) return;
    if constexpr (kSplit == 1) {
        schur_pairs_rows<M, 0, ND>(d, q, lane, lds);
    } else {
        constexpr int kHalf = (ND + 1) / 2;
        if (blockIdx.y == 0) schur_pairs_rows<M, 0, kHalf>(d, q, lane, lds);
        else schur_pairs_rows<M, kHalf, ND>(d, q, lane, lds);
    }
}

// The chunk table as one 16-byte record per chunk (set-up).
__global__ __launch_bounds__(kBlock) void k_pchunk_records(Dev d, int4* __restrict__ rec) {
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= d.n_pchunks) return;
    const int blk = d.pchunk_blk[q];
    rec[q] = make_int4(d.pchunk_beg[q], d.pchunk_beg[q + 1], d.ublk_r[blk], d.ublk_c[blk]);
}

// Bsum[u][e] = sum over the chunks of upper block u, fixed order.
__global__ __launch_bounds__(kBlock) void k_block_reduce(Dev d, int nn) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= d.n_ublk * nn) return;
    const int u = i / nn, e = i - nn * u;
    double s = 0.0;
    for (int q = d.ublk_chunk_ptr[u]; q < d.ublk_chunk_ptr[u + 1]; ++q) s += d.pair_part[nn * (size_t)q + e];
    d.Bsum[i] = s;
}

// Finishing passes over the (all-ranks) upper sums B: S_rc = -B, S_cr = -B', S_cc = U*_c - B_cc.
// With the matrix split by points over the ranks, B holds this rank's off-diagonal sums and the
// all-ranks diagonal ones: the stored diagonal block is `share` = 1 / nranks of the true one, so
// that the ranks' matrices add up to S, while the preconditioner block is the inverse of the true one.
//
// k_finish_diag, one thread per camera: the diagonal block of the full block-CSR matrix
// (U*_c - T_c - B_cc with T_c from the camera-major pass k_sdiag_cm), its
// Cholesky factor, the inverse (block-Jacobi preconditioner), the reduced right-hand side
// rhs = -g_c + sum W Vinv g_p and, for the single-launch PCG, the scaling F = L^-1, bhat = F rhs,
// the scaled diagonal block F S_cc F' and the armed (all bits set) exchange buffers.
//
// A camera's sums and blocks are rows of 27 to 81 doubles, so every load and store of a thread
// touches a cache line of its own: what bounds the kernel is the number of line accesses one
// compute unit's vector cache serves per cycle.  One-wavefront workgroups in which only the first
// kDiagCams lanes own a camera spread the cameras over all compute units (1 778 cameras: 112
// workgroups instead of 7; 22 -> 7 us per launch).
constexpr int kDiagCams = 16;
template <class M>
__global__ __launch_bounds__(64) void k_finish_diag(Dev d, double lambda, const double* __restrict__ tpart, FusedPcg f, double share) {
    constexpr int ND = M::ND, NU = M::NU, NV = M::NV, NN = ND * ND;
    if (f.F != nullptr) {
        // arm the exchange buffers of the single-launch PCG: every word all bits set
        unsigned long long* vw = reinterpret_cast<unsigned long long*>(f.vec);
        unsigned long long* dw = reinterpret_cast<unsigned long long*>(f.dots);
        const size_t nv = (size_t)kPcgSlots * d.n_cam * kPcgVecRec, nd = (size_t)kPcgRecWords * kPcgSlots * f.blocks + 1;
        const size_t t = blockIdx.x * 64 + threadIdx.x, nt = (size_t)gridDim.x * 64;
        for (size_t i = t; i < nv; i += nt) vw[i] = ~0ull;
        for (size_t i = t; i < nd; i += nt) dw[i] = ~0ull;
    }
    const int c = blockIdx.x * kDiagCams + threadIdx.x;
    if (threadIdx.x >= kDiagCams || c >= d.n_cam) return;
    const double* U = d.UG + NV * (size_t)c;
    const double* T = tpart + NV * (size_t)c;          // packed upper sum W Vinv W' | sum W Vinv g_p
    const double* B = d.Bsum + NN * (size_t)d.ublk_diag[c];   // pairs of two observations of one point in this camera
    double S[ND][ND], A[ND][ND], Li[ND][ND], Ai[ND][ND];
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const int a = i < j ? i : j, b = i < j ? j : i;
            double uv = U[upidx(ND, a, b)];
            if (i == j) uv += lambda * fmin(fmax(uv, 1e-6), 1e32);
            S[i][j] = uv - T[upidx(ND, a, b)] - B[ND * i + j];
            Li[i][j] = 0.0;
        }
    {
        bool seen = false;
#pragma unroll
        for (int i = 0; i < ND; ++i) seen = seen || U[upidx(ND, i, i)] != 0.0;
        d.cam_seen[c] = seen ? 1 : 0;
    }
    double* So = d.Sfull + NN * (size_t)d.bsr_diag[c];
    double dg[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        dg[i] = S[i][i];
#pragma unroll
        for (int j = 0; j < ND; ++j) { So[ND * i + j] = share * S[i][j]; A[i][j] = 0.5 * (S[i][j] + S[j][i]); }
    }
    const bool ok = spd_inverse(A, Ai, Li);
    if (!ok) {
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                Ai[i][j] = (i == j) ? 1.0 / fmax(fabs(dg[i]), 1e-300) : 0.0;
                Li[i][j] = (i == j) ? 1.0 / sqrt(fmax(fabs(dg[i]), 1e-300)) : 0.0;
            }
    }
    double* Mo = d.Minv + NN * (size_t)c;
#pragma unroll
    for (int i = 0; i < ND; ++i)
#pragma unroll
        for (int j = 0; j < ND; ++j) Mo[ND * i + j] = Ai[i][j];
    const double* g = U + NU;
    double r[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        r[i] = -g[i] + T[NU + i];
        d.rhs[ND * (size_t)c + i] = r[i];
    }
    if (f.F == nullptr) return;
    double* Fo = const_cast<double*>(f.F) + NN * (size_t)c;
    double* bo = const_cast<double*>(f.bhat) + ND * (size_t)c;
    double* Ho = const_cast<double*>(f.Shat) + NN * (size_t)d.bsr_diag[c];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        double v = 0.0;
        double fs[ND];                // row i of F S
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const double lij = j <= i ? Li[i][j] : 0.0;
            Fo[ND * i + j] = lij;
            v += lij * r[j];
            double t = 0.0;
#pragma unroll
            for (int k = 0; k <= i; ++k) t += Li[i][k] * S[k][j];
            fs[j] = t;
        }
        bo[i] = v;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            double h = 0.0;
#pragma unroll
            for (int l = 0; l <= j; ++l) h += fs[l] * Li[j][l];
            Ho[ND * i + j] = h;
        }
    }
}

// k_finish_offdiag, one thread per element of every upper off-diagonal block: both orientations
// of the block in the full matrix and, for the single-launch PCG, of the scaled block
// F_r S_rc F_c' (its transpose is the scaled S_cr).
// A workgroup takes kBlock / NN whole blocks: a thread's own element of B goes through LDS to the
// others of its block, t = B F_c' is formed once per block (NN short sums instead of one per
// thread and row), and every thread finishes its element of F_r t.  (One thread per element that
// read all of B and both rows of F itself: 16 us per launch at the Venice shape, now 11.)
template <int ND>
__global__ __launch_bounds__(kBlock) void k_finish_offdiag(Dev d, FusedPcg f) {
    constexpr int NN = ND * ND, kPer = kBlock / NN;
    __shared__ double sB[kPer][NN], sT[kPer][NN];
    const int lb = threadIdx.x / NN, e = threadIdx.x - NN * lb, i = e / ND, j = e - ND * i;
    const int u = blockIdx.x * kPer + lb;
    const bool act = lb < kPer && u < d.n_ublk;
    int r = 0, c = 0;
    if (act) {
        r = d.ublk_r[u]; c = d.ublk_c[u];
        sB[lb][e] = d.Bsum[NN * (size_t)u + e];
    }
    __syncthreads();
    const bool off = act && r != c;
    size_t pos = 0, pos_t = 0;
    if (off) {
        pos = NN * (size_t)d.ublk_pos[u] + e;
        pos_t = NN * (size_t)d.ublk_pos_t[u] + ND * j + i;
        const double v = -sB[lb][e];
        d.Sfull[pos] = v;
        d.Sfull[pos_t] = v;
    }
    if (f.F == nullptr) return;
    if (off) {
        // t[i][j] = sum_l B[i][l] F_c[j][l]
        const double* Fj = f.F + NN * (size_t)c + ND * j;
        double tk = 0.0;
#pragma unroll
        for (int l = 0; l < ND; ++l) tk += sB[lb][ND * i + l] * Fj[l];
        sT[lb][e] = tk;
    }
    __syncthreads();
    if (!off) return;
    const double* Fi = f.F + NN * (size_t)r + ND * i;
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < ND; ++k) s += Fi[k] * sT[lb][ND * k + j];
    double* H = const_cast<double*>(f.Shat);
    H[pos] = -s;
    H[pos_t] = -s;
}

// y = S x on the block-CSR matrix, one thread per (camera, row of the block); 42 (28) cameras per
// workgroup.  Diagnostic entry (sfmba_schur_matvec); PCG uses k_pcg_spmv_bsr.
template <int ND>
__global__ __launch_bounds__(kBlock) void k_bsr_spmv(Dev d, const double* __restrict__ x, double* __restrict__ y) {
    constexpr int kCams = kBlock / ND, NN = ND * ND;
    const int local = threadIdx.x / ND, k = threadIdx.x - ND * local;
    const int c = blockIdx.x * kCams + local;
    if (local >= kCams || c >= d.n_cam) return;
    double acc = 0.0;
    const int b0 = d.bsr_ptr[c], b1 = d.bsr_ptr[c + 1];
    for (int b = b0; b < b1; ++b) {
        const double* s = d.Sfull + NN * (size_t)b + ND * k;
        const double* xv = x + ND * (size_t)d.bsr_col[b];
        if constexpr (ND % 2 == 0) {
            const double2* s2 = reinterpret_cast<const double2*>(s);
            const double2* x2 = reinterpret_cast<const double2*>(xv);
#pragma unroll
            for (int i = 0; i < ND / 2; ++i) { const double2 a = s2[i], v = x2[i]; acc += a.x * v.x + a.y * v.y; }
        } else {
#pragma unroll
            for (int i = 0; i < ND; ++i) acc += s[i] * xv[i];
        }
    }
    y[ND * (size_t)c + k] = acc;
}

// Dense lower-triangular copy of S (row-major n x n, n = ND n_cam) for the Cholesky solver.
__global__ __launch_bounds__(kBlock) void k_bsr_to_dense(Dev d, double* __restrict__ A, int n, int nd) {
    const int nn = nd * nd;
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= d.nnzb * nn) return;
    const int b = t / nn, e = t - nn * b, i = e / nd, j = e - nd * i;
    // row camera of block b: binary search in bsr_ptr
    int lo = 0, hi = d.n_cam;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (d.bsr_ptr[mid] <= b) lo = mid; else hi = mid;
    }
    const int row = nd * lo + i, col = nd * d.bsr_col[b] + j;
    A[(size_t)row * n + col] = d.Sfull[t];
}

int bsr_blocks(const Dev& d) { return cdiv(d.n_cam, kBlock / cam_dim(d)); }


void launch_schur_pairs(const Dev& d, hipStream_t st) {
    if (d.n_pchunks == 0) return;
    if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_schur_pairs<Bal, 2, 128>), dim3(cdiv(d.n_pchunks, 2), 2), dim3(128), 0, st, d);
    else hipLaunchKernelGGL((k_schur_pairs<Pinhole, 1, 64>), dim3(d.n_pchunks), dim3(64), 0, st, d);
}
void launch_pchunk_records(const Dev& d, int4* rec, hipStream_t st) {
    hipLaunchKernelGGL(k_pchunk_records, dim3(cdiv(d.n_pchunks, kBlock)), dim3(kBlock), 0, st, d, rec);
}
void launch_block_reduce(const Dev& d, hipStream_t st) {
    const int nn = cam_dim(d) * cam_dim(d);
    if (d.n_ublk > 0)
        hipLaunchKernelGGL(k_block_reduce, dim3(cdiv(d.n_ublk * nn, kBlock)), dim3(kBlock), 0, st, d, nn);
}
// The per-camera numbers a split matrix still sums over the ranks, in one buffer: the diagonal
// blocks of B (scattered in Bsum) followed by the camera-major sums `tpart`.
__global__ __launch_bounds__(kBlock) void k_diag_collect(Dev d, int nn, int nv, const double* __restrict__ tpart, double* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t nb = (size_t)d.n_cam * nn, nt = (size_t)d.n_cam * nv;
    if (i < nb) {
        const int c = (int)(i / nn), e = (int)(i - (size_t)nn * c);
        out[i] = d.Bsum[(size_t)nn * d.ublk_diag[c] + e];
    } else if (i < nb + nt) {
        out[i] = tpart[i - nb];
    }
}
__global__ __launch_bounds__(kBlock) void k_diag_spread(Dev d, int nn, int nv, const double* __restrict__ in, double* __restrict__ tpart) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    const size_t nb = (size_t)d.n_cam * nn, nt = (size_t)d.n_cam * nv;
    if (i < nb) {
        const int c = (int)(i / nn), e = (int)(i - (size_t)nn * c);
        d.Bsum[(size_t)nn * d.ublk_diag[c] + e] = in[i];
    } else if (i < nb + nt) {
        tpart[i - nb] = in[i];
    }
}
size_t diag_collect_words(const Dev& d, int nv) { return (size_t)d.n_cam * (cam_dim(d) * cam_dim(d) + nv); }
void launch_diag_collect(const Dev& d, int nv, const double* tpart, double* out, hipStream_t st) {
    const int nn = cam_dim(d) * cam_dim(d);
    hipLaunchKernelGGL(k_diag_collect, dim3((unsigned)((diag_collect_words(d, nv) + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, d, nn, nv, tpart, out);
}
void launch_diag_spread(const Dev& d, int nv, const double* in, double* tpart, hipStream_t st) {
    const int nn = cam_dim(d) * cam_dim(d);
    hipLaunchKernelGGL(k_diag_spread, dim3((unsigned)((diag_collect_words(d, nv) + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, d, nn, nv, in, tpart);
}

void launch_schur_finish(const Dev& d, double lambda, const double* tpart, const FusedPcg* scaled, hipStream_t st, double share) {
    FusedPcg f{};
    if (scaled) f = *scaled;
    const int nn = cam_dim(d) * cam_dim(d);
    if (d.model == MODEL_BAL) {
        hipLaunchKernelGGL((k_finish_diag<Bal>), dim3(cdiv(d.n_cam, kDiagCams)), dim3(64), 0, st, d, lambda, tpart, f, share);
        if (d.n_ublk > 0)
            hipLaunchKernelGGL((k_finish_offdiag<Bal::ND>), dim3(cdiv(d.n_ublk, kBlock / nn)), dim3(kBlock), 0, st, d, f);
    } else {
        hipLaunchKernelGGL((k_finish_diag<Pinhole>), dim3(cdiv(d.n_cam, kDiagCams)), dim3(64), 0, st, d, lambda, tpart, f, share);
        if (d.n_ublk > 0)
            hipLaunchKernelGGL((k_finish_offdiag<Pinhole::ND>), dim3(cdiv(d.n_ublk, kBlock / nn)), dim3(kBlock), 0, st, d, f);
    }
}
void launch_bsr_spmv(const Dev& d, const double* x, double* y, hipStream_t st) {
    if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_bsr_spmv<Bal::ND>), dim3(bsr_blocks(d)), dim3(kBlock), 0, st, d, x, y);
    else hipLaunchKernelGGL((k_bsr_spmv<Pinhole::ND>), dim3(bsr_blocks(d)), dim3(kBlock), 0, st, d, x, y);
}
void launch_bsr_to_dense(const Dev& d, double* A, int n, hipStream_t st) {
    const int nd = cam_dim(d);
    hipLaunchKernelGGL(k_bsr_to_dense, dim3(cdiv(d.nnzb * nd * nd, kBlock)), dim3(kBlock), 0, st, d, A, n, nd);
}

}  // namespace sfmba
