// Small dense fp64 Cholesky solve of the reduced camera system in ONE launch.
//
// For the camera counts the reference's own workloads have (11 views, a ring of 47: 66 to 282
// unknowns) the reduced system is best solved exactly: SciPy's trust-region step is a minimum-norm
// Gauss-Newton step (scipy/optimize/_lsq/trf.py:479-480), and an inexact, damped step walks the
// flat valleys of two-view tracks several times slower.  The chain of dependent launches of
// sfmba_dense.hip costs 0.3 ms at that size; this kernel does the same work in one launch.
//
// One workgroup per block row of 32 rows; workgroup i produces the tiles L(i, 0..i) of its row
// left to right (left-looking):
//     L(i,j)  = (A(i,j) - sum_k<j L(i,k) L(j,k)') Linv(j,j)'        j < i
//     L(i,i)  = chol(A(i,i) - sum_k<i L(i,k) L(i,k)'),  Linv(i,i) = L(i,i)^-1
// with the products on v_mfma_f64_16x16x4_f64, the 32 x 32 factorisation and the inverse of the
// triangle by one wavefront in registers.  Forward substitution rides along (y_i = Linv(i,i)
// (b_i - sum_j L(i,j) y_j)); the back substitution x_i = Linv(i,i)' (y_i - sum_j>i L(j,i)' x_j)
// follows in the same launch, last block row first.
//
// Synchronisation is by data flow, as in the single-launch PCG (sfmba_pcg.hip): everything one
// workgroup reads of another - tiles, y, x - lives in a buffer whose words hold a sentinel (all
// bits set) until written, moves through agent-scope relaxed loads and stores, and consumers poll
// exactly the words they need.  Workgroup i only ever waits for workgroups j < i during the
// factorisation and j > i during the back substitution (which all have finished their
// factorisation by then), and workgroups start in index order, so the launch cannot wait on
// itself; a poll that exceeds its time limit raises the abort word, every poller checks it, and
// the caller takes the multi-launch path.
#include <algorithm>

#include "sfmba_internal.h"

namespace sfmba {

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NB = 32;
constexpr int LDT = NB + 1;
constexpr int LDR = NB + 2;
constexpr int kTile = NB * NB;
constexpr unsigned long long kSentinel = ~0ull;

__device__ __forceinline__ double ld_shared(const double* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_shared(double* p, double v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long ld_word(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_word(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool is_sentinel(double v) {
    return (unsigned long long)__double_as_longlong(v) == kSentinel;
}

struct Poll {
    unsigned long long* abort_word;
    long long deadline;
    // true (for whoever asks) once the launch is to be given up
    __device__ __forceinline__ bool expired(int spins) const {
        if ((spins & 15) != 0) return false;
        if (ld_word(abort_word) != kSentinel) return true;
        if ((long long)wall_clock64() > deadline) { st_word(abort_word, 0ull); return true; }
        return false;
    }
};

__host__ __device__ __forceinline__ size_t tile_at(int i, int j) { return ((size_t)i * (i + 1) / 2 + j) * kTile; }

// acc(16 x 16 quadrant) += sum over the tiles k < cnt of RowI(k) RowJ(k)', operands polled straight
// from the tile buffer (row-major 32 x 32 tiles).  Lane l feeds A[i = l & 15][k = l >> 4] and
// B[k = l >> 4][j = l & 15]; two tiles (16 + 16 loads) are in flight at a time.  Returns false
// (wave-uniform) when the launch is given up.
__device__ __forceinline__ bool history(const double* __restrict__ rowI, const double* __restrict__ rowJ, int cnt,
                                        int wr, int wc, int lane, const Poll& poll, v4d& acc) {
    const int ra = (16 * wr + (lane & 15)) * NB + (lane >> 4);
    const int rb = (16 * wc + (lane & 15)) * NB + (lane >> 4);
    for (int k = 0; k < cnt; ++k) {
        const double* pa = rowI + (size_t)k * kTile + ra;
        const double* pb = rowJ + (size_t)k * kTile + rb;
        double a[8], b[8];
        for (int spins = 1;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int m = 0; m < 8; ++m) { a[m] = ld_shared(pa + 4 * m); b[m] = ld_shared(pb + 4 * m); }
#pragma unroll
            for (int m = 0; m < 8; ++m) ok = ok && !is_sentinel(a[m]) && !is_sentinel(b[m]);
            if (__all(ok)) break;
            if (__any(poll.expired(spins))) return false;
        }
#pragma unroll
        for (int m = 0; m < 8; ++m) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[m], acc, 0, 0, 0);
    }
    return true;
}

// out(quadrant) = sum_k P[r][k] Q[c][k] for two 32 x 32 LDS tiles with padded rows.
__device__ __forceinline__ v4d tile_abt(const double* P, const double* Q, int wr, int wc, int lane, v4d acc) {
    const double* pa = P + (16 * wr + (lane & 15)) * LDT + (lane >> 4);
    const double* pb = Q + (16 * wc + (lane & 15)) * LDT + (lane >> 4);
#pragma unroll
    for (int k0 = 0; k0 < NB; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0], acc, 0, 0, 0);
    return acc;
}

struct CholArgs {
    const double* A;        // [n][n] row-major, lower triangle read
    const double* b;        // [n_real]
    double* x;              // [n_real]
    double* buf;            // tiles (lower block triangle, tile-major) | y [n] | x [n] | control words
    int* fail;              // 1: a pivot was not positive, 2: given up (time limit)
    double* fail_slot;      // the same as a double into a scalar slot (device, and pinned host), may be null
    double* fail_host;
    int n, n_real, nb;
    long long timeout_ticks;
    long long* trace;       // [nb][8] time stamps of the phases of every block row (measurements), or null
};

__device__ __forceinline__ void report(const CholArgs& a, int code) {
    *a.fail = code;
    if (a.fail_slot) *a.fail_slot = (double)code;
    if (a.fail_host) *a.fail_host = (double)code;
}
__device__ __forceinline__ void give_up(const CholArgs& a, int tid) { if (tid == 0) report(a, 2); }

__global__ __launch_bounds__(256) void k_chol_flow(CholArgs a) {
    __shared__ double T[NB * LDT];        // the tile in progress
    __shared__ double Lj[NB * LDT];       // Linv(j,j) of the column in progress
    __shared__ double Mi[NB * LDT];       // Linv(i,i)
    __shared__ double red[8 * NB];
    __shared__ __attribute__((aligned(16))) double Lcol[NB * LDR];     // L(i,i) transposed, columns 16-byte aligned
    __shared__ __attribute__((aligned(16))) double colbuf[2 * NB];
    __shared__ double yj[NB], bacc[NB], invd[NB], zs[NB], xj[NB];
    __shared__ int s_bad;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int i = blockIdx.x, n = a.n, nb = a.nb;
    double* tiles = a.buf;
    double* yv = a.buf + tile_at(nb, 0);
    double* xv = yv + n;
    unsigned long long* ctl = reinterpret_cast<unsigned long long*>(xv + n);
    Poll poll{ctl, (long long)wall_clock64() + a.timeout_ticks};
    double* myrow = tiles + tile_at(i, 0);
    auto stamp = [&](int k) { if (a.trace && tid == 0) { a.trace[8 * i + k] = (long long)wall_clock64(); a.trace[8 * a.nb + 8 * i + k] = (long long)clock64(); } };
    stamp(0);

    if (tid < NB) { const int r = NB * i + tid; bacc[tid] = r < a.n_real ? a.b[r] : 0.0; }
    if (tid == 0) s_bad = 0;
    v4d dacc = {0.0, 0.0, 0.0, 0.0};          // running sum_k L(i,k) L(i,k)' for the diagonal tile
    __syncthreads();

    for (int j = 0; j < i; ++j) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        bool alive = history(myrow, tiles + tile_at(j, 0), j, wr, wc, lane, poll, acc);
        // T = A(i,j) - history
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * wr + (lane >> 4) + 4 * g, c = 16 * wc + (lane & 15);
            T[r * LDT + c] = a.A[(size_t)(NB * i + r) * n + NB * j + c] - acc[g];
        }
        // Linv(j,j) and y_j, polled by the whole workgroup
        const double* src = tiles + tile_at(j, j);
        if (__syncthreads_or(!alive)) { give_up(a, tid); return; }
        for (int spins = 1;; ++spins) {
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ld_shared(src + tid + 256 * q);
            const double yy = tid < NB ? ld_shared(yv + NB * j + tid) : 0.0;
            bool ok = !is_sentinel(yy);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ok = ok && !is_sentinel(v[q]);
                const int e = tid + 256 * q;
                Lj[(e >> 5) * LDT + (e & 31)] = v[q];
            }
            if (tid < NB) yj[tid] = yy;
            if (__syncthreads_and(ok)) break;
            // (the time limit is looked at every 16th round, by everybody, and decided together)
            if ((spins & 15) == 0 && __syncthreads_or(poll.expired(spins))) { give_up(a, tid); return; }
        }
        // L(i,j) = T Linv(j,j)'
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
        const v4d out = tile_abt(T, Lj, wr, wc, lane, zero);
        __syncthreads();
        double* dst = myrow + (size_t)j * kTile;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int r = 16 * wr + (lane >> 4) + 4 * g, c = 16 * wc + (lane & 15);
            st_shared(dst + r * NB + c, out[g]);
            T[r * LDT + c] = out[g];
        }
        __syncthreads();
        dacc = tile_abt(T, T, wr, wc, lane, dacc);
        {   // forward substitution: b_i -= L(i,j) y_j ; thread = (row tid / 8, 4 columns)
            const int r = tid >> 3, part = tid & 7;
            double s = 0.0;
#pragma unroll
            for (int c = 4 * part; c < 4 * part + 4; ++c) s += T[r * LDT + c] * yj[c];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            if (part == 0) bacc[r] -= s;
        }
        __syncthreads();
    }

    // diagonal tile
    stamp(1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = 16 * wr + (lane >> 4) + 4 * g, c = 16 * wc + (lane & 15);
        const int gr = NB * i + r, gc = NB * i + c;
        // (rows beyond the real system: identity)
        const double v = gr < a.n_real && gc < a.n_real ? a.A[(size_t)gr * n + gc] : (gr == gc ? 1.0 : 0.0);
        T[r * LDT + c] = v - dacc[g];
    }
    __syncthreads();
    if (w == 0) {
        // Cholesky of the 32 x 32 tile by one wavefront: lane r (and r + 32, which mirrors it) holds
        // row r of the trailing matrix in registers.  Per column c the lanes put their entry of
        // column c - by symmetry row c - into LDS, everybody reads the pivot and the entries right of
        // it from there (one LDS round trip per column, all reads at wave-uniform addresses), and
        // S[r][j] -= S[r][c] S[j][c] / S[c][c] for j > c.  Column c of L goes to LDS transposed.
        const int r = lane & 31;
        double row[NB];
#pragma unroll
        for (int c = 0; c < NB; ++c) row[c] = T[r * LDT + c];
        bool bad = false;
#pragma unroll
        for (int c = 0; c < NB; ++c) {
            double* col = colbuf + NB * (c & 1);
            col[r] = row[c];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            double piv = col[c];
            if (!(piv > 0.0)) { bad = true; piv = 1.0; }
            double rinv = __builtin_amdgcn_rsq(piv);
            rinv = fma(0.5 * rinv, fma(-piv * rinv, rinv, 1.0), rinv);
            rinv = fma(0.5 * rinv, fma(-piv * rinv, rinv, 1.0), rinv);
            const double f = row[c] * (rinv * rinv);
#pragma unroll
            for (int jj = c + 1; jj < NB; ++jj) row[jj] = fma(-f, col[jj], row[jj]);
            Lcol[c * LDR + r] = row[c] * rinv;         // (rows above the diagonal: never read)
            if (lane == 0) invd[c] = rinv;
        }
        if (bad && lane == 0) s_bad = 1;
        stamp(2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // inverse of the triangle: lane c solves L m = e_c column by column of L (the updates of one
        // step are independent of each other); rows above the diagonal come out as exact zeros
        {
            const int c = lane & 31;
            double v[NB];
#pragma unroll
            for (int rr = 0; rr < NB; ++rr) v[rr] = rr == c ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const double mk = v[k] * invd[k];
                Mi[k * LDT + c] = mk;
#pragma unroll
                for (int rr = k + 1; rr < NB; ++rr) v[rr] = fma(-Lcol[k * LDR + rr], mk, v[rr]);
            }
        }
    }
    __syncthreads();
    stamp(3);
    {   // publish Linv(i,i) and y_i = Linv(i,i) b_i
        double* dst = myrow + (size_t)i * kTile;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            st_shared(dst + e, Mi[(e >> 5) * LDT + (e & 31)]);
        }
        if (tid < NB) {
            double v = 0.0;
            for (int k = 0; k <= tid; ++k) v += Mi[tid * LDT + k] * bacc[k];
            st_shared(yv + NB * i + tid, v);
            zs[tid] = v;
        }
        if (tid == 0 && s_bad) report(a, 1);
    }
    __syncthreads();
    stamp(4);

    // back substitution: z = y_i - sum_j>i L(j,i)' x_j, last block row first
    for (int j = nb - 1; j > i; --j) {
        const double* src = tiles + tile_at(j, i);
        for (int spins = 1;; ++spins) {
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ld_shared(src + tid + 256 * q);
            const double xx = tid < NB ? ld_shared(xv + NB * j + tid) : 0.0;
            bool ok = !is_sentinel(xx);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ok = ok && !is_sentinel(v[q]);
                const int e = tid + 256 * q;
                T[(e >> 5) * LDT + (e & 31)] = v[q];
            }
            if (tid < NB) xj[tid] = xx;
            if (__syncthreads_and(ok)) break;
            if ((spins & 15) == 0 && __syncthreads_or(poll.expired(spins))) { give_up(a, tid); return; }
        }
        {   // thread = (column tid % 32, rows 4 (tid / 32) ..+3)
            const int c = tid & 31, part = tid >> 5;
            double s = 0.0;
#pragma unroll
            for (int r = 4 * part; r < 4 * part + 4; ++r) s += T[r * LDT + c] * xj[r];
            red[part * NB + c] = s;
        }
        __syncthreads();
        if (tid < NB) {
            double s = 0.0;
#pragma unroll
            for (int p8 = 0; p8 < 8; ++p8) s += red[p8 * NB + tid];
            zs[tid] -= s;
        }
        __syncthreads();
    }
    stamp(5);
    if (tid < NB) {
        double v = 0.0;
        for (int r = tid; r < NB; ++r) v += Mi[r * LDT + tid] * zs[r];
        st_shared(xv + NB * i + tid, v);
        if (NB * i + tid < a.n_real) a.x[NB * i + tid] = v;
    }
    stamp(6);
}

// Everything the factorisation launch needs, in one pass: the lower triangle of the dense matrix
// gathered from the block-CSR matrix (blocks of a row are sorted by column: binary search, absent
// blocks are zero), the exchange buffer armed with the sentinel, the failure words cleared.
__global__ __launch_bounds__(256) void k_chol_prepare(Dev d, int nd, double* __restrict__ A, int n, int n_real,
                                                      unsigned long long* __restrict__ buf, size_t words,
                                                      int* __restrict__ fail, double* __restrict__ fail_slot) {
    const size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (t0 == 0) { *fail = 0; if (fail_slot) *fail_slot = 0.0; }
    for (size_t t = t0; t < words; t += stride) buf[t] = kSentinel;
    for (size_t t = t0; t < (size_t)n * n; t += stride) {
        const int row = (int)(t / n), col = (int)(t - (size_t)row * n);
        if (col > row) continue;
        double v = 0.0;
        if (row < n_real) {
            const int r = row / nd, c = col / nd;
            int lo = d.bsr_ptr[r], hi = d.bsr_ptr[r + 1];
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (d.bsr_col[mid] < c) lo = mid + 1; else hi = mid;
            }
            if (lo < d.bsr_ptr[r + 1] && d.bsr_col[lo] == c)
                v = d.Sfull[(size_t)lo * nd * nd + (row - nd * r) * nd + (col - nd * c)];
        }
        A[t] = v;
    }
}

}  // namespace

size_t chol_flow_words(int n) {
    const int nb = n / NB;
    return tile_at(nb, 0) + 2 * (size_t)n + 8;
}

// Solves S x = b for the reduced matrix in d.Sfull (block-CSR) through the dense work array A
// (leading dimension n = n_real rounded up to a multiple of 32).  buf holds chol_flow_words(n)
// doubles.  *fail, and fail_slot / fail_host when given (a scalar slot on the device and the same
// slot in pinned host memory): 0 fine, 1 a pivot was not positive, 2 the launch was given up.
// Two launches; nothing is waited for.
hipError_t launch_chol_flow(const Dev& d, double* A, int n, int n_real, const double* b, double* x, double* buf, int* fail,
                            double* fail_slot, double* fail_host, hipStream_t st, long long* trace) {
    CholArgs a{};
    a.trace = trace;
    a.A = A; a.b = b; a.x = x; a.buf = buf; a.fail = fail; a.fail_slot = fail_slot; a.fail_host = fail_host;
    a.n = n; a.n_real = n_real; a.nb = n / NB;
    a.timeout_ticks = 20000000LL;       // 0.2 s of the 100 MHz wall clock
    if (fail_host) *fail_host = 0.0;    // (the kernel writes it only on failure, the host reads it after its wait)
    const size_t words = chol_flow_words(n);
    const int blocks = (int)std::min<size_t>(1024, ((size_t)n * n + 255) / 256);
    hipLaunchKernelGGL(k_chol_prepare, dim3(blocks), dim3(256), 0, st, d, cam_dim(d), A, n, n_real,
                       reinterpret_cast<unsigned long long*>(buf), words, fail, fail_slot);
    hipLaunchKernelGGL(k_chol_flow, dim3(a.nb), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace sfmba
