// Dense fp64 Cholesky solve of the reduced camera system on the gfx950 matrix cores.
//
// Used when the reduced system is small and dense (6 n_cameras <= dense_max_dim): the block-CSR
// matrix is scattered into a dense row-major array padded to a multiple of 32 and factored by a
// left-looking blocked Cholesky with 32x32 blocks.  Per block column j:
//   k_chol_diag   (1 workgroup)   T = A_jj - L_j L_j' with v_mfma_f64_16x16x4_f64, Cholesky of T
//                                 in LDS, inverse triangle Linv_jj, and y_j = Linv_jj b_j
//                                 (b_j is final by then: forward substitution rides along)
//   k_chol_panel  (1 workgroup per 32-row tile below)   L_ij = (A_ij - L_i L_j') Linv_jj' with the
//                                 same MFMA, then b_i -= L_ij y_j
// and after the factorisation one light launch per block column, last to first:
//   k_chol_back   x_j = Linv_jj' y_j ;  y_c -= L[j, c]' x_j for every column c left of the block
//                                 (row-major rows of L read fully coalesced)
//
// Takes the place of the Gauss-Newton solve the reference obtains from LSMR
// (scipy/optimize/_lsq/trf.py:479-480) when the camera system is small.
#include "sfmba_internal.h"

namespace sfmba {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NB = 32;          // block size
constexpr int LDT = NB + 1;     // padded LDS row

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// acc(16x16 quadrant) = sum_k L[row0 + i][k] * L[col0 + j][k], k < kmax (a multiple of 16).
// v_mfma_f64_16x16x4_f64 operands: lane l holds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// result register g of lane l is D[row = (l >> 4) + 4 g][col = l & 15].  The sum over k does not
// care which k a lane group takes, so group g = l >> 4 takes the four consecutive columns
// 16 m + 4 g .. + 3 of every 16-wide slice: one 32-byte load per operand feeds four MFMAs.
__device__ __forceinline__ v4d history(const double* __restrict__ A, int n, int row0, int col0, int kmax, int lane) {
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    const double* pa = A + (size_t)(row0 + (lane & 15)) * n + 4 * (lane >> 4);
    const double* pb = A + (size_t)(col0 + (lane & 15)) * n + 4 * (lane >> 4);
#pragma unroll 2
    for (int k0 = 0; k0 < kmax; k0 += 16) {
        const double4 a = *reinterpret_cast<const double4*>(pa + k0);
        const double4 b = *reinterpret_cast<const double4*>(pb + k0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.w, b.w, acc, 0, 0, 0);
    }
    return acc;
}

// Diagonal block j0.  256 threads.
__global__ __launch_bounds__(256) void k_chol_diag(double* __restrict__ A, int n, int j0,
                                                   double* __restrict__ Linv_all, double* __restrict__ bwork,
                                                   double* __restrict__ yvec, int* __restrict__ fail) {
    __shared__ double T[NB * LDT];
    __shared__ double Li[NB * LDT];
    __shared__ int bad;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    v4d acc = history(A, n, j0 + 16 * wr, j0 + 16 * wc, j0, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = 16 * wr + (lane >> 4) + 4 * g, c = 16 * wc + (lane & 15);
        T[r * LDT + c] = A[(size_t)(j0 + r) * n + j0 + c] - acc[g];
    }
    if (tid == 0) bad = 0;
    __syncthreads();
    // right-looking Cholesky of the 32x32 block with the whole workgroup:
    // thread = (row i = tid / 8, column slot tid % 8)
    const int ti = tid >> 3, tj = tid & 7;
    for (int c = 0; c < NB; ++c) {
        const double dgl = T[c * LDT + c];
        const double l = sqrt(dgl > 0.0 ? dgl : 1.0);
        __syncthreads();
        if (tid == 0) { if (!(dgl > 0.0)) bad = 1; T[c * LDT + c] = l; }
        if (tj == 0 && ti > c) T[ti * LDT + c] /= l;
        __syncthreads();
        if (ti > c) {
            const double lic = T[ti * LDT + c];
            for (int j = c + 1 + tj; j <= ti; j += 8) T[ti * LDT + j] -= lic * T[j * LDT + c];
        }
        __syncthreads();
    }
    // inverse of the lower triangle: thread c < 32 solves L x = e_c in registers
    if (tid < NB) {
        double x[NB];
#pragma unroll
        for (int r = 0; r < NB; ++r) {
            double v = (r == tid) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) v -= T[r * LDT + k] * x[k];
            x[r] = (r < tid) ? 0.0 : v / T[r * LDT + r];
        }
#pragma unroll
        for (int r = 0; r < NB; ++r) Li[r * LDT + tid] = x[r];
    }
    __syncthreads();
    double* Linv = Linv_all + (size_t)(j0 / NB) * NB * NB;
    for (int e = tid; e < NB * NB; e += 256) {
        const int r = e / NB, c = e - NB * r;
        A[(size_t)(j0 + r) * n + j0 + c] = (c <= r) ? T[r * LDT + c] : 0.0;
        Linv[e] = (c <= r) ? Li[r * LDT + c] : 0.0;
    }
    if (tid < NB) {          // forward substitution for this block: y_j = Linv_jj b_j
        double v = 0.0;
        for (int k = 0; k <= tid; ++k) v += Li[tid * LDT + k] * bwork[j0 + k];
        yvec[j0 + tid] = v;
    }
    if (tid == 0 && bad) *fail = 1;
}

// Tile (i0, j0) below the diagonal block.  256 threads.
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ A, int n, int j0,
                                                    const double* __restrict__ Linv_all,
                                                    double* __restrict__ bwork, const double* __restrict__ yvec) {
    __shared__ double T[NB * LDT];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 1, wc = w & 1;
    const int i0 = j0 + NB * (blockIdx.x + 1);
    v4d acc = history(A, n, i0 + 16 * wr, j0 + 16 * wc, j0, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = 16 * wr + (lane >> 4) + 4 * g, c = 16 * wc + (lane & 15);
        T[r * LDT + c] = A[(size_t)(i0 + r) * n + j0 + c] - acc[g];
    }
    __syncthreads();
    const double* Linv = Linv_all + (size_t)(j0 / NB) * NB * NB;
    v4d out = {0.0, 0.0, 0.0, 0.0};
    const double* ta = T + (16 * wr + (lane & 15)) * LDT + (lane >> 4);
    const double* lb = Linv + (16 * wc + (lane & 15)) * NB + (lane >> 4);
#pragma unroll
    for (int k0 = 0; k0 < NB; k0 += 4) out = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[k0], lb[k0], out, 0, 0, 0);
    __syncthreads();
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = 16 * wr + (lane >> 4) + 4 * g, c = 16 * wc + (lane & 15);
        A[(size_t)(i0 + r) * n + j0 + c] = out[g];
        T[r * LDT + c] = out[g];
    }
    __syncthreads();
    // forward substitution rides along: b_i -= L_ij y_j ; thread = (row tid / 8, 4 columns)
    const int r = tid >> 3, part = tid & 7;
    double s = 0.0;
#pragma unroll
    for (int c = 4 * part; c < 4 * part + 4; ++c) s += T[r * LDT + c] * yvec[j0 + c];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (part == 0) bwork[i0 + r] -= s;
}

// Back substitution, block j0 (launched last block first): x_j = Linv_jj' y_j, then every
// column c < j0 gets y_c -= sum_r L[j0 + r][c] x_j[r].
__global__ __launch_bounds__(256) void k_chol_back(const double* __restrict__ A, int n, int j0,
                                                   const double* __restrict__ Linv_all,
                                                   double* __restrict__ yvec, double* __restrict__ x, int n_real) {
    __shared__ double xs[NB];
    const int tid = threadIdx.x;
    if (tid < NB) {
        const double* li = Linv_all + (size_t)(j0 / NB) * NB * NB;
        double v = 0.0;
        for (int k = tid; k < NB; ++k) v += li[k * NB + tid] * yvec[j0 + k];
        xs[tid] = v;
        if (blockIdx.x == 0 && j0 + tid < n_real) x[j0 + tid] = v;
    }
    __syncthreads();
    const int c = blockIdx.x * 256 + tid;
    if (c < j0) {
        double s = 0.0;
#pragma unroll 8
        for (int r = 0; r < NB; ++r) s += A[(size_t)(j0 + r) * n + c] * xs[r];
        yvec[c] -= s;
    }
}

// bwork = b padded with zeros; identity on the padded tail of the diagonal.
__global__ void k_dense_prepare(double* __restrict__ A, int n, int n_real, const double* __restrict__ b,
                                double* __restrict__ bwork) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    bwork[i] = i < n_real ? b[i] : 0.0;
    if (i >= n_real) A[(size_t)i * n + i] = 1.0;
}

int dense_padded(int n_real) { return cdiv(n_real, NB) * NB; }

// Factor the dense matrix A (n x n, n a multiple of 32) and solve A x = b.  work holds 2 n
// doubles.  *fail (device) is set when a pivot is not positive.
void launch_dense_cholesky_solve(double* A, int n, int n_real, double* Linv_all, double* work, int* fail,
                                 const double* b, double* x, hipStream_t st) {
    double* bwork = work;
    double* yvec = work + n;
    hipLaunchKernelGGL(k_dense_prepare, dim3(cdiv(n, 256)), dim3(256), 0, st, A, n, n_real, b, bwork);
    for (int j0 = 0; j0 < n; j0 += NB) {
        hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, st, A, n, j0, Linv_all, bwork, yvec, fail);
        const int below = (n - j0 - NB) / NB;
        if (below > 0) hipLaunchKernelGGL(k_chol_panel, dim3(below), dim3(256), 0, st, A, n, j0, Linv_all, bwork, yvec);
    }
    for (int j0 = n - NB; j0 >= 0; j0 -= NB)
        hipLaunchKernelGGL(k_chol_back, dim3(j0 > 0 ? cdiv(j0, 256) : 1), dim3(256), 0, st, A, n, j0, Linv_all, yvec, x, n_real);
}

}  // namespace sfmba
