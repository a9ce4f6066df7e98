// Dense fp64 Cholesky solve of the reduced camera system on the gfx950 matrix cores.
//
// Used when the reduced system is small and dense (6 n_cameras <= dense_max_dim): the block-CSR
// matrix is scattered into a dense row-major array padded to a multiple of 32 and factored by a
// left-looking blocked Cholesky with 32x32 blocks.  Per block column: k_chol_diag (one
// workgroup) forms the diagonal block minus its history with v_mfma_f64_16x16x4_f64, factors it in
// LDS and inverts the 32x32 triangle; k_chol_panel (one workgroup per 32-row tile below) forms
// its tile minus history with the same MFMA and multiplies by the inverse triangle (again MFMA).
// k_chol_solve does the two triangular solves with the stored inverse triangles.
//
// Takes the place of the Gauss-Newton solve the reference obtains from LSMR
// (scipy/optimize/_lsq/trf.py:479-480) when the camera system is small.
#include "sfmba_internal.h"

namespace sfmba {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NB = 32;          // block size
constexpr int LDT = NB + 1;     // padded LDS row

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// acc(16x16 quadrant) = sum_k L[row0 + i][k] * L[col0 + j][k], k < kmax.
// v_mfma_f64_16x16x4_f64 operands: lane l holds A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
// result register g of lane l is D[row = (l >> 4) + 4 g][col = l & 15].
__device__ __forceinline__ v4d history(const double* __restrict__ A, int n, int row0, int col0, int kmax, int lane) {
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    const double* pa = A + (size_t)(row0 + (lane & 15)) * n + (lane >> 4);
    const double* pb = A + (size_t)(col0 + (lane & 15)) * n + (lane >> 4);
    for (int k0 = 0; k0 < kmax; k0 += 4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0], acc, 0, 0, 0);
    return acc;
}

// Diagonal block j0: T = A_jj - L_j L_j', L_jj = chol(T), Linv = L_jj^-1.  256 threads.
__global__ __launch_bounds__(256) void k_chol_diag(double* __restrict__ A, int n, int j0,
                                                   double* __restrict__ Linv_all, int* __restrict__ fail) {
    __shared__ double T[NB * LDT];
    __shared__ double Li[NB * LDT];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
    v4d acc = history(A, n, j0 + 16 * wr, j0 + 16 * wc, j0, lane);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = 16 * wr + (lane >> 4) + 4 * g, c = 16 * wc + (lane & 15);
        T[r * LDT + c] = A[(size_t)(j0 + r) * n + j0 + c] - acc[g];
    }
    __syncthreads();
    if (w == 0) {
        // unblocked right-looking Cholesky of the 32x32 block, lane i owns row i (lanes >= 32 idle)
        const int i = lane;
        bool bad = false;
        for (int c = 0; c < NB; ++c) {
            const double d = T[c * LDT + c];
            if (!(d > 0.0)) bad = true;
            const double l = sqrt(d > 0.0 ? d : 1.0), il = 1.0 / l;
            __builtin_amdgcn_wave_barrier();
            if (i == c) T[c * LDT + c] = l;
            if (i > c && i < NB) T[i * LDT + c] *= il;
            __builtin_amdgcn_wave_barrier();
            if (i > c && i < NB) {
                const double lic = T[i * LDT + c];
                for (int j = c + 1; j <= i; ++j) T[i * LDT + j] -= lic * T[j * LDT + c];
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (bad && lane == 0) *fail = 1;
        // inverse of the lower triangle: lane c solves L x = e_c
        if (i < NB) {
            double x[NB];
#pragma unroll
            for (int r = 0; r < NB; ++r) {
                double v = (r == i) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < r; ++k) v -= T[r * LDT + k] * x[k];
                x[r] = (r < i) ? 0.0 : v / T[r * LDT + r];
            }
#pragma unroll
            for (int r = 0; r < NB; ++r) Li[r * LDT + i] = x[r];
        }
    }
    __syncthreads();
    double* Linv = Linv_all + (size_t)(j0 / NB) * NB * NB;
    for (int e = threadIdx.x; e < NB * NB; e += 256) {
        const int r = e / NB, c = e - NB * r;
        A[(size_t)(j0 + r) * n + j0 + c] = (c <= r) ? T[r * LDT + c] : 0.0;
        Linv[e] = (c <= r) ? Li[r * LDT + c] : 0.0;
    }
}

// Tile (i0, j0) below the diagonal block: L_ij = (A_ij - L_i L_j') Linv_jj'.  256 threads.
__global__ __launch_bounds__(256) void k_chol_panel(double* __restrict__ A, int n, int j0,
                                                    const double* __restrict__ Linv_all) {
    __shared__ double T[NB * LDT];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wr = w >> 1, wc = w & 1;
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
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = 16 * wr + (lane >> 4) + 4 * g, c = 16 * wc + (lane & 15);
        A[(size_t)(i0 + r) * n + j0 + c] = out[g];
    }
}

// x = (L L')^-1 b with the factor in the lower triangle of A and the inverse diagonal triangles
// in Linv_all.  One workgroup of 256 threads; the padded tail of b is treated as zero.
__global__ __launch_bounds__(256) void k_chol_solve(const double* __restrict__ A, int n, int n_real,
                                                    const double* __restrict__ Linv_all,
                                                    const double* __restrict__ b, double* __restrict__ x) {
    extern __shared__ double sh[];
    double* y = sh;           // [n]
    double* t = sh + n;       // [NB]
    const int i = threadIdx.x >> 3, part = threadIdx.x & 7;
    const int nblk = n / NB;
    for (int e = threadIdx.x; e < n; e += 256) y[e] = (e < n_real) ? b[e] : 0.0;
    __syncthreads();
    for (int jb = 0; jb < nblk; ++jb) {                      // L y = b
        const int j0 = jb * NB;
        double s = 0.0;
        const double* row = A + (size_t)(j0 + i) * n;
        for (int k = part; k < j0; k += 8) s += row[k] * y[k];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (part == 0) t[i] = y[j0 + i] - s;
        __syncthreads();
        if (threadIdx.x < NB) {
            const double* li = Linv_all + (size_t)jb * NB * NB + threadIdx.x * NB;
            double v = 0.0;
            for (int k = 0; k <= (int)threadIdx.x; ++k) v += li[k] * t[k];
            y[j0 + threadIdx.x] = v;
        }
        __syncthreads();
    }
    for (int jb = nblk - 1; jb >= 0; --jb) {                 // L' x = y
        const int j0 = jb * NB;
        double s = 0.0;
        for (int k = j0 + NB + part; k < n; k += 8) s += A[(size_t)k * n + j0 + i] * y[k];
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        if (part == 0) t[i] = y[j0 + i] - s;
        __syncthreads();
        if (threadIdx.x < NB) {
            const double* li = Linv_all + (size_t)jb * NB * NB;
            double v = 0.0;
            for (int k = threadIdx.x; k < NB; ++k) v += li[k * NB + threadIdx.x] * t[k];
            y[j0 + threadIdx.x] = v;
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < n_real; e += 256) x[e] = y[e];
}

// Identity on the padded tail of the diagonal (rows n_real .. n-1).
__global__ void k_dense_pad(double* __restrict__ A, int n, int n_real) {
    const int i = n_real + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) A[(size_t)i * n + i] = 1.0;
}

int dense_padded(int n_real) { return cdiv(n_real, NB) * NB; }

// Factor the dense matrix A (n x n, n a multiple of 32) and solve A x = b.  *fail (device) is set
// when a pivot is not positive.
void launch_dense_cholesky_solve(double* A, int n, int n_real, double* Linv_all, int* fail,
                                 const double* b, double* x, hipStream_t st) {
    if (n_real < n)
        hipLaunchKernelGGL(k_dense_pad, dim3(cdiv(n - n_real, 64)), dim3(64), 0, st, A, n, n_real);
    for (int j0 = 0; j0 < n; j0 += NB) {
        hipLaunchKernelGGL(k_chol_diag, dim3(1), dim3(256), 0, st, A, n, j0, Linv_all, fail);
        const int below = (n - j0 - NB) / NB;
        if (below > 0) hipLaunchKernelGGL(k_chol_panel, dim3(below), dim3(256), 0, st, A, n, j0, Linv_all);
    }
    hipLaunchKernelGGL(k_chol_solve, dim3(1), dim3(256), (size_t)(n + NB) * sizeof(double), st, A, n, n_real, Linv_all, b, x);
}

}  // namespace sfmba
