// Block-Jacobi preconditioned conjugate gradients on the reduced camera system (gfx950, fp64).
// Takes the place of LSMR in the reference's step computation
// (scipy/optimize/_lsq/trf.py:479-480, scipy/sparse/linalg/_isolve/lsmr.py:29-486).
//
// The solver state lives in device memory and is double-buffered by iteration parity `cur`:
// a kernel of iteration i reads pcg[cur] / pp[cur] and writes pcg[cur^1] / pp[cur^1], so no
// workgroup can see a value another workgroup of the same launch has just replaced.  All dot
// products are per-workgroup partials added in a fixed order by every consumer, hence bitwise
// identical everywhere and from run to run.  Two launches per iteration in explicit mode:
//
//   k_pcg_spmv_bsr   head: rz = sum of r'z partials, convergence test, beta = rz / rz_prev;
//                    direction p = z + beta p_old formed on the fly for every gathered column;
//                    q = S p on the block-CSR matrix (one wavefront per block row); p'q partials
//   k_pcg_update     alpha = rz / p'q;  x += alpha p;  r -= alpha q;  z = M^-1 r;  r'z partials
//
// In implicit mode k_pcg_dir does the head and the direction, the observation kernels of
// sfmba_kernels.hip form the Schur product, k_pcg_q adds U* p, and k_pcg_update is shared.
// Kernels queued after convergence see pcg[.].done and return at once.
#include "sfmba_internal.h"

namespace sfmba {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

constexpr int kCamsPerUpdateBlock = 42;      // 42 cameras x 6 rows = 252 of 256 threads
constexpr int kCamsPerSpmvBlock = kBlock / 64;   // one wavefront per camera row

int pcg_update_blocks(const Dev& d) { return cdiv(d.n_cam, kCamsPerUpdateBlock); }
int pcg_spmv_blocks(const Dev& d) { return cdiv(d.n_cam, kCamsPerSpmvBlock); }

// Sum of n partials by the whole workgroup, result broadcast to every thread (fixed order).
__device__ __forceinline__ double block_total(const double* __restrict__ part, int n, double* scratch) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) v += part[i];
    const double s = block_sum(v, scratch);
    __syncthreads();
    if (threadIdx.x == 0) scratch[0] = s;
    __syncthreads();
    const double r = scratch[0];
    __syncthreads();
    return r;
}

// x = 0, r = b, z = M^-1 r, p_old = 0, r'z partials; thread = (camera, row).
__global__ __launch_bounds__(kBlock) void k_pcg_init(Dev d, double tol, int max_iter) {
    __shared__ double red[16];
    __shared__ double rs[kBlock];
    const int local = threadIdx.x / 6, k = threadIdx.x - 6 * local;
    const int c = blockIdx.x * kCamsPerUpdateBlock + local;
    const bool act = local < kCamsPerUpdateBlock && c < d.n_cam;
    const size_t o = 6 * (size_t)c + k;
    const double rk = act ? d.rhs[o] : 0.0;
    rs[threadIdx.x] = rk;
    __syncthreads();
    double dot = 0.0;
    if (act) {
        const double* m = d.Minv + 36 * (size_t)c + 6 * k;
        const double* r6 = rs + 6 * local;
        const double z = m[0] * r6[0] + m[1] * r6[1] + m[2] * r6[2] + m[3] * r6[3] + m[4] * r6[4] + m[5] * r6[5];
        d.px[o] = 0.0; d.pr[o] = rk; d.pz[o] = z; d.pp[0][o] = 0.0;
        dot = rk * z;
    }
    const double s = block_sum(dot, red);
    if (threadIdx.x == 0) d.dot_part[d.dot_off + blockIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PcgState st;
        st.rz = 0.0; st.rz0 = 0.0; st.tol2 = tol * tol; st.iter = 0; st.max_iter = max_iter; st.done = 0; st.pad = 0;
        d.pcg[0] = st;
        d.pcg[1] = st;
    }
}

// Head of an iteration, executed identically by every workgroup.  Returns false when the
// iteration must not run; thread 0 of workgroup 0 writes the next state.
struct Head { bool run; double beta; };
__device__ __forceinline__ Head pcg_head(const Dev& d, int cur, double* scratch) {
    const PcgState st = d.pcg[cur];
    Head h;
    h.run = false;
    h.beta = 0.0;
    if (st.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) d.pcg[cur ^ 1] = st;
        return h;
    }
    const double rz = block_total(d.dot_part + d.dot_off, (d.n_cam + kCamsPerUpdateBlock - 1) / kCamsPerUpdateBlock, scratch);
    PcgState nx = st;
    nx.rz = rz;
    if (st.iter == 0) {
        nx.rz0 = rz;
        nx.done = !(rz > 0.0);                      // zero (or NaN) right-hand side
    } else {
        h.beta = rz / st.rz;
        nx.done = !(rz > st.tol2 * st.rz0) || st.iter >= st.max_iter;
    }
    h.run = !nx.done;
    if (h.run) nx.iter = st.iter + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) d.pcg[cur ^ 1] = nx;
    return h;
}

// Explicit mode: head, direction and q = S p in one launch.  One wavefront per camera row:
// lane = (block slot 0..9, row 0..5); each lane walks every tenth block of the row.
__global__ __launch_bounds__(kBlock) void k_pcg_spmv_bsr(Dev d, int cur) {
    __shared__ double red[16];
    __shared__ double part[kBlock];
    const Head h = pcg_head(d, cur, red);
    if (!h.run) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * kCamsPerSpmvBlock + w;
    const int sub = lane / 6, k = lane - 6 * sub;
    const double* __restrict__ z = d.pz;
    const double* __restrict__ po = d.pp[cur];
    double acc = 0.0;
    if (c < d.n_cam && sub < 10) {
        const int b1 = d.bsr_ptr[c + 1];
        for (int b = d.bsr_ptr[c] + sub; b < b1; b += 10) {
            const double2* s = reinterpret_cast<const double2*>(d.Sfull + 36 * (size_t)b + 6 * k);
            const size_t col = 6 * (size_t)d.bsr_col[b];
            const double2* zv = reinterpret_cast<const double2*>(z + col);
            const double2* pv = reinterpret_cast<const double2*>(po + col);
            const double2 s0 = s[0], s1 = s[1], s2 = s[2];
            const double2 z0 = zv[0], z1 = zv[1], z2 = zv[2];
            const double2 p0 = pv[0], p1 = pv[1], p2 = pv[2];
            acc += s0.x * (z0.x + h.beta * p0.x) + s0.y * (z0.y + h.beta * p0.y) +
                   s1.x * (z1.x + h.beta * p1.x) + s1.y * (z1.y + h.beta * p1.y) +
                   s2.x * (z2.x + h.beta * p2.x) + s2.y * (z2.y + h.beta * p2.y);
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    double dot = 0.0;
    if (c < d.n_cam && lane < 6) {
        const double* pw = part + 64 * w + lane;
        double q = 0.0;
#pragma unroll
        for (int s = 0; s < 10; ++s) q += pw[6 * s];
        const size_t o = 6 * (size_t)c + lane;
        const double pn = z[o] + h.beta * po[o];
        d.pp[cur ^ 1][o] = pn;
        d.pq[o] = q;
        dot = pn * q;
    }
    const double s = block_sum(dot, red);
    if (threadIdx.x == 0) d.dot_part[blockIdx.x] = s;
}

// Implicit mode: head and direction only; thread = camera.
__global__ __launch_bounds__(kBlock) void k_pcg_dir(Dev d, int cur) {
    __shared__ double red[16];
    const Head h = pcg_head(d, cur, red);
    if (!h.run) return;
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= d.n_cam) return;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const size_t o = 6 * (size_t)c + i;
        d.pp[cur ^ 1][o] = d.pz[o] + h.beta * d.pp[cur][o];
    }
}

// Implicit mode: q = U* p - qS, p'q partials; thread = camera.
__global__ __launch_bounds__(kBlock) void k_pcg_q(Dev d, double lambda, int cur) {
    __shared__ double red[16];
    if (d.pcg[cur ^ 1].done) return;
    const int c = blockIdx.x * kBlock + threadIdx.x;
    double dot = 0.0;
    if (c < d.n_cam) {
        double p[6], q[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) p[i] = d.pp[cur ^ 1][6 * (size_t)c + i];
        ustar_times(d.UG + 27 * (size_t)c, lambda, p, q);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            q[i] -= d.qS[6 * (size_t)c + i];
            d.pq[6 * (size_t)c + i] = q[i];
            dot += p[i] * q[i];
        }
    }
    const double s = block_sum(dot, red);
    if (threadIdx.x == 0) d.dot_part[blockIdx.x] = s;
}

// alpha = rz / p'q ; x += alpha p ; r -= alpha q ; z = M^-1 r ; r'z partials.  thread = (camera, row).
__global__ __launch_bounds__(kBlock) void k_pcg_update(Dev d, int cur, int n_pq_parts) {
    __shared__ double red[16];
    __shared__ double rs[kBlock];
    const PcgState st = d.pcg[cur ^ 1];
    if (st.done) return;
    const double pq = block_total(d.dot_part, n_pq_parts, red);
    const double alpha = st.rz / pq;
    const int local = threadIdx.x / 6, k = threadIdx.x - 6 * local;
    const int c = blockIdx.x * kCamsPerUpdateBlock + local;
    const bool act = local < kCamsPerUpdateBlock && c < d.n_cam;
    const size_t o = 6 * (size_t)c + k;
    double rk = 0.0;
    if (act) {
        d.px[o] += alpha * d.pp[cur ^ 1][o];
        rk = d.pr[o] - alpha * d.pq[o];
        d.pr[o] = rk;
    }
    rs[threadIdx.x] = rk;
    __syncthreads();
    double dot = 0.0;
    if (act) {
        const double* m = d.Minv + 36 * (size_t)c + 6 * k;
        const double* r6 = rs + 6 * local;
        const double z = m[0] * r6[0] + m[1] * r6[1] + m[2] * r6[2] + m[3] * r6[3] + m[4] * r6[4] + m[5] * r6[5];
        d.pz[o] = z;
        dot = rk * z;
    }
    const double s = block_sum(dot, red);
    if (threadIdx.x == 0) d.dot_part[d.dot_off + blockIdx.x] = s;
}

void launch_pcg_init(const Dev& d, double tol, int max_iter, hipStream_t st) {
    hipLaunchKernelGGL(k_pcg_init, dim3(pcg_update_blocks(d)), dim3(kBlock), 0, st, d, tol, max_iter);
}
void launch_pcg_spmv_bsr(const Dev& d, int cur, hipStream_t st) {
    hipLaunchKernelGGL(k_pcg_spmv_bsr, dim3(pcg_spmv_blocks(d)), dim3(kBlock), 0, st, d, cur);
}
void launch_pcg_dir(const Dev& d, int cur, hipStream_t st) {
    hipLaunchKernelGGL(k_pcg_dir, dim3(blocks_cam(d)), dim3(kBlock), 0, st, d, cur);
}
void launch_pcg_q(const Dev& d, double lambda, int cur, hipStream_t st) {
    hipLaunchKernelGGL(k_pcg_q, dim3(blocks_cam(d)), dim3(kBlock), 0, st, d, lambda, cur);
}
void launch_pcg_update(const Dev& d, int cur, int n_pq_parts, hipStream_t st) {
    hipLaunchKernelGGL(k_pcg_update, dim3(pcg_update_blocks(d)), dim3(kBlock), 0, st, d, cur, n_pq_parts);
}

}  // namespace sfmba
