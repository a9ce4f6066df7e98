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
        st.rz = 0.0; st.rz0 = 0.0; st.tol2 = tol * tol; st.iter = 0; st.max_iter = max_iter; st.done = 0; st.pad = 0; st.ticks = 0;
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

// ---- whole solve in one launch ----------------------------------------------------------
// Conjugate gradients on the scaled system Shat xhat = bhat (block-Jacobi preconditioning folded
// into the matrix), in the single-reduction form of Chronopoulos and Gear:
//     gamma = r'r, delta = w'r (w = Shat r);  beta = gamma / gamma_prev;
//     alpha = gamma / (delta - beta gamma / alpha_prev)
//     p = r + beta p;  s = w + beta s;  xhat += alpha p;  r -= alpha s;  w = Shat r
// so that one exchange between the workgroups per iteration is enough: every workgroup adds the
// same per-workgroup (gamma, delta) records in the same order, gets the same alpha and beta,
// updates the rows it owns and, for every column its rows gather, rebuilds that column's new r
// from the previous r, w and s (three reads instead of a second exchange).  r, w and s are
// double-buffered by iteration parity; p and xhat are only touched by their owner.  One
// wavefront per camera row at a time, one lane per 6x6 block of the row, so that the gather of a
// row is a single round trip to memory.
//
// The exchange is also the grid barrier.  Records live in three rotating slots that hold a
// sentinel (all bits set) until written: workgroup B publishes its record of iteration k in
// slot k % 3 after all its vector stores have been acknowledged, thread i of every workgroup
// polls the record of workgroup i until it is no sentinel, and once B has collected all records
// of iteration k it re-arms its record in slot (k + 2) % 3 (last read at iteration k - 1, which
// everybody has left by then).  No atomic
// read-modify-write and no cache write-back or invalidation is involved: everything another
// workgroup reads within the launch moves through agent-scope relaxed loads and stores, which
// go past the caches that are not coherent between the dies, and the matrix stays in L2.
//
// The grid must be co-resident (cooperative launch).  A poll that exceeds timeout_ticks clears
// the abort word, which every polling thread checks, so the grid always drains.
constexpr int kFusedThreads = 512;
constexpr int kFusedWaves = kFusedThreads / 64;
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

// y += S x for one row-major 6x6 block.
__device__ __forceinline__ void block_times(const double* __restrict__ S, const double (&x)[6], double (&y)[6]) {
    const double2* sp = reinterpret_cast<const double2*>(S);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double2 s0 = sp[3 * i], s1 = sp[3 * i + 1], s2 = sp[3 * i + 2];
        y[i] += s0.x * x[0] + s0.y * x[1] + s1.x * x[2] + s1.y * x[3] + s2.x * x[4] + s2.y * x[5];
    }
}

// Sum of the per-lane 6-vectors over the wavefront; lane k < 6 receives component k.
__device__ __forceinline__ double row_total(const double (&y)[6], int lane) {
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double t = wave_sum(y[i]);
        if (lane == i) q = t;
    }
    return q;
}

struct Totals { double gam, del; bool ok; };

// Publish this workgroup's (g, dl) in `slot`, collect everybody's, return the totals.
__device__ __forceinline__ Totals exchange(const FusedPcg& a, int slot, double g, double dl, long long deadline,
                                           double* red, int* fail_lds) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, G = gridDim.x;
    unsigned long long* rec = reinterpret_cast<unsigned long long*>(a.dots);
    unsigned long long* abort_word = rec + 6 * (size_t)G;
    g = wave_sum(g);
    dl = wave_sum(dl);
    if (lane == 0) { red[w] = g; red[kFusedWaves + w] = dl; }
    if (tid == 0) *fail_lds = 0;
    __builtin_amdgcn_s_waitcnt(0);          // this wavefront's vector stores are acknowledged
    __syncthreads();
    if (tid == 0) {
        double sg = 0.0, sd = 0.0;
        for (int i = 0; i < kFusedWaves; ++i) { sg += red[i]; sd += red[kFusedWaves + i]; }
        unsigned long long* mine = rec + 2 * ((size_t)slot * G + blockIdx.x);
        st_word(mine, (unsigned long long)__double_as_longlong(sg));
        st_word(mine + 1, (unsigned long long)__double_as_longlong(sd));
    }
    double vg = 0.0, vd = 0.0;
    int failed = 0;
    for (int i = tid; i < G; i += kFusedThreads) {
        const unsigned long long* theirs = rec + 2 * ((size_t)slot * G + i);
        unsigned long long ug, ud;
        for (int spins = 1;; ++spins) {
            ug = ld_word(theirs);
            ud = ld_word(theirs + 1);
            if (ug != kSentinel && ud != kSentinel) break;
            if ((spins & 31) == 0) {
                if (ld_word(abort_word) != kSentinel) { failed = 1; break; }
                if ((long long)wall_clock64() > deadline) { st_word(abort_word, 0ull); failed = 1; break; }
            }
        }
        if (failed) break;
        vg += __longlong_as_double((long long)ug);
        vd += __longlong_as_double((long long)ud);
    }
    vg = wave_sum(vg);
    vd = wave_sum(vd);
    if (lane == 0) { red[2 * kFusedWaves + w] = vg; red[3 * kFusedWaves + w] = vd; }
    if (failed) *fail_lds = 1;
    __syncthreads();
    Totals t;
    t.gam = 0.0;
    t.del = 0.0;
#pragma unroll
    for (int i = 0; i < kFusedWaves; ++i) { t.gam += red[2 * kFusedWaves + i]; t.del += red[3 * kFusedWaves + i]; }
    t.ok = *fail_lds == 0;
    __syncthreads();
    if (tid == 0) {
        // re-arm the slot of the exchange after the next one: it was last read one exchange ago,
        // which everybody has completed; the store is acknowledged (s_waitcnt at the entry of
        // the next exchange) before this workgroup publishes again
        unsigned long long* arm = rec + 2 * ((size_t)((slot + 2) % 3) * G + blockIdx.x);
        st_word(arm, kSentinel);
        st_word(arm + 1, kSentinel);
    }
    return t;
}

__global__ __launch_bounds__(kFusedThreads) void k_pcg_fused(FusedPcg a) {
    __shared__ double red[4 * kFusedWaves];
    __shared__ int fail_lds;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int G = gridDim.x, W = G * kFusedWaves, gw = blockIdx.x * kFusedWaves + w;
    const long long deadline = (long long)wall_clock64() + a.timeout_ticks;

    // the first row of this wavefront stays in registers for the whole solve: block range,
    // this lane's column and its 6x6 block
    int c_b = 0, c_b1 = 0;
    size_t c_col = 0;
    double c_S[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) c_S[i] = 0.0;
    if (gw < a.n_cam) {
        c_b = a.bsr_ptr[gw] + lane;
        c_b1 = a.bsr_ptr[gw + 1];
        if (c_b < c_b1) {
            c_col = 6 * (size_t)a.bsr_col[c_b];
            const double2* sp = reinterpret_cast<const double2*>(a.Shat + 36 * (size_t)c_b);
#pragma unroll
            for (int i = 0; i < 18; ++i) { const double2 v = sp[i]; c_S[2 * i] = v.x; c_S[2 * i + 1] = v.y; }
        }
    }

    // iteration "-1": r0 = bhat, w0 = Shat r0, p = s = xhat = 0
    double g = 0.0, dl = 0.0;
    for (int row = gw; row < a.n_cam; row += W) {
        double y[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        const int b1 = a.bsr_ptr[row + 1];
        for (int b = a.bsr_ptr[row] + lane; b < b1; b += 64) {
            const double2* rv = reinterpret_cast<const double2*>(a.bhat + 6 * (size_t)a.bsr_col[b]);
            const double2 r0 = rv[0], r1 = rv[1], r2 = rv[2];
            const double rc[6] = {r0.x, r0.y, r1.x, r1.y, r2.x, r2.y};
            block_times(a.Shat + 36 * (size_t)b, rc, y);
        }
        const double q = row_total(y, lane);
        if (lane < 6) {
            const size_t o = 6 * (size_t)row + lane;
            const double rk = a.bhat[o];
            st_shared(a.r[0] + o, rk); st_shared(a.w[0] + o, q); st_shared(a.s[1] + o, 0.0);
            a.p[o] = 0.0; a.xh[o] = 0.0;
            g += rk * rk;
            dl += q * rk;
        }
    }
    Totals tot = exchange(a, 0, g, dl, deadline, red, &fail_lds);

    double gam = tot.gam, gam0 = tot.gam, gam_prev = 0.0, alpha_prev = 0.0;
    int it = 0, flag = tot.ok ? 0 : 2;
    while (flag == 0) {
        const int par = it & 1;
        gam = tot.gam;
        if (it == 0) {
            if (!(gam > 0.0)) break;                        // zero (or NaN) right-hand side
        } else if (!(gam > a.tol2 * gam0) || it >= a.max_iter) {
            break;
        }
        const double beta = it ? gam / gam_prev : 0.0;
        const double den = it ? tot.del - beta * gam / alpha_prev : tot.del;
        if (!(den > 0.0)) { flag = 1; break; }               // breakdown: caller falls back
        const double alpha = gam / den;
        gam_prev = gam;
        alpha_prev = alpha;
        const double* ro = a.r[par];
        const double* wo = a.w[par];
        const double* so = a.s[par ^ 1];
        g = 0.0;
        dl = 0.0;
        for (int row = gw; row < a.n_cam; row += W) {
            const bool first = row == gw;
            const int b1 = first ? c_b1 : a.bsr_ptr[row + 1];
            int b = first ? c_b : a.bsr_ptr[row] + lane;
            // the owner's own values travel with the gather: one round trip for both
            const size_t o = 6 * (size_t)row + (lane < 6 ? lane : 0);
            const double rk = ld_shared(ro + o), wk = ld_shared(wo + o), sk = ld_shared(so + o);
            const double pk = a.p[o], xk = a.xh[o];
            double y[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if (b < b1) {
                const size_t col = first ? c_col : 6 * (size_t)a.bsr_col[b];
                double rc[6];
#pragma unroll
                for (int j = 0; j < 6; ++j)
                    rc[j] = fma(-alpha, fma(beta, ld_shared(so + col + j), ld_shared(wo + col + j)), ld_shared(ro + col + j));
                if (first) {
#pragma unroll
                    for (int i = 0; i < 6; ++i)
                        y[i] = c_S[6 * i] * rc[0] + c_S[6 * i + 1] * rc[1] + c_S[6 * i + 2] * rc[2] +
                               c_S[6 * i + 3] * rc[3] + c_S[6 * i + 4] * rc[4] + c_S[6 * i + 5] * rc[5];
                } else {
                    block_times(a.Shat + 36 * (size_t)b, rc, y);
                }
                for (b += 64; b < b1; b += 64) {
                    const size_t col2 = 6 * (size_t)a.bsr_col[b];
#pragma unroll
                    for (int j = 0; j < 6; ++j)
                        rc[j] = fma(-alpha, fma(beta, ld_shared(so + col2 + j), ld_shared(wo + col2 + j)), ld_shared(ro + col2 + j));
                    block_times(a.Shat + 36 * (size_t)b, rc, y);
                }
            }
            const double q = row_total(y, lane);
            if (lane < 6) {
                const double sn = fma(beta, sk, wk);
                const double pn = fma(beta, pk, rk);
                const double rn = fma(-alpha, sn, rk);
                a.xh[o] = fma(alpha, pn, xk);
                a.p[o] = pn;
                st_shared(a.s[par] + o, sn);
                st_shared(a.r[par ^ 1] + o, rn);
                st_shared(a.w[par ^ 1] + o, q);
                g += rn * rn;
                dl += q * rn;
            }
        }
        ++it;
        tot = exchange(a, it % 3, g, dl, deadline, red, &fail_lds);
        if (!tot.ok) flag = 2;
    }
    // x = F' xhat for the rows this wavefront owns
    if (flag != 2) {
        for (int row = gw; row < a.n_cam; row += W) {
            const double xk = lane < 6 ? a.xh[6 * (size_t)row + lane] : 0.0;
            double v = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double xj = __shfl(xk, j, 64);
                if (lane < 6) v += a.F[36 * (size_t)row + 6 * j + lane] * xj;
            }
            if (lane < 6) a.x[6 * (size_t)row + lane] = v;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        PcgState st;
        st.rz = gam; st.rz0 = gam0; st.tol2 = a.tol2; st.iter = it; st.max_iter = a.max_iter; st.done = 1; st.pad = flag;
        st.ticks = (long long)wall_clock64() - (deadline - a.timeout_ticks);
        *a.out = st;
    }
}

int pcg_fused_max_blocks(int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg_fused, kFusedThreads, 0) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    return per_cu > 0 ? cus : 0;          // one workgroup per compute unit keeps the exchange cheap
}

int pcg_fused_rows_per_block() { return kFusedWaves; }
size_t pcg_fused_record_words(int blocks) { return 6 * (size_t)blocks + 1; }

// Arms every record slot and the abort word (all bits set), then launches the solve.
hipError_t launch_pcg_fused(const FusedPcg& f, int blocks, hipStream_t st) {
    hipError_t e = hipMemsetAsync(f.dots, 0xFF, pcg_fused_record_words(blocks) * sizeof(unsigned long long), st);
    if (e != hipSuccess) return e;
    FusedPcg arg = f;
    void* params[] = {&arg};
    return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(k_pcg_fused), dim3(blocks), dim3(kFusedThreads), params, 0, st);
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
