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

// thread = (camera, row) in the vector kernels: 42 cameras x 6 rows (28 x 9) = 252 of 256 threads
template <int ND> constexpr int cams_per_update_block() { return kBlock / ND; }
constexpr int kCamsPerSpmvBlock = kBlock / 64;   // one wavefront per camera row

int pcg_update_blocks(const Dev& d) { return cdiv(d.n_cam, kBlock / cam_dim(d)); }
int pcg_spmv_blocks(const Dev& d) { return cdiv(d.n_cam, kCamsPerSpmvBlock); }

// Sum of n partials by the whole workgroup, result broadcast to every thread (fixed order).
__device__ __forceinline__ double block_total(const double* __restrict__ part, int n, double* scratch) {
    // eight loads of a thread are requested together (indices clamped into the array, what lies
    // beyond it masked afterwards): a loop that waits for every load costs a memory round trip per
    // 256 partials, and thousands of partials arrive per iteration at ten thousand cameras
    double v = 0.0;
    const int step = blockDim.x;
    for (int i = threadIdx.x; i < n; i += 8 * step) {
        double a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = part[min(i + j * step, n - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) v += i + j * step < n ? a[j] : 0.0;
    }
    const double s = block_sum(v, scratch);
    __syncthreads();
    if (threadIdx.x == 0) scratch[0] = s;
    __syncthreads();
    const double r = scratch[0];
    __syncthreads();
    return r;
}

// One row of z = M^-1 r: m is the row of the camera's inverse block, r its ND residual numbers.
template <int ND>
__device__ __forceinline__ double precond_row(const double* __restrict__ m, const double* r) {
    double z = m[0] * r[0];
#pragma unroll
    for (int i = 1; i < ND; ++i) z += m[i] * r[i];
    return z;
}

// x = 0, r = b, z = M^-1 r, p_old = 0, r'z partials; thread = (camera, row).
template <int ND>
__global__ __launch_bounds__(kBlock) void k_pcg_init(Dev d, double tol, int max_iter, double tol_loose, double q_abs) {
    constexpr int kCams = cams_per_update_block<ND>();
    __shared__ double red[16];
    __shared__ double rs[kBlock];
    const int local = threadIdx.x / ND, k = threadIdx.x - ND * local;
    const int c = blockIdx.x * kCams + local;
    const bool act = local < kCams && c < d.n_cam;
    const size_t o = ND * (size_t)c + k;
    const double rk = act ? d.rhs[o] : 0.0;
    rs[threadIdx.x] = rk;
    __syncthreads();
    double dot = 0.0;
    if (act) {
        const double z = precond_row<ND>(d.Minv + ND * ND * (size_t)c + ND * k, rs + ND * local);
        d.px[o] = 0.0; d.pr[o] = rk; d.pz[o] = z; d.pp[0][o] = 0.0;
        dot = rk * z;
    }
    const double s = block_sum(dot, red);
    if (threadIdx.x == 0) d.dot_part[d.dot_off + blockIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        PcgState st;
        st.rz = 0.0; st.rz0 = 0.0; st.tol2 = tol * tol; st.iter = 0; st.max_iter = max_iter; st.done = 0; st.pad = 0; st.ticks = 0;
        st.tol2_loose = tol_loose * tol_loose; st.q_abs = q_abs; st.dq1 = st.dq2 = 1e300;
        d.pcg[0] = st;
        d.pcg[1] = st;
        if (d.pcg_mirror) d.pcg_mirror[0] = st;
    }
}

// Head of an iteration, executed identically by every workgroup.  Returns false when the
// iteration must not run; thread 0 of workgroup 0 writes the next state.
struct Head { bool run; double beta; };
template <int ND>
__device__ __forceinline__ Head pcg_head(const Dev& d, int cur, double* scratch) {
    constexpr int kCams = cams_per_update_block<ND>();
    // (the record is read and written field by field: a copy of the whole structure leaves its
    // untouched tail in a stack slot, which the compiler moves to LDS at an offset it derives from
    // the workgroup shape in the dispatch packet - a read from host memory by the first wavefront
    // of every compute unit, 13 us per launch measured on a kernel that did nothing else)
    const PcgState& in = d.pcg[cur];
    PcgState st;
    st.rz = in.rz; st.rz0 = in.rz0; st.tol2 = in.tol2; st.tol2_loose = in.tol2_loose; st.q_abs = in.q_abs;
    st.dq1 = in.dq1; st.dq2 = in.dq2; st.iter = in.iter; st.max_iter = in.max_iter; st.done = in.done;
    st.pad = in.pad; st.ticks = in.ticks;
    auto put = [&](const PcgState& v) {
        PcgState& o = d.pcg[cur ^ 1];
        o.rz = v.rz; o.rz0 = v.rz0; o.tol2 = v.tol2; o.tol2_loose = v.tol2_loose; o.q_abs = v.q_abs;
        o.dq1 = v.dq1; o.dq2 = v.dq2; o.iter = v.iter; o.max_iter = v.max_iter; o.done = v.done;
        o.pad = v.pad; o.ticks = v.ticks;
        if (PcgState* m = d.pcg_mirror) {
            // the host's copy: `done` last, behind a fence, so that a host that sees it set finds the
            // fields that go with it
            m->rz = v.rz; m->rz0 = v.rz0; m->tol2 = v.tol2; m->tol2_loose = v.tol2_loose; m->q_abs = v.q_abs;
            m->iter = v.iter; m->max_iter = v.max_iter;
            __threadfence_system();
            m->done = v.done;
        }
    };
    Head h;
    h.run = false;
    h.beta = 0.0;
    if (st.done) {
        if (blockIdx.x == 0 && threadIdx.x == 0) put(st);
        return h;
    }
    const double rz = block_total(d.dot_part + d.dot_off, (d.n_cam + kCams - 1) / kCams, scratch);
    PcgState nx = st;
    nx.rz = rz;
    if (st.iter == 0) {
        nx.rz0 = rz;
        nx.done = !(rz > 0.0);                      // zero (or NaN) right-hand side
    } else {
        h.beta = rz / st.rz;
        // (the model rule: the residual has passed the loose tolerance and the last two iterations
        // each gained at most q_abs)
        const bool small_gain = st.q_abs >= 0.0 && !(rz > st.tol2_loose * st.rz0) && st.dq1 <= st.q_abs && st.dq2 <= st.q_abs;
        nx.done = !(rz > st.tol2 * st.rz0) || small_gain || st.iter >= st.max_iter;
        if (small_gain && rz > st.tol2 * st.rz0) nx.tol2 = st.tol2_loose;
    }
    h.run = !nx.done;
    if (h.run) nx.iter = st.iter + 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) put(nx);
    return h;
}

// Explicit mode: head, direction and q = S p in one launch.  One wavefront per camera row:
// lane = (block slot, row of the block) with 64 / ND slots (10 for 6x6 blocks, 7 for 9x9); each
// lane walks every slot-count-th block of the row.
template <int ND>
__global__ __launch_bounds__(kBlock) void k_pcg_spmv_bsr(Dev d, int cur) {
    constexpr int kSlots = 64 / ND, NN = ND * ND;
    __shared__ double red[16];
    __shared__ double part[kBlock];
    const Head h = pcg_head<ND>(d, cur, red);
    if (!h.run) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * kCamsPerSpmvBlock + w;
    const int sub = lane / ND, k = lane - ND * sub;
    const double* __restrict__ z = d.pz;
    const double* __restrict__ po = d.pp[cur];
    double acc = 0.0;
    if constexpr (ND == 6) {
        if (c < d.n_cam && sub < kSlots) {
            const int b1 = d.bsr_ptr[c + 1];
            for (int b = d.bsr_ptr[c] + sub; b < b1; b += kSlots) {
                const size_t col = 6 * (size_t)d.bsr_col[b];
                const double2* s = reinterpret_cast<const double2*>(d.Sfull + 36 * (size_t)b + 6 * k);
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
    } else {
        // rows of an odd width cannot be read with aligned 16-byte loads, and one 8-byte load per
        // lane and column would touch every cache line of the group nine times: the kSlots blocks
        // a wavefront works on are contiguous, so it copies them to LDS with coalesced loads and
        // every lane reads its row from there
        constexpr int kGroups = 1;              // slot groups staged at once (2 and 4 measured: no faster, 4 slower)
        constexpr int kStage = kGroups * kSlots * NN, kLoads = (kStage + 63) / 64;
        __shared__ double stage[kBlock / 64][kStage];
        double* st = stage[w];
        if (c < d.n_cam) {                      // c is the same for the whole wavefront
            const int b1 = d.bsr_ptr[c + 1];
            for (int gb = d.bsr_ptr[c]; gb < b1; gb += kGroups * kSlots) {
                const int nblk = min(kGroups * kSlots, b1 - gb), n = nblk * NN;
                const double* __restrict__ src = d.Sfull + NN * (size_t)gb;
                double v[kLoads];
#pragma unroll
                for (int j = 0; j < kLoads; ++j) { const int i = lane + 64 * j; v[j] = i < n ? src[i] : 0.0; }
#pragma unroll
                for (int j = 0; j < kLoads; ++j) { const int i = lane + 64 * j; if (i < kStage) st[i] = v[j]; }
                __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
                __builtin_amdgcn_wave_barrier();
                if (sub < kSlots) {
#pragma unroll
                    for (int g = 0; g < kGroups; ++g) {
                        const int bl = g * kSlots + sub;
                        if (bl < nblk) {
                            const size_t col = ND * (size_t)d.bsr_col[gb + bl];
                            const double* row = st + bl * NN + ND * k;
#pragma unroll
                            for (int i = 0; i < ND; ++i) acc += row[i] * (z[col + i] + h.beta * po[col + i]);
                        }
                    }
                }
                __builtin_amdgcn_s_waitcnt(0xc07f);
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    double dot = 0.0;
    if (c < d.n_cam && lane < ND) {
        const double* pw = part + 64 * w + lane;
        double q = 0.0;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) q += pw[ND * s];
        const size_t o = ND * (size_t)c + lane;
        const double pn = z[o] + h.beta * po[o];
        d.pp[cur ^ 1][o] = pn;
        d.pq[o] = q;
        dot = pn * q;
    }
    const double s = block_sum(dot, red);
    if (threadIdx.x == 0) d.dot_part[blockIdx.x] = s;
}

// Implicit mode: head and direction only; thread = camera.
template <int ND>
__global__ __launch_bounds__(kBlock) void k_pcg_dir(Dev d, int cur) {
    __shared__ double red[16];
    const Head h = pcg_head<ND>(d, cur, red);
    if (!h.run) return;
    const int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= d.n_cam) return;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const size_t o = ND * (size_t)c + i;
        d.pp[cur ^ 1][o] = d.pz[o] + h.beta * d.pp[cur][o];
    }
}

// Implicit mode: q = U* p - qS, p'q partials; thread = camera.
template <int ND>
__global__ __launch_bounds__(kBlock) void k_pcg_q(Dev d, double lambda, int cur) {
    __shared__ double red[16];
    if (d.pcg[cur ^ 1].done) return;
    const int c = blockIdx.x * kBlock + threadIdx.x;
    double dot = 0.0;
    if (c < d.n_cam) {
        double p[ND], q[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) p[i] = d.pp[cur ^ 1][ND * (size_t)c + i];
        ustar_times<ND>(d.UG + (ND * (ND + 1) / 2 + ND) * (size_t)c, lambda, p, q);
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            q[i] -= d.qS[ND * (size_t)c + i];
            d.pq[ND * (size_t)c + i] = q[i];
            dot += p[i] * q[i];
        }
    }
    const double s = block_sum(dot, red);
    if (threadIdx.x == 0) d.dot_part[blockIdx.x] = s;
}

// p'q partials of the direction and a product that was completed after the product kernel (the
// matrix split over the ranks: q is the ranks' sum); thread = (camera, row) as in k_pcg_update,
// which adds these partials up.
template <int ND>
__global__ __launch_bounds__(kBlock) void k_pcg_dot(Dev d, int cur) {
    constexpr int kCams = cams_per_update_block<ND>();
    __shared__ double red[16];
    if (d.pcg[cur ^ 1].done) return;
    const int local = threadIdx.x / ND, k = threadIdx.x - ND * local;
    const int c = blockIdx.x * kCams + local;
    double dot = 0.0;
    if (local < kCams && c < d.n_cam) {
        const size_t o = ND * (size_t)c + k;
        dot = d.pp[cur ^ 1][o] * d.pq[o];
    }
    const double s = block_sum(dot, red);
    if (threadIdx.x == 0) d.dot_part[blockIdx.x] = s;
}

// alpha = rz / p'q ; x += alpha p ; r -= alpha q ; z = M^-1 r ; r'z partials.  thread = (camera, row).
// kRestrict: the two-level preconditioner's restriction of the new residual as well: the sums Z'r over
// the workgroup's cameras, per cluster (its cameras lie in at most two), into rpart[workgroup][2][7];
// k_coarse_prolong adds up the workgroups of a cluster.  Saves a launch per iteration.
template <int ND, bool kRestrict>
__global__ __launch_bounds__(kBlock) void k_pcg_update(Dev d, int cur, int n_pq_parts, const double* __restrict__ Z,
                                                        const int* __restrict__ cam_cl, double* __restrict__ rpart) {
    constexpr int kCams = cams_per_update_block<ND>();
    __shared__ double red[16];
    __shared__ double rs[kBlock];
    const PcgState st = d.pcg[cur ^ 1];
    if (st.done) return;
    const double pq = block_total(d.dot_part, n_pq_parts, red);
    const double alpha = st.rz / pq;
    if (blockIdx.x == 0 && threadIdx.x == 0) {      // what this iteration gains in the quadratic model
        d.pcg[cur ^ 1].dq2 = st.dq1;
        d.pcg[cur ^ 1].dq1 = 0.5 * alpha * st.rz;
    }
    const int local = threadIdx.x / ND, k = threadIdx.x - ND * local;
    const int c = blockIdx.x * kCams + local;
    const bool act = local < kCams && c < d.n_cam;
    const size_t o = ND * (size_t)c + k;
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
        const double z = precond_row<ND>(d.Minv + ND * ND * (size_t)c + ND * k, rs + ND * local);
        d.pz[o] = z;
        dot = rk * z;
    }
    if constexpr (kRestrict) {
        // this thread's share of the seven sums: Z[c][a][k] r_k; summed over the wavefront per cluster
        // slot, then over the four wavefronts in a fixed order
        constexpr int kV = kCoarseVecs;
        __shared__ double wpart[kBlock / 64][2][kV];
        const int c_first = blockIdx.x * kCams;
        const int I0 = cam_cl[c_first], I1 = cam_cl[min(c_first + kCams, d.n_cam) - 1];     // uniform
        const int slot = act ? cam_cl[c] - I0 : 0;
        double v[kV];
#pragma unroll
        for (int a = 0; a < kV; ++a) v[a] = act ? Z[(kV * (size_t)c + a) * ND + k] * rk : 0.0;
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
        for (int a = 0; a < kV; ++a) {
            const double t0 = wave_sum(slot == 0 ? v[a] : 0.0);
            if (lane == 0) wpart[w][0][a] = t0;
        }
        if (I1 != I0) {
#pragma unroll
            for (int a = 0; a < kV; ++a) {
                const double t1 = wave_sum(slot == 1 ? v[a] : 0.0);
                if (lane == 0) wpart[w][1][a] = t1;
            }
        }
        __syncthreads();
        if (threadIdx.x < 2 * kV) {
            const int sl = threadIdx.x / kV, a = threadIdx.x - kV * sl;
            double t = 0.0;
            if (sl == 0 || I1 != I0) {
#pragma unroll
                for (int q = 0; q < kBlock / 64; ++q) t += wpart[q][sl][a];
            }
            rpart[2 * kV * (size_t)blockIdx.x + threadIdx.x] = t;
        }
    }
    const double s = block_sum(dot, red);
    if (threadIdx.x == 0) d.dot_part[d.dot_off + blockIdx.x] = s;
}

// ---- whole solve in one launch ----------------------------------------------------------
// Conjugate gradients on the scaled system Shat xhat = bhat (block-Jacobi preconditioning folded
// into the matrix), in the single-reduction form of Chronopoulos and Gear:
//     gamma = r'r, delta = w'r (w = Shat r);  beta = gamma / gamma_prev;
//     alpha = gamma / (delta - beta gamma / alpha_prev)   [= gamma / (delta - beta^2 den_prev)]
//     p = r + beta p;  s = w + beta s;  xhat += alpha p;  r -= alpha s;  w = Shat r
// Every workgroup adds the same per-workgroup (gamma, delta) records in the same order, gets the
// same alpha and beta, updates the rows it owns and, for every column its rows gather, rebuilds
// that column's new r from the column's previous (r, w, s) record; p and xhat are only touched
// by their owner.  One wavefront per camera row at a time, one lane per 6x6 block of the row;
// the first row of a wavefront (its blocks, columns and own vectors) stays in registers.
//
// Synchronisation is by data flow, there is no grid barrier.  What iteration k produces - the
// (r, w, s) record of every camera and the (gamma, delta) record of every workgroup, together
// "generation k" - goes into slot k % 4 of rotating buffers whose words hold a sentinel (all bits
// set) until written; consumers poll exactly the words they need until none is a sentinel.  A
// workgroup that has collected all (gamma, delta) records of generation k knows that everybody
// has finished reading generation k - 1 and re-arms its own words of that slot, which will take
// generation k + 3.  Before re-arming it waits for its earlier stores to be acknowledged (they
// are a whole iteration old by then), so the re-arming of iteration k - 1 has landed before
// anything of generation k + 1 is stored: whoever has seen a word of this workgroup's generation
// k + 2 cannot find stale words in the slot of generation k + 3.  Everything another workgroup
// reads within the launch moves through agent-scope relaxed loads and stores, which go past the
// caches that are not coherent between the dies: no read-modify-write atomics, no cache
// write-back or invalidation, and the matrix stays in registers (or L2).  Workgroup-internal
// barriers order LDS traffic only and never wait for those stores.
//
// The grid must be co-resident: one workgroup per compute unit at most, launched plainly (an
// otherwise idle device schedules all of them at once; the runtime's cooperative launch, which
// would guarantee it, costs tens of microseconds per launch and goes through a queue of its own
// that the runtime tears down at process exit).  A poll that exceeds its time limit - admit_ticks for the first one, which waits for every
// workgroup to show up, timeout_ticks afterwards - clears the abort word, which every polling
// wavefront checks, so the grid always drains; the caller then solves launch by launch.
constexpr int kFusedThreads = 512;
constexpr int kFusedWaves = kFusedThreads / 64;
constexpr int kVecRec = kPcgVecRec;
constexpr unsigned long long kSentinel = ~0ull;
constexpr int kSlots = kPcgSlots;

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for
// outstanding global stores to be acknowledged.
__device__ __forceinline__ void block_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


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

// y += S x for one row-major 6x6 block.
__device__ __forceinline__ void block_times(const double* __restrict__ S, const double (&x)[6], double (&y)[6]) {
    const double2* sp = reinterpret_cast<const double2*>(S);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double2 s0 = sp[3 * i], s1 = sp[3 * i + 1], s2 = sp[3 * i + 2];
        y[i] += s0.x * x[0] + s0.y * x[1] + s1.x * x[2] + s1.y * x[3] + s2.x * x[4] + s2.y * x[5];
    }
}

// The same with one row of the block in flight at a time.
__device__ __forceinline__ void block_times_rolled(const double* __restrict__ S, const double (&x)[6], double (&y)[6]) {
    const double2* sp = reinterpret_cast<const double2*>(S);
#pragma unroll 1
    for (int i = 0; i < 6; ++i) {
        const double2 s0 = sp[3 * i], s1 = sp[3 * i + 1], s2 = sp[3 * i + 2];
        const double t = s0.x * x[0] + s0.y * x[1] + s1.x * x[2] + s1.y * x[3] + s2.x * x[4] + s2.y * x[5];
        y[0] = i == 0 ? y[0] + t : y[0]; y[1] = i == 1 ? y[1] + t : y[1]; y[2] = i == 2 ? y[2] + t : y[2];
        y[3] = i == 3 ? y[3] + t : y[3]; y[4] = i == 4 ? y[4] + t : y[4]; y[5] = i == 5 ? y[5] + t : y[5];
    }
}

// Sum of the per-lane 6-vectors over the wavefront through the wavefront's LDS tile (at least
// 64 x 6 + 48 doubles); lane k < 6 receives component k.  Two levels in fixed order: 48 lanes
// (component k = lane % 6, part = lane / 6) add 8 rows each, then lane k adds the 8 parts.
__device__ __forceinline__ double row_total(const double (&y)[6], int lane, double* tile) {
    __builtin_amdgcn_wave_barrier();
    double2* row = reinterpret_cast<double2*>(tile + 6 * lane);
    row[0] = make_double2(y[0], y[1]);
    row[1] = make_double2(y[2], y[3]);
    row[2] = make_double2(y[4], y[5]);
    __builtin_amdgcn_wave_barrier();
    const int k = lane % 6, part = lane / 6;
    double q = 0.0;
    if (part < 8) {
#pragma unroll
        for (int r = 0; r < 8; ++r) q += tile[6 * (8 * part + r) + k];
        tile[64 * 6 + lane] = q;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 6) {
        q = 0.0;
#pragma unroll
        for (int p8 = 0; p8 < 8; ++p8) q += tile[64 * 6 + 6 * p8 + lane];
    }
    __builtin_amdgcn_wave_barrier();
    return q;
}

// Wave-level polling.  The records of up to 64 gathered columns are fetched cooperatively:
// lane l of step t loads word e = 64 t + l of the concatenation of the records (column e / 18,
// word e % 18), so that one instruction touches a handful of cache lines instead of 64, and
// stores it into the wavefront's LDS tile; the poll repeats until no fetched word is a sentinel.
constexpr int kGatherSteps = (64 * kVecRec + 63) / 64;       // 18 steps cover 64 records

struct Poller {
    unsigned long long* abort_word;
    long long deadline;
    __device__ __forceinline__ bool expired(int spins) const {
        if ((spins & 15) != 0) return false;
        if (ld_word(abort_word) != kSentinel) return true;
        if ((long long)wall_clock64() > deadline) { st_word(abort_word, 0ull); return true; }
        return false;
    }
    // off[t] = word offset (column * 18 + word) of this lane's step t, negative when beyond the
    // last record.  Returns false (wave-uniform) on abort or timeout.
    __device__ __forceinline__ bool records(const double* gen, const int (&off)[kGatherSteps], double* tile, int lane) const {
        constexpr int kHalf = kGatherSteps / 2;
        for (int spins = 1;; ++spins) {
            bool ok = true;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double val[kHalf];
#pragma unroll
                for (int t = 0; t < kHalf; ++t)
                    if (off[kHalf * h + t] >= 0) val[t] = ld_shared(gen + off[kHalf * h + t]);
#pragma unroll
                for (int t = 0; t < kHalf; ++t)
                    if (off[kHalf * h + t] >= 0) {
                        ok = ok && !is_sentinel(val[t]);
                        tile[64 * (kHalf * h + t) + lane] = val[t];
                    }
            }
            if (__all(ok)) return true;
            if (expired(spins)) return false;
        }
    }
    // The same for the blocks [b0, b0 + nb) of any row, offsets formed on the fly, three steps in
    // flight at a time (keeps the register footprint of this rarely hot path small).
    __device__ __forceinline__ bool records_of(const double* gen, const int* __restrict__ bsr_col, int b0, int nb,
                                               double* tile, int lane) const {
        const int words = nb * kVecRec;
        for (int spins = 1;; ++spins) {
            bool ok = true;
#pragma unroll 1
            for (int e0 = 0; e0 < words; e0 += 3 * 64) {
                double val[3];
                int e[3];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    e[t] = e0 + 64 * t + lane;
                    if (e[t] < words) {
                        const int blk = e[t] / kVecRec;
                        val[t] = ld_shared(gen + (size_t)bsr_col[b0 + blk] * kVecRec + (e[t] - kVecRec * blk));
                    }
                }
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if (e[t] < words) {
                        ok = ok && !is_sentinel(val[t]);
                        tile[e[t]] = val[t];
                    }
            }
            if (__all(ok)) return true;
            if (expired(spins)) return false;
        }
    }
};

// Word offsets of one lane for the blocks [b0, b0 + nb) of a row.
__device__ __forceinline__ void gather_offsets(const int* __restrict__ bsr_col, int b0, int nb, int lane, int (&off)[kGatherSteps]) {
#pragma unroll
    for (int t = 0; t < kGatherSteps; ++t) {
        const int e = 64 * t + lane, blk = e / kVecRec, j = e - kVecRec * blk;
        off[t] = blk < nb ? bsr_col[b0 + blk] * kVecRec + j : -1;
    }
}

__global__ __launch_bounds__(kFusedThreads) void k_pcg_fused(FusedPcg a) {
    __shared__ double red[2 * kFusedWaves];
    __shared__ double2 dotbuf[kFusedWaves];
    __shared__ __attribute__((aligned(16))) double tiles[kFusedWaves][64 * kVecRec];
    __shared__ int fail_lds;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int G = gridDim.x, W = G * kFusedWaves, gw = blockIdx.x * kFusedWaves + w;
    const long long t_start = (long long)wall_clock64();
    unsigned long long* drec = reinterpret_cast<unsigned long long*>(a.dots);
    double* tile = tiles[w];
    Poller poll;
    poll.abort_word = drec + 2 * kSlots * (size_t)G;
    poll.deadline = t_start + a.admit_ticks;          // until every workgroup has shown up
    const size_t gen_stride = (size_t)a.n_cam * kVecRec;
    if (tid == 0) fail_lds = 0;

    // the first (up to) 64 blocks of the first row of this wavefront stay in registers for the
    // whole solve: this lane's 6x6 block and its word offsets of the cooperative gather
    int c_b0 = 0, c_nb = 0, c_rest = 0;
    int c_off[kGatherSteps];
    double c_S[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) c_S[i] = 0.0;
#pragma unroll
    for (int t = 0; t < kGatherSteps; ++t) c_off[t] = -1;
    if (gw < a.n_cam) {
        c_b0 = a.bsr_ptr[gw];
        c_rest = a.bsr_ptr[gw + 1];
        c_nb = c_rest - c_b0 < 64 ? c_rest - c_b0 : 64;
        gather_offsets(a.bsr_col, c_b0, c_nb, lane, c_off);
        if (lane < c_nb) {
            const double2* sp = reinterpret_cast<const double2*>(a.Shat + 36 * (size_t)(c_b0 + lane));
#pragma unroll
            for (int i = 0; i < 18; ++i) { const double2 v = sp[i]; c_S[2 * i] = v.x; c_S[2 * i + 1] = v.y; }
        }
    }
    // own vectors of the first row (lanes < 6): r, w of the current generation, s of the previous, p, xhat
    double o_r = 0.0, o_w = 0.0, o_s = 0.0, o_p = 0.0, o_x = 0.0;

    // workgroup partial of (gamma, delta) -> record of generation `gen`
    auto publish = [&](double g, double dl, int gen) {
        // only lanes < 6 hold anything: lane 0 adds the six pairs through the LDS tile
        __builtin_amdgcn_wave_barrier();
        if (lane < 6) reinterpret_cast<double2*>(tile)[lane] = make_double2(g, dl);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            double sg = 0.0, sd = 0.0;
#pragma unroll
            for (int i = 0; i < 6; ++i) { const double2 v = reinterpret_cast<const double2*>(tile)[i]; sg += v.x; sd += v.y; }
            red[w] = sg; red[kFusedWaves + w] = sd;
        }
        block_barrier();
        if (tid == 0) {
            double sg = 0.0, sd = 0.0;
            for (int i = 0; i < kFusedWaves; ++i) { sg += red[i]; sd += red[kFusedWaves + i]; }
            unsigned long long* mine = drec + 2 * ((size_t)(gen % kSlots) * G + blockIdx.x);
            st_word(mine, (unsigned long long)__double_as_longlong(sg));
            st_word(mine + 1, (unsigned long long)__double_as_longlong(sd));
        }
    };

    // generation 0: r0 = bhat, w0 = Shat r0, s = p = xhat = 0
    {
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
            const double q = row_total(y, lane, tile);
            if (lane < 6) {
                const double rk = a.bhat[6 * (size_t)row + lane];
                double* rec = a.vec + (size_t)row * kVecRec;
                st_shared(rec + lane, rk); st_shared(rec + 6 + lane, q); st_shared(rec + 12 + lane, 0.0);
                if (row == gw) { o_r = rk; o_w = q; }
                else { a.p[6 * (size_t)row + lane] = 0.0; a.xh[6 * (size_t)row + lane] = 0.0; }
                g += rk * rk;
                dl += q * rk;
            }
        }
        publish(g, dl, 0);
    }

    double gam = 0.0, gam0 = 0.0, gam_prev = 0.0, den_prev = 0.0;
    double dq1 = 1e300, dq2 = 1e300, tol2_met = a.tol2;
    int it = 0, flag = 0;
    for (;;) {
        const int slot = it % kSlots;
        const double* vcur = a.vec + (size_t)slot * gen_stride;
        // generation `it`: the records the first row gathers and every workgroup's (gamma, delta)
        bool okw = poll.records(vcur, c_off, tile, lane);
        double sum_g = 0.0, sum_d = 0.0;
        if (okw) {
            // (polled after the records: the (gamma, delta) records are published last, and a
            // joint poll would keep re-fetching the vector records while waiting for them)
            const bool reader = tid < G;
            const unsigned long long* theirs = drec + 2 * ((size_t)slot * G + (reader ? tid : 0));
            for (int spins = 1;; ++spins) {
                bool ok = true;
                if (reader) {
                    const unsigned long long ug = ld_word(theirs), ud = ld_word(theirs + 1);
                    ok = ug != kSentinel && ud != kSentinel;
                    sum_g = __longlong_as_double((long long)ug);
                    sum_d = __longlong_as_double((long long)ud);
                }
                if (__all(ok)) break;
                if (poll.expired(spins)) { okw = false; break; }
            }
            if (!reader || !okw) { sum_g = 0.0; sum_d = 0.0; }
        }
        sum_g = wave_sum(sum_g);
        sum_d = wave_sum(sum_d);
        if (lane == 0) dotbuf[w] = make_double2(sum_g, sum_d);
        if (!okw && lane == 0) fail_lds = 1;
        block_barrier();
        if (fail_lds) { flag = 2; break; }
        poll.deadline = t_start + a.timeout_ticks;
        double del = 0.0;
        gam = 0.0;
#pragma unroll
        for (int i = 0; i < kFusedWaves; ++i) { const double2 v = dotbuf[i]; gam += v.x; del += v.y; }
        if (it == 0) {
            gam0 = gam;
            if (!(gam > 0.0)) break;                        // zero (or NaN) right-hand side
        } else {
            const bool small_gain = a.q_abs >= 0.0 && !(gam > a.tol2_loose * gam0) && dq1 <= a.q_abs && dq2 <= a.q_abs;
            if (small_gain && gam > a.tol2 * gam0) tol2_met = a.tol2_loose;
            if (!(gam > a.tol2 * gam0) || small_gain || it >= a.max_iter) break;
        }
        // 1 / alpha = delta / gamma - beta / alpha_prev, i.e. den = delta - beta^2 den_prev
        const double beta = it ? gam / gam_prev : 0.0;
        const double den = del - beta * beta * den_prev;
        if (!(den > 0.0)) { flag = 1; break; }               // breakdown: caller falls back
        const double alpha = gam / den;
        dq2 = dq1;
        dq1 = 0.5 * alpha * gam;             // what this iteration gains in the quadratic model
        gam_prev = gam;
        den_prev = den;

        __builtin_amdgcn_s_waitcnt(0);          // see the re-arming below
        double* vnext = a.vec + (size_t)((it + 1) % kSlots) * gen_stride;
        double g = 0.0, dl = 0.0;
        for (int row = gw; row < a.n_cam && okw; row += W) {
            const bool first = row == gw;
            double y[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            double rk = o_r, wk = o_w, sk = o_s, pk = o_p, xk = o_x;
            int b0 = c_b0 + c_nb;                        // blocks of the row not yet done
            const int b1 = first ? c_rest : a.bsr_ptr[row + 1];
            if (first) {
                __builtin_amdgcn_wave_barrier();
                if (lane < c_nb) {
                    const double* u = tile + kVecRec * lane;
                    double rc[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) rc[j] = fma(-alpha, fma(beta, u[12 + j], u[6 + j]), u[j]);
#pragma unroll
                    for (int i = 0; i < 6; ++i)
                        y[i] = c_S[6 * i] * rc[0] + c_S[6 * i + 1] * rc[1] + c_S[6 * i + 2] * rc[2] +
                               c_S[6 * i + 3] * rc[3] + c_S[6 * i + 4] * rc[4] + c_S[6 * i + 5] * rc[5];
                }
            } else {
                b0 = a.bsr_ptr[row];
                if (lane < 6) { pk = a.p[6 * (size_t)row + lane]; xk = a.xh[6 * (size_t)row + lane]; }
            }
            const int dpos = first ? -1 : a.bsr_diag[row];
            for (; b0 < b1 && okw; b0 += 64) {
                const int nb = b1 - b0 < 64 ? b1 - b0 : 64;
                __builtin_amdgcn_wave_barrier();
                okw = poll.records_of(vcur, a.bsr_col, b0, nb, tile, lane);
                __builtin_amdgcn_wave_barrier();
                if (!okw) break;
                if (lane < nb) {
                    const double* u = tile + kVecRec * lane;
                    double rc[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) rc[j] = fma(-alpha, fma(beta, u[12 + j], u[6 + j]), u[j]);
                    block_times_rolled(a.Shat + 36 * (size_t)(b0 + lane), rc, y);
                }
                if (dpos >= b0 && dpos < b0 + nb && lane < 6) {     // the owner's own record came along
                    const double* u = tile + kVecRec * (dpos - b0);
                    rk = u[lane]; wk = u[6 + lane]; sk = u[12 + lane];
                }
            }
            if (!okw) break;
            const double q = row_total(y, lane, tile);
            if (lane < 6) {
                const double sn = fma(beta, sk, wk);
                const double pn = fma(beta, pk, rk);
                const double rn = fma(-alpha, sn, rk);
                const double xn = fma(alpha, pn, xk);
                double* rec = vnext + (size_t)row * kVecRec;
                st_shared(rec + lane, rn); st_shared(rec + 6 + lane, q); st_shared(rec + 12 + lane, sn);
                if (first) { o_r = rn; o_w = q; o_s = sn; o_p = pn; o_x = xn; }
                else { a.p[6 * (size_t)row + lane] = pn; a.xh[6 * (size_t)row + lane] = xn; }
                g += rn * rn;
                dl += q * rn;
            }
        }
        if (!okw && lane == 0) fail_lds = 1;
        ++it;
        publish(g, dl, it);
        // Everybody has left generation it - 1 (all records of generation `it` were collected
        // above): re-arm this workgroup's words of its slot, which will take generation it + 3.
        // These stores trail the data stores so that nothing in this iteration waits for their
        // acknowledgement; the wait further up (before the data stores of the next iteration,
        // when every load of that iteration's poll has already come back) makes sure they have
        // landed before anything of generation it + 2 is stored.
        {
            double* arm = a.vec + (size_t)((it + kSlots - 2) % kSlots) * gen_stride;
            if (lane < 6)
                for (int row = gw; row < a.n_cam; row += W) {
                    unsigned long long* rec = reinterpret_cast<unsigned long long*>(arm + (size_t)row * kVecRec);
                    st_word(rec + lane, kSentinel); st_word(rec + 6 + lane, kSentinel); st_word(rec + 12 + lane, kSentinel);
                }
            if (tid == 0) {
                unsigned long long* rec = drec + 2 * ((size_t)((it + kSlots - 2) % kSlots) * G + blockIdx.x);
                st_word(rec, kSentinel);
                st_word(rec + 1, kSentinel);
            }
        }

        if (fail_lds) { flag = 2; break; }      // read after the barrier inside publish
    }
    // x = F' xhat for the rows this wavefront owns
    if (flag != 2) {
        for (int row = gw; row < a.n_cam; row += W) {
            double xk = 0.0;
            if (lane < 6) xk = row == gw ? o_x : a.xh[6 * (size_t)row + lane];
            double val = 0.0;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const double xj = __shfl(xk, j, 64);
                if (lane < 6) val += a.F[36 * (size_t)row + 6 * j + lane] * xj;
            }
            if (lane < 6) a.x[6 * (size_t)row + lane] = val;
        }
    }
    if (blockIdx.x == 0 && tid == 0) {
        PcgState st;
        st.rz = gam; st.rz0 = gam0; st.tol2 = tol2_met; st.tol2_loose = a.tol2_loose; st.q_abs = a.q_abs; st.dq1 = dq1; st.dq2 = dq2; st.iter = it; st.max_iter = a.max_iter; st.done = 1; st.pad = flag;
        st.ticks = (long long)wall_clock64() - t_start;
        *a.out = st;
        if (a.flag_slot) *a.flag_slot = (double)flag;
        if (a.flag_host) *a.flag_host = (double)flag;
    }
}

// ---- the same solve with the reduction taken off the critical path -----------------------
// Pipelined conjugate gradients (Ghysels and Vanroose): with w = Shat r carried along,
//     gamma = r'r, delta = w'r                     (published, summed by everybody later)
//     q = Shat w                                   (needs the neighbours' w only)
//     beta = gamma / gamma_prev;  alpha = gamma / (delta - beta gamma / alpha_prev)
//     z = q + beta z;  s = w + beta s;  p = r + beta p;  x += alpha p;  r -= alpha s;  w -= alpha z
// In k_pcg_fused an iteration waits for the neighbours' records and then for every workgroup's
// (gamma, delta), both published at the end of the iteration before.  Here the (gamma, delta) of an
// iteration are published BEFORE its matrix-vector product and only needed after it, so that their
// trip across the dies overlaps the exchange of the 6-number w records and the product.  Same
// exchange protocol (generation slots, sentinels, re-arming, time limits) as above.  Restrictions,
// checked by the host: every wavefront owns at most one camera row, a row has at most 64 blocks.
// Same iterates as k_pcg_fused in exact arithmetic (measured: the same iteration counts), 3.3
// instead of 4.2 us per iteration at 1 778 cameras.
// Written for ND x ND blocks, ND = 6 (pinhole camera) or 9 (camera with its own intrinsics: the
// block of a lane takes 162 of the 256 registers a wavefront may use at two per SIMD).

// Sum of the per-lane ND-vectors over the wavefront through its LDS tile (64 ND + 64 doubles);
// lane k < ND receives component k.  Two levels in fixed order, as row_total: kParts = 64 / ND
// groups of ND lanes add kRows = ceil(64 / kParts) rows each, then lane k adds the parts.
template <int ND>
__device__ __forceinline__ double row_total_n(const double (&y)[ND], int lane, double* tile) {
    if constexpr (ND == 6) {
        return row_total(y, lane, tile);
    } else {
        constexpr int kParts = 64 / ND, kRows = (64 + kParts - 1) / kParts;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < ND; ++i) tile[ND * lane + i] = y[i];
        __builtin_amdgcn_wave_barrier();
        const int k = lane % ND, part = lane / ND;
        double q = 0.0;
        if (part < kParts) {
#pragma unroll
            for (int r = 0; r < kRows; ++r)
                if (kRows * part + r < 64) q += tile[ND * (kRows * part + r) + k];
            tile[64 * ND + lane] = q;
        }
        __builtin_amdgcn_wave_barrier();
        if (lane < ND) {
            q = 0.0;
#pragma unroll
            for (int p = 0; p < kParts; ++p) q += tile[64 * ND + ND * p + lane];
        }
        __builtin_amdgcn_wave_barrier();
        return q;
    }
}

template <int ND>
__global__ __launch_bounds__(kFusedThreads) void k_pcg_pipe(FusedPcg a) {
    constexpr int kPipeRec = ND, kPipeSteps = ND, NN = ND * ND;
    __shared__ double red[2 * kFusedWaves];
    __shared__ double2 dotbuf[kFusedWaves];
    __shared__ __attribute__((aligned(16))) double tiles[kFusedWaves][64 * ND + 64];
    __shared__ int fail_lds;
    // With 9 x 9 blocks the registers are full: the four vectors only the owning lanes update
    // (s, p, z, x) and the gather offsets then live in LDS.
    constexpr bool kLean = ND > 6;
    __shared__ double own_lds[kLean ? kFusedWaves * 4 * 16 : 1];
    __shared__ int off_lds[kLean ? kFusedWaves * kPipeSteps * 64 : 1];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int G = gridDim.x, row = blockIdx.x * kFusedWaves + w;
    const bool mine = row < a.n_cam;
    const long long t_start = (long long)wall_clock64();
    unsigned long long* drec = reinterpret_cast<unsigned long long*>(a.dots);
    double* tile = tiles[w];
    double* own = own_lds + (kLean ? w * 4 * 16 : 0);
    int* offs = off_lds + (kLean ? w * kPipeSteps * 64 : 0);
    Poller poll;
    poll.abort_word = drec + 2 * kSlots * (size_t)G;
    poll.deadline = t_start + a.admit_ticks;
    const size_t gen_stride = (size_t)a.n_cam * kPipeRec;
    if (tid == 0) fail_lds = 0;
    if constexpr (kLean) {
        if (lane < 16) { own[lane] = 0.0; own[16 + lane] = 0.0; own[32 + lane] = 0.0; own[48 + lane] = 0.0; }
    }

    // this lane's block of the row and the word offsets of the cooperative gather
    int nb = 0;
    int off[kPipeSteps];
    double S[NN];
#pragma unroll
    for (int i = 0; i < NN; ++i) S[i] = 0.0;
#pragma unroll
    for (int t = 0; t < kPipeSteps; ++t) off[t] = -1;
    double o_r = 0.0, o_w = 0.0, o_s = 0.0, o_p = 0.0, o_z = 0.0, o_x = 0.0;
    if (mine) {
        const int b0 = a.bsr_ptr[row];
        nb = a.bsr_ptr[row + 1] - b0;
#pragma unroll
        for (int t = 0; t < kPipeSteps; ++t) {
            const int e = 64 * t + lane, blk = e / kPipeRec, j = e - kPipeRec * blk;
            off[t] = blk < nb ? a.bsr_col[b0 + blk] * kPipeRec + j : -1;
        }
        double y[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) y[i] = 0.0;
        if (lane < nb) {
            // generation 0: r0 = bhat is in memory, w0 = Shat r0 needs no exchange
            double rc[ND];
            if constexpr (ND == 6) {
                const double2* sp = reinterpret_cast<const double2*>(a.Shat + 36 * (size_t)(b0 + lane));
#pragma unroll
                for (int i = 0; i < 18; ++i) { const double2 v = sp[i]; S[2 * i] = v.x; S[2 * i + 1] = v.y; }
                const double2* rv = reinterpret_cast<const double2*>(a.bhat + 6 * (size_t)a.bsr_col[b0 + lane]);
                const double2 r0 = rv[0], r1 = rv[1], r2 = rv[2];
                rc[0] = r0.x; rc[1] = r0.y; rc[2] = r1.x; rc[3] = r1.y; rc[4] = r2.x; rc[5] = r2.y;
            } else {
                const double* sp = a.Shat + NN * (size_t)(b0 + lane);
#pragma unroll
                for (int i = 0; i < NN; ++i) S[i] = sp[i];
                const double* rv = a.bhat + ND * (size_t)a.bsr_col[b0 + lane];
#pragma unroll
                for (int i = 0; i < ND; ++i) rc[i] = rv[i];
            }
#pragma unroll
            for (int i = 0; i < ND; ++i) {
                double t = S[ND * i] * rc[0];
#pragma unroll
                for (int j = 1; j < ND; ++j) t += S[ND * i + j] * rc[j];
                y[i] = t;
            }
        }
        const double q = row_total_n<ND>(y, lane, tile);
        if (lane < ND) { o_r = a.bhat[ND * (size_t)row + lane]; o_w = q; }
    }
    if constexpr (kLean) {
#pragma unroll
        for (int t = 0; t < kPipeSteps; ++t) offs[64 * t + lane] = off[t];
    }

    auto publish = [&](double g, double dl, int gen) {
        __builtin_amdgcn_wave_barrier();
        if (lane < ND) reinterpret_cast<double2*>(tile)[lane] = make_double2(g, dl);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            double sg = 0.0, sd = 0.0;
#pragma unroll
            for (int i = 0; i < ND; ++i) { const double2 v = reinterpret_cast<const double2*>(tile)[i]; sg += v.x; sd += v.y; }
            red[w] = sg; red[kFusedWaves + w] = sd;
        }
        block_barrier();
        if (tid == 0) {
            double sg = 0.0, sd = 0.0;
            for (int i = 0; i < kFusedWaves; ++i) { sg += red[i]; sd += red[kFusedWaves + i]; }
            unsigned long long* rec = drec + 2 * ((size_t)(gen % kSlots) * G + blockIdx.x);
            st_word(rec, (unsigned long long)__double_as_longlong(sg));
            st_word(rec + 1, (unsigned long long)__double_as_longlong(sd));
        }
    };

    double gam = 0.0, gam0 = 0.0, gam_prev = 0.0, alpha_prev = 1.0;
    double dq1 = 1e300, dq2 = 1e300, tol2_met = a.tol2;
    int it = 0, flag = 0;
    for (;;) {
        const int slot = it % kSlots;
        double* vcur = a.vec + (size_t)slot * gen_stride;
        // the stores of the iteration before (the re-arming among them, a whole iteration old by
        // now) have landed before anything of this generation is stored
        __builtin_amdgcn_s_waitcnt(0);
        if (mine && lane < ND) st_shared(vcur + (size_t)row * kPipeRec + lane, o_w);
        publish(o_r * o_r, o_w * o_r, it);
        // Everybody's (gamma, delta) of generation it - 1 were collected in the iteration before,
        // so everybody has published that generation and left generation it - 2: re-arm this
        // workgroup's words of that slot, which will take generation it + 2.  The stores trail this
        // iteration's data and are waited for at the top of the next one, before anything of
        // generation it + 1 is stored; a reader can only look for generation it + 2 after it has
        // seen this workgroup's (gamma, delta) of generation it + 1.
        if (it >= 2) {
            double* arm = a.vec + (size_t)((it - 2) % kSlots) * gen_stride;
            if (mine && lane < ND) st_word(reinterpret_cast<unsigned long long*>(arm + (size_t)row * kPipeRec) + lane, kSentinel);
            if (tid == 0) {
                unsigned long long* rec = drec + 2 * ((size_t)((it - 2) % kSlots) * G + blockIdx.x);
                st_word(rec, kSentinel);
                st_word(rec + 1, kSentinel);
            }
        }
        // q = Shat w: the w of the row's columns, fetched cooperatively (lane-contiguous words)
        bool okw = true;
        for (int spins = 1;; ++spins) {
            bool ok = true;
            constexpr int kFlight = ND == 6 ? 6 : 3;        // loads in flight (register budget)
#pragma unroll
            for (int t0 = 0; t0 < kPipeSteps; t0 += kFlight) {
                double val[kFlight];
                int o[kFlight];
#pragma unroll
                for (int t = 0; t < kFlight; ++t) o[t] = kLean ? offs[64 * (t0 + t) + lane] : off[t0 + t];
#pragma unroll
                for (int t = 0; t < kFlight; ++t)
                    if (o[t] >= 0) val[t] = ld_shared(vcur + o[t]);
#pragma unroll
                for (int t = 0; t < kFlight; ++t)
                    if (o[t] >= 0) { ok = ok && !is_sentinel(val[t]); tile[64 * (t0 + t) + lane] = val[t]; }
            }
            if (__all(ok)) break;
            if (poll.expired(spins)) { okw = false; break; }
        }
        double y[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) y[i] = 0.0;
        __builtin_amdgcn_wave_barrier();
        if (okw && lane < nb) {
            // (column by column: one component of w in a register at a time; per row the same
            // sequence of operations as a dot product taken left to right)
            const double* u = tile + kPipeRec * lane;
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const double wj = u[j];
#pragma unroll
                for (int i = 0; i < ND; ++i) y[i] = j == 0 ? S[ND * i] * wj : fma(S[ND * i + j], wj, y[i]);
            }
        }
        const double q = row_total_n<ND>(y, lane, tile);
        // every workgroup's (gamma, delta) of this generation: published before the product above
        // (requesting them before the product and looking at them after it was measured: slower)
        double sum_g = 0.0, sum_d = 0.0;
        if (okw) {
            const bool reader = tid < G;
            const unsigned long long* theirs = drec + 2 * ((size_t)slot * G + (reader ? tid : 0));
            for (int spins = 1;; ++spins) {
                bool ok = true;
                if (reader) {
                    const unsigned long long ug = ld_word(theirs), ud = ld_word(theirs + 1);
                    ok = ug != kSentinel && ud != kSentinel;
                    sum_g = __longlong_as_double((long long)ug);
                    sum_d = __longlong_as_double((long long)ud);
                }
                if (__all(ok)) break;
                if (poll.expired(spins)) { okw = false; break; }
            }
            if (!reader || !okw) { sum_g = 0.0; sum_d = 0.0; }
        }
        sum_g = wave_sum(sum_g);
        sum_d = wave_sum(sum_d);
        if (lane == 0) dotbuf[w] = make_double2(sum_g, sum_d);
        if (!okw && lane == 0) fail_lds = 1;
        block_barrier();
        if (fail_lds) { flag = 2; break; }
        poll.deadline = t_start + a.timeout_ticks;
        double del = 0.0;
        gam = 0.0;
#pragma unroll
        for (int i = 0; i < kFusedWaves; ++i) { const double2 v = dotbuf[i]; gam += v.x; del += v.y; }
        if (it == 0) {
            gam0 = gam;
            if (!(gam > 0.0)) break;                        // zero (or NaN) right-hand side
        } else {
            const bool small_gain = a.q_abs >= 0.0 && !(gam > a.tol2_loose * gam0) && dq1 <= a.q_abs && dq2 <= a.q_abs;
            if (small_gain && gam > a.tol2 * gam0) tol2_met = a.tol2_loose;
            if (!(gam > a.tol2 * gam0) || small_gain || it >= a.max_iter) break;
        }
        const double beta = it ? gam / gam_prev : 0.0;
        const double den = del - (it ? beta * gam / alpha_prev : 0.0);
        if (!(den > 0.0)) { flag = 1; break; }               // breakdown: caller falls back
        const double alpha = gam / den;
        dq2 = dq1;
        dq1 = 0.5 * alpha * gam;             // what this iteration gains in the quadratic model
        gam_prev = gam;
        alpha_prev = alpha;
        if (lane < ND) {
            if constexpr (kLean) { o_z = own[lane]; o_s = own[16 + lane]; o_p = own[32 + lane]; o_x = own[48 + lane]; }
            o_z = fma(beta, o_z, q);
            o_s = fma(beta, o_s, o_w);
            o_p = fma(beta, o_p, o_r);
            o_x = fma(alpha, o_p, o_x);
            o_r = fma(-alpha, o_s, o_r);
            o_w = fma(-alpha, o_z, o_w);
            if constexpr (kLean) { own[lane] = o_z; own[16 + lane] = o_s; own[32 + lane] = o_p; own[48 + lane] = o_x; }
        }
        ++it;
    }
    // x = F' xhat for the row this wavefront owns
    if (flag != 2 && mine) {
        double val = 0.0;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const double xj = kLean ? own[48 + j] : __shfl(o_x, j, 64);
            if (lane < ND) val += a.F[NN * (size_t)row + ND * j + lane] * xj;
        }
        if (lane < ND) a.x[ND * (size_t)row + lane] = val;
    }
    if (blockIdx.x == 0 && tid == 0) {
        PcgState st;
        st.rz = gam; st.rz0 = gam0; st.tol2 = tol2_met; st.tol2_loose = a.tol2_loose; st.q_abs = a.q_abs; st.dq1 = dq1; st.dq2 = dq2; st.iter = it; st.max_iter = a.max_iter; st.done = 1; st.pad = flag;
        st.ticks = (long long)wall_clock64() - t_start;
        *a.out = st;
        if (a.flag_slot) *a.flag_slot = (double)flag;
        if (a.flag_host) *a.flag_host = (double)flag;
    }
}

// ---- the pipelined solve with the two-level preconditioner inside the launch ---------------
// Block-Jacobi leaves the modes that move whole groups of cameras together (sfmba_coarse.hip);
// the last linear solves of a run spend nine tenths of their iterations on them.  Here the
// additive coarse correction M^-1 = M_J^-1 + Z C Z' (C = D Einv D of a coarse operator the host
// hands over: the one of an earlier LM iteration, set up beside the other kernels) runs inside the
// single launch.  In the scaled variables the block-Jacobi part is the identity and
// Mhat^-1 = I + Zhat C Zhat' with Zhat_c = F_c^-T Z_c.  Preconditioned pipelined CG:
//     gamma = r'u, delta = w'u                         (published, summed by everybody later)
//     m = Mhat^-1 w = w + Zhat y,  y = C Zhat' w
//     n = Shat m = Shat w + G y,   G_r = sum_c Shat_rc Zhat_c per cluster of the row's columns
//     beta = gamma / gamma_prev;  alpha = gamma / (delta - beta gamma / alpha_prev)
//     z = n + beta z; q = m + beta q; s = w + beta s; p = u + beta p
//     x += alpha p; r -= alpha s; u -= alpha q; w -= alpha z
// The 7 coarse sums of a cluster are global sums like gamma and delta and travel with them: a
// workgroup's record has 16 words (gamma, delta, the partial sums Zhat' w of the at most two
// clusters its eight rows lie in), published before the product; thread (cluster, vector) of every
// workgroup reads that sum's terms from the records of the workgroups with rows in the cluster and
// adds them in a fixed order, then the workgroup multiplies the rows of C it needs (the clusters of
// the columns of its rows, at most kC2WgCl x 7 rows, kept in LDS; thread = (row, column mod 16)) -
// so the coarse correction costs no further trip across the dies, only one more workgroup barrier.
// Written for ND x ND blocks like k_pcg_pipe.  With ND = 9 the block of a lane leaves no registers
// for the factor F_c of a column's camera: Zhat of every camera is then formed by k_coarse_zhat in
// front of the launch, the vectors only the owning lanes update live in LDS, and values every lane
// holds alike are moved to scalar registers (7.6 us per iteration at 1 778 cameras against 4.4
// without the correction, for a quarter of the iterations).
// Restrictions, checked by the host: those of k_pcg_pipe, at most 256 workgroups,
// clusters of at least 8 cameras whose rows lie in at most kC2Span workgroups, a coarse dimension of
// at most 288 after padding, at most kC2WgCl clusters among the columns of a workgroup's rows
// (and so at most kC2RowCl = kC2WgCl among those of one row).
constexpr int kC2Rec = kPcgRecWords;    // words of a workgroup's record
constexpr int kC2RowCl = 4;
constexpr int kC2WgCl = 4;
constexpr int kC2Cols = 18;             // columns of C per thread: n_pad <= 16 * 18
constexpr int kC2MaxBlocks = 256;
constexpr int kC2Span = 7;              // workgroups with rows in one cluster, at most

// A value every lane holds alike, moved to scalar registers (where vector registers are short).
__device__ __forceinline__ double to_scalar(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffll)), hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int ND>
__global__ __launch_bounds__(kFusedThreads) void k_pcg_pipe2(FusedPcg a) {
    constexpr int NV = kCoarseVecs, NN = ND * ND;
    __shared__ double red[kFusedWaves][kC2Rec];
    __shared__ double2 dotbuf[kFusedWaves];
    __shared__ __attribute__((aligned(16))) double tiles[kFusedWaves][64 * ND + 64];
    __shared__ double Gl[kFusedWaves][kC2RowCl][ND][NV];
    __shared__ double zhat[kFusedWaves][ND][NV];
    __shared__ int rowwg[kFusedWaves][kC2RowCl], wslot[kFusedWaves];
    __shared__ double tvec[16 * kC2Cols];
    __shared__ double ywg[kC2WgCl * NV];
    __shared__ double crows[kC2WgCl * NV][16 * kC2Cols + 4];     // (the 4: rows start 8 banks apart)
    __shared__ int off_lds[kFusedWaves][ND][64];
    __shared__ int fail_lds;
    // With 9 x 9 blocks the registers are full: the five vectors only the owning lanes update
    // (z, q, s, p, x) then live in LDS, and the gather keeps three loads in flight instead of ND.
    constexpr bool kLean = ND > 6;
    __shared__ double own_lds[kLean ? kFusedWaves * 5 * 16 : 1];
    __shared__ int tb_lds[kLean ? kFusedThreads : 1];           // per thread: first workgroup | span << 8 | second-cluster bits << 12
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    double* own = own_lds + (kLean ? (tid >> 6) * 5 * 16 : 0);
    if constexpr (kLean) {
        if (lane < 16) { own[lane] = 0.0; own[16 + lane] = 0.0; own[32 + lane] = 0.0; own[48 + lane] = 0.0; own[64 + lane] = 0.0; }
    }
    const int G = gridDim.x, row = blockIdx.x * kFusedWaves + w;
    const bool mine = row < a.n_cam;
    const long long t_start = (long long)wall_clock64();
    unsigned long long* drec = reinterpret_cast<unsigned long long*>(a.dots);
    double* tile = tiles[w];
    Poller poll;
    poll.abort_word = drec + kC2Rec * kSlots * (size_t)G;
    poll.deadline = t_start + a.admit_ticks;
    const size_t gen_stride = (size_t)a.n_cam * ND;
    if (tid == 0) fail_lds = 0;
    for (int i = tid; i < 16 * kC2Cols; i += kFusedThreads) tvec[i] = 0.0;

    // this lane's block of the row, the word offsets of the cooperative gather, the cluster of
    // the block's column
    int nb = 0, b0 = 0;
    double S[NN];
#pragma unroll
    for (int i = 0; i < NN; ++i) S[i] = 0.0;
#pragma unroll
    for (int t = 0; t < ND; ++t) off_lds[w][t][lane] = -1;
    int cl = -1;
    if (mine) {
        b0 = a.bsr_ptr[row];
        nb = a.bsr_ptr[row + 1] - b0;
#pragma unroll
        for (int t = 0; t < ND; ++t) {
            const int e = 64 * t + lane, blk = e / ND, j = e - ND * blk;
            off_lds[w][t][lane] = blk < nb ? a.bsr_col[b0 + blk] * ND + j : -1;
        }
        if (lane < nb) {
            if constexpr (ND == 6) {
                const double2* sp = reinterpret_cast<const double2*>(a.Shat + 36 * (size_t)(b0 + lane));
#pragma unroll
                for (int i = 0; i < 18; ++i) { const double2 v = sp[i]; S[2 * i] = v.x; S[2 * i + 1] = v.y; }
            } else {
                // (blocks of an odd size start on 8-byte boundaries only)
                const double* sp = a.Shat + NN * (size_t)(b0 + lane);
#pragma unroll
                for (int i = 0; i < NN; ++i) S[i] = sp[i];
            }
            cl = a.cam_cl[a.bsr_col[b0 + lane]];
        }
    }
    // clusters among the row's columns: the columns ascend and clusters are runs of consecutive
    // cameras, so every cluster is a run of lanes; slot = its rank in the row
    const int cl_prev = __shfl_up(cl, 1, 64);
    const bool is_new = lane < nb && (lane == 0 || cl != cl_prev);
    const unsigned long long new_mask = __ballot(is_new);
    const int slot = __popcll(new_mask & ((2ull << lane) - 1ull)) - 1;
    const int n_rowcl = __builtin_amdgcn_readfirstlane(__popcll(new_mask));
    // the clusters among the columns of this workgroup's rows: listed by the host (count, then ids)
    const int wgcl_n = __builtin_amdgcn_readfirstlane(a.wg_cl[(kC2WgCl + 1) * blockIdx.x]);
    int wgcl[kC2WgCl];
#pragma unroll
    for (int e = 0; e < kC2WgCl; ++e) wgcl[e] = a.wg_cl[(kC2WgCl + 1) * blockIdx.x + 1 + e];
    if (is_new && slot < kC2RowCl) {
        int at = 0;
#pragma unroll
        for (int e = 0; e < kC2WgCl; ++e) if (e < wgcl_n && wgcl[e] == cl) at = e;
        rowwg[w][slot] = at;
    }
    if (lane == 0) wslot[w] = mine ? a.cam_cl[row] - a.cam_cl[blockIdx.x * kFusedWaves] : -1;
    if (tid == 0 && (G > kC2MaxBlocks || a.c_npad > 16 * kC2Cols || wgcl_n > kC2WgCl)) {
        fail_lds = 1;
        st_word(poll.abort_word, 0ull);
    }
    block_barrier();
    int flag = fail_lds ? 2 : 0;
    // the rows of C = D Einv D this workgroup needs: thread = (row of the list, column mod 16)
    const int c_rho = tid >> 4, c_part = tid & 15;
    const bool c_act = !flag && c_rho < NV * wgcl_n;
    int c_grow = 0;
    double crow_scale = 0.0;
    if (c_act) {
        const int ws = c_rho / NV;
        c_grow = NV * wgcl[ws] + (c_rho - NV * ws);
        crow_scale = a.c_scale[c_grow];
    }
    // the workgroups whose records hold the partial sums of cluster I (thread = (I, j)): a range,
    // and for each of them whether the cluster is the first or the second of that workgroup
    int tb0 = 0, tb1 = -1;
    unsigned tsecond = 0u;
    if (!flag && tid < a.c_n) {
        const int I = tid / NV;
        tb0 = a.cl_ptr[I] / kFusedWaves;
        tb1 = (a.cl_ptr[I + 1] - 1) / kFusedWaves;
        for (int b = tb0; b <= tb1 && b - tb0 < kC2Span; ++b)
            if (a.cam_cl[b * kFusedWaves] != I) tsecond |= 1u << (b - tb0);
        tb1 = min(tb1, tb0 + kC2Span - 1);      // (the host admits no longer range)
    }
    if constexpr (kLean) tb_lds[tid] = tb1 >= tb0 ? (tb0 | ((tb1 - tb0 + 1) << 8) | ((int)tsecond << 12)) : 0;

    // Zhat of this row: lane j < 7 solves F_r' x = Z_r[j] (F lower triangular, so F' x = z is a back
    // substitution) and keeps column j; lane k < 6 keeps row k
    auto back_subst = [&](const double* __restrict__ F, const double* __restrict__ z, double (&x)[ND]) {
#pragma unroll
        for (int k = ND - 1; k >= 0; --k) {
            double v = z[k];
#pragma unroll
            for (int l = k + 1; l < ND; ++l) v -= F[ND * l + k] * x[l];
            x[k] = v / F[ND * k + k];
        }
    };
    if (lane < NV) {
        double zc[ND];
#pragma unroll
        for (int k = 0; k < ND; ++k) zc[k] = 0.0;
        if (!flag && mine) {
            if constexpr (ND == 6) {
                back_subst(a.F + NN * (size_t)row, a.c_z + (NV * (size_t)row + lane) * ND, zc);
            } else {
                // (larger blocks: Zhat of every camera was formed by k_coarse_zhat in front of this launch)
                const double* __restrict__ zh = a.c_zhat + (NV * (size_t)row + lane) * ND;
#pragma unroll
                for (int k = 0; k < ND; ++k) zc[k] = zh[k];
            }
        }
#pragma unroll
        for (int k = 0; k < ND; ++k) zhat[w][k][lane] = zc[k];
    }
    __builtin_amdgcn_wave_barrier();
    // own cluster's place in the workgroup's list
    int own_ws = 0;
    if (!flag && mine) {
        const int J = a.cam_cl[row];
#pragma unroll
        for (int e = 0; e < kC2WgCl; ++e) if (e < wgcl_n && wgcl[e] == J) own_ws = e;
    }
    own_ws = __builtin_amdgcn_readfirstlane(own_ws);

    double o_r = 0.0, o_u = 0.0, o_w = 0.0, o_z = 0.0, o_q = 0.0, o_s = 0.0, o_p = 0.0, o_x = 0.0;

    // tau_j = sum_k Zhat[k][j] v_k for the vector whose component k lane k < 6 holds (lanes j < 7)
    auto restrict_row = [&](double vk) -> double {
        __builtin_amdgcn_wave_barrier();
        if (lane < ND) tile[lane] = vk;
        __builtin_amdgcn_wave_barrier();
        double t = 0.0;
        if (lane < NV) {
#pragma unroll
            for (int k = 0; k < ND; ++k) t = fma(zhat[w][k][lane], tile[k], t);
        }
        __builtin_amdgcn_wave_barrier();
        return t;
    };
    // the record of generation gen: gamma and delta summed over the workgroup's rows, the coarse
    // partial sums per cluster of the workgroup
    auto publish = [&](double g, double dl, double tau, int gen) {
        __builtin_amdgcn_wave_barrier();
        if (lane < ND) reinterpret_cast<double2*>(tile)[lane] = make_double2(g, dl);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) {
            double sg = 0.0, sd = 0.0;
#pragma unroll
            for (int i = 0; i < ND; ++i) { const double2 v = reinterpret_cast<const double2*>(tile)[i]; sg += v.x; sd += v.y; }
            red[w][0] = sg; red[w][1] = sd;
        }
        if (lane < NV) red[w][2 + lane] = tau;
        block_barrier();
        if (tid < kC2Rec) {
            // (all eight terms requested before the first is added: a loop with a branch per term
            // pays the LDS latency eight times over)
            const int sl = tid < 2 ? -2 : (tid - 2) / NV, col = tid < 2 ? tid : 2 + (tid - 2) - NV * sl;
            double sum = 0.0;
#pragma unroll
            for (int h = 0; h < kFusedWaves; h += 4) {
                double term[4];
                int ws[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) { term[v] = red[h + v][col]; ws[v] = wslot[h + v]; }
#pragma unroll
                for (int v = 0; v < 4; ++v) sum += (sl < 0 || ws[v] == sl) ? term[v] : 0.0;
            }
            st_word(drec + kC2Rec * ((size_t)(gen % kSlots) * G + blockIdx.x) + tid, (unsigned long long)__double_as_longlong(sum));
        }
    };
    // every workgroup's record of generation gen: the dots, then y = C (sum of the coarse partials)
    // for the clusters of this workgroup's list.  Returns false on a timeout or an abort.
    double gam = 0.0, del = 0.0;
    auto collect = [&](int gen, bool okw) -> bool {
        // thread t < G reads the dots of workgroup t; thread t < c_n = (cluster I, vector j) reads
        // its sum's term from the record of every workgroup with rows in cluster I
        double sum_g = 0.0, sum_d = 0.0, tsum = 0.0;
        if (okw) {
            const unsigned long long* base = drec + kC2Rec * (size_t)(gen % kSlots) * G;
            const bool reader = tid < G;
            const int tj = 2 + tid % NV;
            if constexpr (kLean) {
                const int info = tb_lds[tid];
                tb0 = info & 255; tb1 = tb0 + ((info >> 8) & 15) - 1; tsecond = (unsigned)(info >> 12);
            }
            for (int spins = 1;; ++spins) {
                bool ok = true;
                if (reader) {
                    const unsigned long long ug = ld_word(base + kC2Rec * (size_t)tid), ud = ld_word(base + kC2Rec * (size_t)tid + 1);
                    ok = ug != kSentinel && ud != kSentinel;
                    sum_g = __longlong_as_double((long long)ug);
                    sum_d = __longlong_as_double((long long)ud);
                }
                // (all loads of a round requested before the first is looked at)
                unsigned long long u[kC2Span];
#pragma unroll
                for (int q = 0; q < kC2Span; ++q)
                    if (tb0 + q <= tb1) u[q] = ld_word(base + kC2Rec * (size_t)(tb0 + q) + tj + ((tsecond >> q) & 1u ? NV : 0));
                tsum = 0.0;
#pragma unroll
                for (int q = 0; q < kC2Span; ++q)
                    if (tb0 + q <= tb1) { ok = ok && u[q] != kSentinel; tsum += __longlong_as_double((long long)u[q]); }
                if (__all(ok)) break;
                if (poll.expired(spins)) { okw = false; break; }
            }
            if (!reader || !okw) { sum_g = 0.0; sum_d = 0.0; }
        }
        sum_g = wave_sum(sum_g);
        sum_d = wave_sum(sum_d);
        if (lane == 0) dotbuf[w] = make_double2(sum_g, sum_d);
        if (tid < a.c_n) tvec[tid] = tsum;
        if (!okw && lane == 0) fail_lds = 1;
        block_barrier();
        if (fail_lds) return false;
        gam = 0.0; del = 0.0;
#pragma unroll
        for (int i = 0; i < kFusedWaves; ++i) { const double2 d2 = dotbuf[i]; gam += d2.x; del += d2.y; }
        if constexpr (kLean) { gam = to_scalar(gam); del = to_scalar(del); }
        if (NV * wgcl_n > 4 * w) {          // this wavefront has rows of the list (four per wavefront)
            double part = 0.0;
            if (c_act) {
#pragma unroll
                for (int u = 0; u < kC2Cols; ++u) part = fma(crows[c_rho][c_part + 16 * u], tvec[c_part + 16 * u], part);
            }
            part += __shfl_xor(part, 1, 64);
            part += __shfl_xor(part, 2, 64);
            part += __shfl_xor(part, 4, 64);
            part += __shfl_xor(part, 8, 64);
            if (c_part == 0 && c_act) ywg[c_rho] = crow_scale * part;
        }
        block_barrier();
        return true;
    };
    // (G y)_i of this row for lane i < 6
    auto coarse_row = [&]() -> double {
        double t = 0.0;
        for (int sl = 0; sl < n_rowcl; ++sl) {
            const double* yy = ywg + NV * rowwg[w][sl];
#pragma unroll
            for (int j = 0; j < NV; ++j) t = fma(Gl[w][sl][lane][j], yy[j], t);
        }
        return t;
    };
    auto own_correction = [&]() -> double {
        double t = 0.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) t = fma(zhat[w][lane][j], ywg[NV * own_ws + j], t);
        return t;
    };

    double gam0 = 0.0, gam_prev = 0.0, alpha_prev = 1.0;
    double dq1 = 1e300, dq2 = 1e300, tol2_met = a.tol2;
    int it = 0;
    double qa0 = 0.0;
    if (!flag) {
        // generation 0: r0 = bhat, u0 = Mhat^-1 r0 needs the coarse sums of bhat; w0 = Shat u0 =
        // Shat bhat + G y (bhat of the row's columns is in memory)
        double y[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) y[i] = 0.0;
        if (mine && lane < nb) {
            double rc[ND];
            if constexpr (ND == 6) {
                const double2* rv = reinterpret_cast<const double2*>(a.bhat + 6 * (size_t)a.bsr_col[b0 + lane]);
                const double2 r0 = rv[0], r1 = rv[1], r2 = rv[2];
                rc[0] = r0.x; rc[1] = r0.y; rc[2] = r1.x; rc[3] = r1.y; rc[4] = r2.x; rc[5] = r2.y;
            } else {
                const double* rv = a.bhat + ND * (size_t)a.bsr_col[b0 + lane];
#pragma unroll
                for (int i = 0; i < ND; ++i) rc[i] = rv[i];
            }
#pragma unroll
            for (int j = 0; j < ND; ++j) {
#pragma unroll
                for (int i = 0; i < ND; ++i) y[i] = j == 0 ? S[ND * i] * rc[0] : fma(S[ND * i + j], rc[j], y[i]);
            }
        }
        qa0 = row_total_n<ND>(y, lane, tile);
        if (mine && lane < ND) o_r = a.bhat[ND * (size_t)row + lane];
        const double tau = restrict_row(o_r);
        publish(0.0, 0.0, tau, 0);
    }
    // (while the records of generation 0 travel)
    // rows of C = D Einv D this workgroup needs, in LDS
    if (c_act) {
        const double* __restrict__ e = a.c_einv + (size_t)c_grow * a.c_npad;
#pragma unroll
        for (int u = 0; u < kC2Cols; ++u) {
            const int col = c_part + 16 * u;
            crows[c_rho][col] = col < a.c_npad ? e[col] * a.c_scale[col] : 0.0;
        }
    }
    // G_r: per cluster slot of the row the sum over its lanes of Shat_rc Zhat_c, one coarse vector at a
    // time (the column's factor is read once; the back substitution multiplies by the reciprocals
    // of its diagonal)
    if constexpr (ND == 6) {
        if (!flag && mine) {
            const bool act = lane < nb;
            const int cc = act ? a.bsr_col[b0 + lane] : row;
            const double* __restrict__ Fp = a.F + 36 * (size_t)cc;
            const double* __restrict__ Zc = a.c_z + NV * ND * (size_t)cc;
            double Fc[15], Fd[ND];          // strict lower triangle (row-major), reciprocal diagonal
            {
                int e = 0;
    #pragma unroll
                for (int l = 1; l < ND; ++l)
    #pragma unroll
                    for (int k = 0; k < l; ++k) Fc[e++] = Fp[ND * l + k];
    #pragma unroll
                for (int k = 0; k < ND; ++k) Fd[k] = 1.0 / Fp[ND * k + k];
            }
            double zn[ND];
    #pragma unroll
            for (int k = 0; k < ND; ++k) zn[k] = Zc[k];
    #pragma unroll 1
            for (int j = 0; j < NV; ++j) {
                double z[ND], x[ND], P[ND];
    #pragma unroll
                for (int k = 0; k < ND; ++k) z[k] = zn[k];
                if (j + 1 < NV) {
    #pragma unroll
                    for (int k = 0; k < ND; ++k) zn[k] = Zc[ND * (j + 1) + k];       // the next vector is on its way
                }
                // F' x = z: x[k] = (z[k] - sum_{l > k} F[l][k] x[l]) / F[k][k]
    #pragma unroll
                for (int k = ND - 1; k >= 0; --k) {
                    double v = z[k];
    #pragma unroll
                    for (int l = k + 1; l < ND; ++l) v -= Fc[l * (l - 1) / 2 + k] * x[l];
                    x[k] = v * Fd[k];
                }
    #pragma unroll
                for (int i = 0; i < ND; ++i) {
                    double t = S[ND * i] * x[0];
    #pragma unroll
                    for (int k = 1; k < ND; ++k) t = fma(S[ND * i + k], x[k], t);
                    P[i] = act ? t : 0.0;
                }
                for (int sl = 0; sl < n_rowcl && sl < kC2RowCl; ++sl) {
                    double v[ND];
    #pragma unroll
                    for (int i = 0; i < ND; ++i) v[i] = slot == sl ? P[i] : 0.0;
                    const double tot = row_total_n<ND>(v, lane, tile);
                    if (lane < ND) Gl[w][sl][lane][j] = tot;
                }
            }
        }
    } else {
        // the same from the precomputed Zhat (no room for a factor beside a 9 x 9 block)
        if (!flag && mine) {
            const bool act = lane < nb;
            const int cc = act ? a.bsr_col[b0 + lane] : row;
            const double* __restrict__ Zh = a.c_zhat + NV * ND * (size_t)cc;
#pragma unroll 1
            for (int j = 0; j < NV; ++j) {
                double x[ND], P[ND];
#pragma unroll
                for (int k = 0; k < ND; ++k) x[k] = Zh[ND * j + k];
#pragma unroll
                for (int i = 0; i < ND; ++i) {
                    double t = S[ND * i] * x[0];
#pragma unroll
                    for (int k = 1; k < ND; ++k) t = fma(S[ND * i + k], x[k], t);
                    P[i] = act ? t : 0.0;
                }
                for (int sl = 0; sl < n_rowcl && sl < kC2RowCl; ++sl) {
                    double v[ND];
#pragma unroll
                    for (int i = 0; i < ND; ++i) v[i] = slot == sl ? P[i] : 0.0;
                    const double tot = row_total_n<ND>(v, lane, tile);
                    if (lane < ND) Gl[w][sl][lane][j] = tot;
                }
            }
        }
    }
    if (!flag) {
        if (!collect(0, true)) flag = 2;
        else if (mine && lane < ND) {
            o_u = o_r + own_correction();
            o_w = qa0 + coarse_row();
        }
        poll.deadline = t_start + a.timeout_ticks;
    }
    while (!flag) {
        const int gen = it + 1, slotg = gen % kSlots;
        double* vcur = a.vec + (size_t)slotg * gen_stride;
        // the stores of the iteration before (the re-arming among them) have landed before
        // anything of this generation is stored
        __builtin_amdgcn_s_waitcnt(0);
        if (mine && lane < ND) st_shared(vcur + (size_t)row * ND + lane, o_w);
        const double tau = restrict_row(o_w);
        publish(o_r * o_u, o_w * o_u, tau, gen);
        // everybody has left generation gen - 2 (its records were collected an iteration ago):
        // re-arm this workgroup's words of that slot, which will take generation gen + 2
        if (gen >= 2) {
            double* arm = a.vec + (size_t)((gen - 2) % kSlots) * gen_stride;
            if (mine && lane < ND) st_word(reinterpret_cast<unsigned long long*>(arm + (size_t)row * ND) + lane, kSentinel);
            if (tid < kC2Rec) st_word(drec + kC2Rec * ((size_t)((gen - 2) % kSlots) * G + blockIdx.x) + tid, kSentinel);
        }
        // Shat w: the w of the row's columns, fetched cooperatively
        bool okw = true;
        for (int spins = 1;; ++spins) {
            bool ok = true;
            constexpr int kFlight = kLean ? 3 : ND;          // loads in flight (register budget)
#pragma unroll
            for (int t0 = 0; t0 < ND; t0 += kFlight) {
                double val[kFlight];
                int off[kFlight];
#pragma unroll
                for (int t = 0; t < kFlight; ++t) off[t] = off_lds[w][t0 + t][lane];
#pragma unroll
                for (int t = 0; t < kFlight; ++t)
                    if (off[t] >= 0) val[t] = ld_shared(vcur + off[t]);
#pragma unroll
                for (int t = 0; t < kFlight; ++t)
                    if (off[t] >= 0) { ok = ok && !is_sentinel(val[t]); tile[64 * (t0 + t) + lane] = val[t]; }
            }
            if (__all(ok)) break;
            if (poll.expired(spins)) { okw = false; break; }
        }
        double y[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) y[i] = 0.0;
        __builtin_amdgcn_wave_barrier();
        if (okw && lane < nb) {
            const double* u = tile + ND * lane;
#pragma unroll
            for (int j = 0; j < ND; ++j) {
                const double wj = u[j];
#pragma unroll
                for (int i = 0; i < ND; ++i) y[i] = j == 0 ? S[ND * i] * wj : fma(S[ND * i + j], wj, y[i]);
            }
        }
        const double qa = row_total_n<ND>(y, lane, tile);
        if (!collect(gen, okw)) { flag = 2; break; }
        if (it == 0) {
            gam0 = gam;
            if (!(gam > 0.0)) break;                        // zero (or NaN) right-hand side
        } else {
            const bool small_gain = a.q_abs >= 0.0 && !(gam > a.tol2_loose * gam0) && dq1 <= a.q_abs && dq2 <= a.q_abs;
            if (small_gain && gam > a.tol2 * gam0) tol2_met = a.tol2_loose;
            if (!(gam > a.tol2 * gam0) || small_gain || it >= a.max_iter) break;
        }
        double beta = it ? gam / gam_prev : 0.0;
        if constexpr (kLean) beta = to_scalar(beta);
        const double den = del - (it ? beta * gam / alpha_prev : 0.0);
        if (!(den > 0.0)) { flag = 1; break; }               // breakdown: caller falls back
        double alpha = gam / den;
        if constexpr (kLean) alpha = to_scalar(alpha);
        dq2 = dq1;
        dq1 = 0.5 * alpha * gam;             // what this iteration gains in the quadratic model
        gam_prev = gam;
        alpha_prev = alpha;
        if (mine && lane < ND) {
            const double m = o_w + own_correction();
            const double n = qa + coarse_row();
            if constexpr (kLean) { o_z = own[lane]; o_q = own[16 + lane]; o_s = own[32 + lane]; o_p = own[48 + lane]; o_x = own[64 + lane]; }
            o_z = fma(beta, o_z, n);
            o_q = fma(beta, o_q, m);
            o_s = fma(beta, o_s, o_w);
            o_p = fma(beta, o_p, o_u);
            o_x = fma(alpha, o_p, o_x);
            o_r = fma(-alpha, o_s, o_r);
            o_u = fma(-alpha, o_q, o_u);
            o_w = fma(-alpha, o_z, o_w);
            if constexpr (kLean) { own[lane] = o_z; own[16 + lane] = o_q; own[32 + lane] = o_s; own[48 + lane] = o_p; own[64 + lane] = o_x; }
        }
        ++it;
    }
    // x = F' xhat for the row this wavefront owns
    if (flag != 2 && mine) {
        double val = 0.0;
#pragma unroll
        for (int j = 0; j < ND; ++j) {
            const double xj = kLean ? own[64 + j] : __shfl(o_x, j, 64);
            if (lane < ND) val += a.F[NN * (size_t)row + ND * j + lane] * xj;
        }
        if (lane < ND) a.x[ND * (size_t)row + lane] = val;
    }
    if (blockIdx.x == 0 && tid == 0) {
        PcgState st;
        st.rz = gam; st.rz0 = gam0; st.tol2 = tol2_met; st.tol2_loose = a.tol2_loose; st.q_abs = a.q_abs; st.dq1 = dq1; st.dq2 = dq2; st.iter = it; st.max_iter = a.max_iter; st.done = 1; st.pad = flag;
        st.ticks = (long long)wall_clock64() - t_start;
        *a.out = st;
        if (a.flag_slot) *a.flag_slot = (double)flag;
        if (a.flag_host) *a.flag_host = (double)flag;
    }
}

int pcg_fused_max_blocks(int device) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pcg_fused, kFusedThreads, 0) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    // one workgroup per compute unit; every workgroup's record is polled by one thread
    return per_cu > 0 ? (cus < kFusedThreads ? cus : kFusedThreads) : 0;
}

int pcg_fused_rows_per_block() { return kFusedWaves; }
size_t pcg_fused_record_words(int blocks) { return kPcgRecWords * kSlots * (size_t)blocks + 1; }
size_t pcg_fused_vector_words(int n_cam) { return kSlots * (size_t)n_cam * kVecRec; }

// Arms every record (all bits set) and the abort word unless that has been done, then launches
// the solve.  The launch is a plain one and the kernel relies on its admission timeout: a grid
// that does not become fully resident within admit_ticks aborts and the caller falls back.
// Zhat_c = F_c^-T Z_c of every camera, [n_cam][7][ND]: thread = (camera, coarse vector).  For the
// blocks too large for k_pcg_pipe2 to do this itself beside the block it holds.
template <int ND>
__global__ __launch_bounds__(kBlock) void k_coarse_zhat(FusedPcg a, double* __restrict__ out) {
    constexpr int NV = kCoarseVecs;
    const int t = blockIdx.x * kBlock + threadIdx.x;
    if (t >= a.n_cam * NV) return;
    const int c = t / NV;
    const double* __restrict__ F = a.F + ND * ND * (size_t)c;
    const double* __restrict__ z = a.c_z + (size_t)t * ND;
    double x[ND];
#pragma unroll
    for (int k = ND - 1; k >= 0; --k) {
        double v = z[k];
#pragma unroll
        for (int l = k + 1; l < ND; ++l) v -= F[ND * l + k] * x[l];
        x[k] = v / F[ND * k + k];
    }
#pragma unroll
    for (int k = 0; k < ND; ++k) out[(size_t)t * ND + k] = x[k];
}

hipError_t launch_pcg_fused(const FusedPcg& f, bool armed, hipStream_t st) {
    if (!armed) {
        hipError_t e = hipMemsetAsync(f.dots, 0xFF, pcg_fused_record_words(f.blocks) * sizeof(unsigned long long), st);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(f.vec, 0xFF, pcg_fused_vector_words(f.n_cam) * sizeof(double), st);
        if (e != hipSuccess) return e;
    }
    FusedPcg arg = f;
    if (f.nd != 6 && !(f.nd == 9 && f.pipelined)) return hipErrorInvalidValue;
    if (f.coarse && !f.pipelined) return hipErrorInvalidValue;
    if (f.coarse && f.nd == 9) {
        if (!f.c_zhat) return hipErrorInvalidValue;
        hipLaunchKernelGGL((k_coarse_zhat<9>), dim3((f.n_cam * kCoarseVecs + kBlock - 1) / kBlock), dim3(kBlock), 0, st, arg, const_cast<double*>(f.c_zhat));
        hipLaunchKernelGGL((k_pcg_pipe2<9>), dim3(f.blocks), dim3(kFusedThreads), 0, st, arg);
    } else if (f.coarse) hipLaunchKernelGGL((k_pcg_pipe2<6>), dim3(f.blocks), dim3(kFusedThreads), 0, st, arg);
    else if (f.pipelined && f.nd == 9) hipLaunchKernelGGL((k_pcg_pipe<9>), dim3(f.blocks), dim3(kFusedThreads), 0, st, arg);
    else if (f.pipelined) hipLaunchKernelGGL((k_pcg_pipe<6>), dim3(f.blocks), dim3(kFusedThreads), 0, st, arg);
    else hipLaunchKernelGGL(k_pcg_fused, dim3(f.blocks), dim3(kFusedThreads), 0, st, arg);
    return hipGetLastError();
}

void launch_pcg_init(const Dev& d, double tol, int max_iter, hipStream_t st, double tol_loose, double q_abs) {
    if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_pcg_init<Bal::ND>), dim3(pcg_update_blocks(d)), dim3(kBlock), 0, st, d, tol, max_iter, tol_loose, q_abs);
    else hipLaunchKernelGGL((k_pcg_init<Pinhole::ND>), dim3(pcg_update_blocks(d)), dim3(kBlock), 0, st, d, tol, max_iter, tol_loose, q_abs);
}
void launch_pcg_spmv_bsr(const Dev& d, int cur, hipStream_t st) {
    if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_pcg_spmv_bsr<Bal::ND>), dim3(pcg_spmv_blocks(d)), dim3(kBlock), 0, st, d, cur);
    else hipLaunchKernelGGL((k_pcg_spmv_bsr<Pinhole::ND>), dim3(pcg_spmv_blocks(d)), dim3(kBlock), 0, st, d, cur);
}
void launch_pcg_dir(const Dev& d, int cur, hipStream_t st) {
    if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_pcg_dir<Bal::ND>), dim3(blocks_cam(d)), dim3(kBlock), 0, st, d, cur);
    else hipLaunchKernelGGL((k_pcg_dir<Pinhole::ND>), dim3(blocks_cam(d)), dim3(kBlock), 0, st, d, cur);
}
void launch_pcg_q(const Dev& d, double lambda, int cur, hipStream_t st) {
    if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_pcg_q<Bal::ND>), dim3(blocks_cam(d)), dim3(kBlock), 0, st, d, lambda, cur);
    else hipLaunchKernelGGL((k_pcg_q<Pinhole::ND>), dim3(blocks_cam(d)), dim3(kBlock), 0, st, d, lambda, cur);
}
void launch_pcg_dot(const Dev& d, int cur, hipStream_t st) {
    if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_pcg_dot<Bal::ND>), dim3(pcg_update_blocks(d)), dim3(kBlock), 0, st, d, cur);
    else hipLaunchKernelGGL((k_pcg_dot<Pinhole::ND>), dim3(pcg_update_blocks(d)), dim3(kBlock), 0, st, d, cur);
}
void launch_pcg_update(const Dev& d, int cur, int n_pq_parts, hipStream_t st, const Coarse* coarse) {
    const dim3 grid(pcg_update_blocks(d)), blk(kBlock);
    if (coarse) {
        if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_pcg_update<Bal::ND, true>), grid, blk, 0, st, d, cur, n_pq_parts, coarse->Z, coarse->cam_cl, coarse->rpart);
        else hipLaunchKernelGGL((k_pcg_update<Pinhole::ND, true>), grid, blk, 0, st, d, cur, n_pq_parts, coarse->Z, coarse->cam_cl, coarse->rpart);
    } else {
        if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_pcg_update<Bal::ND, false>), grid, blk, 0, st, d, cur, n_pq_parts, nullptr, nullptr, nullptr);
        else hipLaunchKernelGGL((k_pcg_update<Pinhole::ND, false>), grid, blk, 0, st, d, cur, n_pq_parts, nullptr, nullptr, nullptr);
    }
}

}  // namespace sfmba
