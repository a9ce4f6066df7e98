// HIP kernels of libsfmba for gfx950 (MI355X).  fp64 throughout.
//
// Two observation orders are kept in HBM.  Point-major ("pm"): the observations of one point
// are contiguous, one thread owns one point and keeps its 3x3 block and 3-vector sums in
// registers.  Camera-major ("cm"): the observations of one camera are contiguous and cut into
// chunks of at most kChunk; one 64-lane wavefront owns one chunk, the camera is wave-uniform,
// per-observation camera-sized vector / block contributions (6 or 9 wide, by camera model) are
// summed in registers, reduced across the
// wavefront with cross-lane shuffles and written as one partial row per chunk; a per-camera
// pass adds the chunk rows in a fixed order, so every sum is bitwise reproducible.
//
// What the kernels replace in the reference: the per-observation residual loop
// (/root/reference/bundle_adjustment.py:22-30,49-50), SciPy's finite-difference Jacobian over
// the jac_sparsity pattern (bundle_adjustment.py:53-74,139-141) and the LSMR trust-region step
// (scipy/optimize/_lsq/trf.py:471-499).
#include <cstdlib>

#include "sfmba_internal.h"
#ifndef SFMBA_BACK_BLOCKS
#define SFMBA_BACK_BLOCKS 512
#endif

namespace sfmba {

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static int blocks_chunk(const Dev& d) { return cdiv(d.n_chunks, kCmBlock / 64); }

// Store one block partial of a scalar slot.
__device__ __forceinline__ void put_partial(const Dev& d, int slot, double v) {
    d.scal_part[(size_t)slot * kMaxScalBlocks + blockIdx.x] = v;
}

// Every observation kernel is a template over the camera model (sfmba_device.h) and launched for
// the model of the problem.
#define SFMBA_LAUNCH_T(d, KERNEL, grid, block, st, ...)                                               \
    do {                                                                                              \
        if ((d).model == MODEL_BAL) hipLaunchKernelGGL((KERNEL<Bal>), grid, block, 0, st, __VA_ARGS__);          \
        else hipLaunchKernelGGL((KERNEL<Pinhole>), grid, block, 0, st, __VA_ARGS__);                 \
    } while (0)
#define SFMBA_LAUNCH(d, KERNEL, grid, st, ...)                                                        \
    do {                                                                                              \
        if ((d).model == MODEL_BAL) hipLaunchKernelGGL((KERNEL<Bal>), grid, dim3(kBlock), 0, st, __VA_ARGS__);   \
        else hipLaunchKernelGGL((KERNEL<Pinhole>), grid, dim3(kBlock), 0, st, __VA_ARGS__);          \
    } while (0)

// The ND tangent numbers of camera c out of a camera-sized vector.
template <int ND>
__device__ __forceinline__ void load_tangent(const double* __restrict__ x, int c, double (&dd)[ND]) {
    if constexpr (ND % 2 == 0) {
        const double2* xc = reinterpret_cast<const double2*>(x + ND * (size_t)c);
#pragma unroll
        for (int i = 0; i < ND / 2; ++i) { const double2 v = xc[i]; dd[2 * i] = v.x; dd[2 * i + 1] = v.y; }
    } else {
#pragma unroll
        for (int i = 0; i < ND; ++i) dd[i] = x[ND * (size_t)c + i];
    }
}

// ------------------------------------------------------------------------------------------
// fun() parity: residuals in the caller's observation order.
template <class M>
__global__ __launch_bounds__(kBlock) void k_residuals(Dev d, double* __restrict__ out) {
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= d.n_obs) return;
    double4 X = d.X[d.pm_pt[j]];
    const typename M::Cam k = M::load(d.cams, d.pm_cam[j]);
    const typename M::Obs ob = M::obs(M::seed(k, X.x, X.y, X.z), k, d.K);
    double2 uv = d.pm_uv[j];
    double ru, rv;
    M::residual(ob, uv.x, uv.y, ru, rv);
    size_t o = 2 * (size_t)d.pm_orig[j];
    out[o] = ru;
    out[o + 1] = rv;
}

template <class M>
__global__ __launch_bounds__(kBlock) void k_jacobian_blocks(Dev d, double* __restrict__ jc_out,
                                                            double* __restrict__ jp_out) {
    constexpr int ND = M::ND;
    int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= d.n_obs) return;
    double4 X = d.X[d.pm_pt[j]];
    const typename M::Cam k = M::load(d.cams, d.pm_cam[j]);
    const typename M::Obs ob = M::obs(M::seed(k, X.x, X.y, X.z), k, d.K);
    double jc[2 * ND], a[3], b[3];
    M::jac_cam(ob, jc);
    M::jac_pt(ob, k.q, a, b);
    size_t o = (size_t)d.pm_orig[j];
#pragma unroll
    for (int i = 0; i < 2 * ND; ++i) jc_out[2 * ND * o + i] = jc[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { jp_out[6 * o + i] = a[i]; jp_out[6 * o + 3 + i] = b[i]; }
}

// ------------------------------------------------------------------------------------------
// A wave tile: a range of point-major observations made of whole points, at most 64 of them
// unless it is one single point with more (then the wave strides over it).  Lane l works on
// observation beg + l, so every stream is read and written fully coalesced, and per-point sums
// are segmented reductions over the lanes of a point.
struct Tile { int beg, end; bool valid, is_long; };
// consecutive tiles that make one job of k_lin_pm: about fifty points, so that the phase with a lane
// per point keeps most lanes at work, and 16 KB of LDS per wavefront for the values in between
constexpr int kTilesPerWave = 4;
__device__ __forceinline__ Tile wave_tile(const Dev& d, int tpw = 1, int rep = 0) {
    const int q = (xcd_block(blockIdx.x, gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6)) * tpw + rep;
    Tile t;
    t.valid = q < d.n_wtiles;
    t.beg = t.valid ? __builtin_amdgcn_readfirstlane(d.wtile_beg[q]) : 0;
    t.end = t.valid ? __builtin_amdgcn_readfirstlane(d.wtile_beg[q + 1]) : 0;
    t.is_long = t.end - t.beg > 64;
    return t;
}

// Cameras of one wave tile staged in LDS.  The lanes of a tile hold the observations of a few
// neighbouring points, and points are numbered by their smallest observing camera, so a tile's
// cameras usually lie in a short index range starting at the tile's smallest one.  A per-lane
// gather of a camera record costs the vector cache one access per lane and 16-byte piece (six for
// a pinhole camera, six more for the trial camera, three for the step); the same records fetched
// once per wavefront as one contiguous range cost a sixteenth of that, and the lanes then read
// their camera from LDS.  A lane whose camera lies outside the staged range gathers as before.
// (Measured at the Venice shape: k_back_pm, which gathers fifteen pieces per lane, 159 -> 138 us per
// launch; k_lin_pm, which gathers six, loses 6 % to the extra barrier and is left as it was.)
constexpr int kStageCams = 16;
template <class M>
struct CamStage {
    double2 cam[kStageCams * M::CW];
    double2 trial[kStageCams * M::CW];
    double dc[kStageCams * M::ND];
};
__device__ __forceinline__ int wave_min_int(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
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
// kGrad: only g_p's largest entry and the cost are wanted (the gradient norm at the point a solve
// ends on): V_p is neither summed nor stored, nothing is damped.
template <class M, bool kStore, bool kGrad>
__device__ __forceinline__ void lin_pm_tile_scan(const Dev& d, const Tile& t, int lane, double lambda, int keep_v,
                                                 double& cost, double& gmax) {
    double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int key = -1 - lane;
    for (int j = t.beg + lane; j < t.end; j += 64) {
        const int p = d.pm_pt[j];
        key = p;
        const double4 X = d.X[p];
        const typename M::Cam k = M::load(d.cams, d.pm_cam[j]);
        const Seed s = M::seed(k, X.x, X.y, X.z);
        const typename M::Obs ob = M::obs(s, k, d.K);
        const double2 uv = d.pm_uv[j];
        double ru, rv;
        M::residual(ob, uv.x, uv.y, ru, rv);
        if (kStore) {
            d.pm_seed[j] = make_double4(s.x, s.y, s.iz, ru);
            d.pm_rv[j] = rv;
        }
        double a[3], b[3];
        M::jac_pt(ob, k.q, a, b);
        if (!kGrad) {
            acc[0] += a[0] * a[0] + b[0] * b[0];
            acc[1] += a[0] * a[1] + b[0] * b[1];
            acc[2] += a[0] * a[2] + b[0] * b[2];
            acc[3] += a[1] * a[1] + b[1] * b[1];
            acc[4] += a[1] * a[2] + b[1] * b[2];
            acc[5] += a[2] * a[2] + b[2] * b[2];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) acc[6 + i] += a[i] * ru + b[i] * rv;
        cost += ru * ru + rv * rv;
    }
    bool writer;
    if (kGrad) {
        double g3[3] = {acc[6], acc[7], acc[8]};
        if (t.is_long) {
#pragma unroll
            for (int i = 0; i < 3; ++i) g3[i] = wave_sum(g3[i]);
            writer = lane == 0;
        } else {
            seg_scan(g3, key, lane);
            const int last = seg_last(key, lane);
            writer = key >= 0 && last == lane;
        }
        if (writer) gmax = fmax(gmax, fmax(fabs(g3[0]), fmax(fabs(g3[1]), fabs(g3[2]))));
        writer = false;
    } else if (t.is_long) {
#pragma unroll
        for (int i = 0; i < 9; ++i) acc[i] = wave_sum(acc[i]);
        writer = lane == 0;
    } else {
        seg_scan(acc, key, lane);
        const int last = seg_last(key, lane);      // every lane takes part in the shuffles
        writer = key >= 0 && last == lane;
    }
    if (writer) {
        // V_p and g_p themselves are only read back to damp again after a rejected step (and by the
        // diagnostic entry points): inside a solve they stay in registers and the rare rejected
        // step linearises again
        if (keep_v) {
            double* Vo = d.V + 6 * (size_t)key;
#pragma unroll
            for (int i = 0; i < 6; ++i) Vo[i] = acc[i];
            double* go = d.gp + 4 * (size_t)key;
            go[0] = acc[6]; go[1] = acc[7]; go[2] = acc[8]; go[3] = 0.0;
        }
        gmax = fmax(gmax, fmax(fabs(acc[6]), fmax(fabs(acc[7]), fabs(acc[8]))));
        if (lambda > 0.0) damp_point(d, key, acc, acc + 6, lambda);
    }
}

// What a wavefront of k_lin_pm keeps in LDS between its two phases: per observation of its
// kTilesPerWave tiles the two rows of the point Jacobian and the residual (value v of slot s at
// val[v][s], so that the lanes of a store touch consecutive words), and the table of its points
// (first slot and point index; start[n] = number of slots).
template <class M>
struct LinStage {
    double val[8][64 * kTilesPerWave];
    double2 cam[kStageCams * M::CW];          // the cameras [c_min, c_min + kStageCams) of the wavefront's observations
    unsigned int start[64 * kTilesPerWave + 2];   // first slot of a point (low 9 bits) and its index less the job's first point
};

// One job of k_lin_pm: kTilesPerWave consecutive wave tiles, i.e. one range [beg, end) of point-major
// observations made of whole points.  Its operands arrive in two steps, each requested a job ahead of
// its use: the index streams (lane l holds observation beg + 64 r + l in round r), then what the
// indices point at.
struct LinStreams {
    int beg, end;
    bool any_long;          // holds a tile of more than 64 observations (one long point): done tile by tile
    int pr[kTilesPerWave], cr[kTilesPerWave];
};
template <class M>
struct LinGather {
    double4 X[kTilesPerWave];
    double2 uv[kTilesPerWave];
    double2 cam[2];         // words lane and lane + 64 of the cameras [c0, c0 + nst), on their way to the LDS stage
    int c0, nst;
};
static_assert(kStageCams * 8 <= 128, "two words of the camera piece per lane");
// (Every request below is issued by all lanes, on an index clamped into its array, and the unused
// results are replaced afterwards: loads inside branches would make the compiler wait for all
// outstanding loads wherever it waits for one, and the pipeline over the jobs would be gone.)
// ranges: lane l holds the first observation of job rbase + l (the boundaries of up to 63 jobs, read
// once per wavefront: a load per job would have to be waited for at once, and with it every request
// in flight).
__device__ __forceinline__ int lin_ranges_load(const Dev& d, int rbase, int lane) {
    return d.wtile_beg[(int)min((long long)(rbase + lane) * kTilesPerWave, (long long)d.n_wtiles)];
}
__device__ __forceinline__ LinStreams lin_streams_fetch(const Dev& d, int jb, int jb_end, int lane, bool streams,
                                                        int ranges, int rbase) {
    LinStreams j;
    j.beg = j.end = 0;
    j.any_long = false;
    if (jb < jb_end) {
        const int q0 = jb * kTilesPerWave, q1 = min(q0 + kTilesPerWave, d.n_wtiles);
        j.beg = __shfl(ranges, jb - rbase, 64);
        j.end = __shfl(ranges, jb + 1 - rbase, 64);
        j.beg = __builtin_amdgcn_readfirstlane(j.beg);
        j.end = __builtin_amdgcn_readfirstlane(j.end);
        j.any_long = j.end - j.beg > 64 * (q1 - q0);
    }
    const bool want = streams && !j.any_long;
#pragma unroll
    for (int r = 0; r < kTilesPerWave; ++r) {
        const int o = j.beg + 64 * r + lane;
        const int oc = min(o, d.n_obs - 1);
        const int p = d.pm_pt[oc], c = d.pm_cam[oc];
        const bool live = want && o < j.end;
        j.pr[r] = live ? p : -2;
        j.cr[r] = live ? c : 0x7fffffff;
    }
    return j;
}
template <class M>
__device__ __forceinline__ LinGather<M> lin_gather_fetch(const Dev& d, const LinStreams& j, int lane) {
    LinGather<M> g;
    int c0 = 0x7fffffff;
#pragma unroll
    for (int r = 0; r < kTilesPerWave; ++r) c0 = min(c0, j.cr[r]);
    c0 = wave_min_int(c0);
    const bool any = c0 != 0x7fffffff;           // (an empty or long job: nothing is used)
    g.c0 = any ? c0 : 0;
    g.nst = any ? min(kStageCams, d.n_cam - c0) : 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = min(lane + 64 * q, max(g.nst, 1) * M::CW - 1);
        g.cam[q] = d.cams[(size_t)g.c0 * M::CW + i];
    }
#pragma unroll
    for (int r = 0; r < kTilesPerWave; ++r) {
        g.X[r] = d.X[max(j.pr[r], 0)];
        g.uv[r] = d.pm_uv[min(j.beg + 64 * r + lane, d.n_obs - 1)];
    }
    return g;
}

// The wavefronts of the grid share the jobs evenly (the grid is sized to what the chip holds at once,
// lin_pm_blocks): wavefront w of W takes jobs [w J / W, (w + 1) J / W), neighbouring wavefronts
// neighbouring points.  A wavefront works in two phases per job.  Phase one, a lane per
// observation: projection, residual and point Jacobian go to the wavefront's LDS stage, the first
// observation of every point enters the point table.  Phase two, a lane per point: the nine sums
// over the point's observations in their order, then damping, the 3x3 inverse and s0 - every lane
// at work and no cross-lane traffic (a segmented scan over the lanes of the observations costs four
// rounds of shuffles for nine values, and only the one lane in five that ends a point then works on
// the inverse).  The stage leaves two wavefronts per SIMD, so latency is covered inside the
// wavefront by a pipeline over its jobs: while job i is worked on, the points, image points and
// cameras of job i + 1 and the index streams of job i + 2 are on their way.  The cameras of a job lie
// in a short index range because points are numbered by their smallest camera: they are fetched
// once per wavefront as one contiguous piece and handed to the lanes through LDS (a camera outside
// the piece is gathered).
template <class M, bool kStore, bool kGrad = false>
__global__ __launch_bounds__(kBlock) void k_lin_pm(Dev d, double lambda, int keep_v) {
    __shared__ double red[block_reduce_scratch<2>()];
    __shared__ __attribute__((aligned(16))) LinStage<M> stage_all[kGrad ? 1 : kBlock / 64];
    const int lane = threadIdx.x & 63;
    double cost = 0.0, gmax = 0.0;
    const int n_jobs = (d.n_wtiles + kTilesPerWave - 1) / kTilesPerWave;
    const int w = xcd_block(blockIdx.x, gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6);
    const int W = gridDim.x * (kBlock / 64);
    const int jb0 = (int)((long long)w * n_jobs / W), jb1_all = (int)((long long)(w + 1) * n_jobs / W);
    LinStage<M>& sg = stage_all[kGrad ? 0 : threadIdx.x >> 6];
    // (the job boundaries of a wavefront fit one register 63 jobs at a time: the pipeline runs over
    // such stretches; only a wavefront of a very large problem has more than one)
#pragma unroll 1
    for (int rbase = jb0; rbase < jb1_all; rbase += 63) {
    const int jb1 = min(rbase + 63, jb1_all);
    const int ranges = lin_ranges_load(d, rbase, lane);
    LinStreams s1 = lin_streams_fetch(d, rbase, jb1, lane, !kGrad, ranges, rbase);
    LinStreams s2 = lin_streams_fetch(d, rbase + 1, jb1, lane, !kGrad, ranges, rbase);
    LinGather<M> g1 = lin_gather_fetch<M>(d, s1, lane);
#pragma unroll 1
    for (int jb = rbase; jb < jb1; ++jb) {
        const LinStreams cu = s1;
        const LinGather<M> gc = g1;
        s1 = s2;
        s2 = lin_streams_fetch(d, jb + 2, jb1, lane, !kGrad, ranges, rbase);
        g1 = lin_gather_fetch<M>(d, s1, lane);
        // a point index has to fit beside its slot in a word of the point table (it does unless
        // millions of unobserved points lie between the points of one job)
        int p_hi = -1;
#pragma unroll
        for (int r = 0; r < kTilesPerWave; ++r) p_hi = max(p_hi, cu.pr[r]);
        const int p_first = __builtin_amdgcn_readfirstlane(cu.pr[0]);
        const bool wide = __any(p_hi - p_first >= (1 << 23));
        if (kGrad || cu.any_long || wide) {
#pragma unroll 1
            for (int q = jb * kTilesPerWave; q < min((jb + 1) * kTilesPerWave, d.n_wtiles); ++q) {
                Tile t;
                t.valid = true;
                t.beg = __builtin_amdgcn_readfirstlane(d.wtile_beg[q]);
                t.end = __builtin_amdgcn_readfirstlane(d.wtile_beg[q + 1]);
                t.is_long = t.end - t.beg > 64;
                lin_pm_tile_scan<M, kStore, kGrad>(d, t, lane, lambda, keep_v, cost, gmax);
            }
            continue;
        }
        const int beg0 = cu.beg, end0 = cu.end;
        const int c0 = gc.c0, nst = gc.nst;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            if (lane + 64 * q < nst * M::CW) sg.cam[lane + 64 * q] = gc.cam[q];
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        int n_points = 0;
        int carry = -1;                       // point of the observation before this round's first
        bool outside = false;                 // some camera of the job lies outside the staged piece
#pragma unroll
        for (int r = 0; r < kTilesPerWave; ++r) outside = outside || (cu.pr[r] >= 0 && cu.cr[r] - c0 >= nst);
        outside = __any(outside);
#pragma unroll
        for (int r = 0; r < kTilesPerWave; ++r) {
            const int slot = 64 * r + lane;
            const int j = beg0 + slot;
            const bool live = j < end0;
            const int p = cu.pr[r];
            if (live) {
                const double4 X = gc.X[r];
                const int rel = cu.cr[r] - c0;
                typename M::Cam k = M::load(sg.cam, min(rel, nst - 1));
                if (outside) {                  // (wave-uniform: a job whose cameras all lie in the piece has no load here)
                    if (rel >= nst) k = M::load(d.cams, cu.cr[r]);
                }
                const Seed s = M::seed(k, X.x, X.y, X.z);
                const typename M::Obs ob = M::obs(s, k, d.K);
                const double2 uv = gc.uv[r];
                double ru, rv;
                M::residual(ob, uv.x, uv.y, ru, rv);
                if (kStore) {
                    d.pm_seed[j] = make_double4(s.x, s.y, s.iz, ru);
                    d.pm_rv[j] = rv;
                }
                double a[3], b[3];
                M::jac_pt(ob, k.q, a, b);
#pragma unroll
                for (int i = 0; i < 3; ++i) { sg.val[i][slot] = a[i]; sg.val[3 + i][slot] = b[i]; }
                sg.val[6][slot] = ru;
                sg.val[7][slot] = rv;
                cost += ru * ru + rv * rv;
            }
            int prev = __shfl_up(p, 1, 64);
            if (lane == 0) prev = carry;
            carry = __shfl(p, 63, 64);
            const bool first = live && p != prev;
            const unsigned long long m = __ballot(first);
            if (first) {
                const int rank = n_points + __builtin_popcountll(m & ((1ull << lane) - 1ull));
                sg.start[rank] = (unsigned)slot | ((unsigned)(p - p_first) << 9);
            }
            n_points += __builtin_popcountll(m);
        }
        if (lane == 0) sg.start[n_points] = (unsigned)(end0 - beg0);
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        // (written as kTilesPerWave guarded passes, not as a loop: in front of a loop with stores the
        // compiler waits for every load in flight, the next job's requests among them)
#pragma unroll
        for (int pass = 0; pass < kTilesPerWave; ++pass) {
            if (64 * pass >= n_points) break;
            const int pp = 64 * pass + lane;
            const bool active = pp < n_points;
            const unsigned word = active ? sg.start[pp] : 0u;
            const int s0 = (int)(word & 511u);
            const int len = active ? (int)(sg.start[pp + 1] & 511u) - s0 : 0;
            // (no load here: it would be younger than the requests for the next job and waiting for
            // it would wait for all of them)
            const int key = p_first + (int)(word >> 9);
            double acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int k = 0; __any(k < len); ++k) {
                if (k < len) {
                    const int s = s0 + k;
                    const double a0 = sg.val[0][s], a1 = sg.val[1][s], a2 = sg.val[2][s];
                    const double b0 = sg.val[3][s], b1 = sg.val[4][s], b2 = sg.val[5][s];
                    const double ru = sg.val[6][s], rv = sg.val[7][s];
                    acc[0] += a0 * a0 + b0 * b0;
                    acc[1] += a0 * a1 + b0 * b1;
                    acc[2] += a0 * a2 + b0 * b2;
                    acc[3] += a1 * a1 + b1 * b1;
                    acc[4] += a1 * a2 + b1 * b2;
                    acc[5] += a2 * a2 + b2 * b2;
                    acc[6] += a0 * ru + b0 * rv;
                    acc[7] += a1 * ru + b1 * rv;
                    acc[8] += a2 * ru + b2 * rv;
                }
            }
            if (active) {
                // V_p and g_p themselves are only read back to damp again after a rejected step (and
                // by the diagnostic entry points): inside a solve they stay in registers and the rare
                // rejected step linearises again
                if (keep_v) {
                    double* Vo = d.V + 6 * (size_t)key;
#pragma unroll
                    for (int i = 0; i < 6; ++i) Vo[i] = acc[i];
                    double* go = d.gp + 4 * (size_t)key;
                    go[0] = acc[6]; go[1] = acc[7]; go[2] = acc[8]; go[3] = 0.0;
                }
                gmax = fmax(gmax, fmax(fabs(acc[6]), fmax(fabs(acc[7]), fabs(acc[8]))));
                if (lambda > 0.0) damp_point(d, key, acc, acc + 6, lambda);
            }
        }
        __builtin_amdgcn_wave_barrier();          // the next job's stores to the stage stay behind these reads
    }
    }
    static_assert(SLOT_GMAX_PT == SLOT_COST + 1, "slot order");
    const double vals[2] = {cost, gmax};
    const double r = block_reduce_n<2>(vals, 2u, red);
    if (threadIdx.x < 2) put_partial(d, SLOT_COST + threadIdx.x, r);
}

// The NV sums of a chunk, reduced over the wavefront through its LDS tile and written as the
// chunk's row of chunk_part.
template <int NV>
__device__ __forceinline__ void chunk_store(const Dev& d, int q, const double (&acc)[NV], double* lds, int lane) {
    double tot[(NV + 13) / 14];
    wave_sum_lds<NV, 14>(acc, lds + (threadIdx.x >> 6) * (64 * 16), lane, tot);
    if (lane < 14) {
        double* o = d.chunk_part + NV * (size_t)q;
#pragma unroll
        for (int p = 0; p < (NV + 13) / 14; ++p)
            if (14 * p + lane < NV) o[14 * p + lane] = tot[p];
    }
}

// Linearisation, camera-major half: seeds per observation (camera-major copy) and the chunk
// partials of U_c (packed upper) and g_c.
// kGrad: only g_c is wanted (see k_lin_pm): the U part of the chunk row is written as zeros and the
// camera-major seeds are left alone.
template <class M, bool kGrad = false>
__global__ __launch_bounds__(kCmBlock) void k_lin_cm(Dev d) {
    constexpr int ND = M::ND, NU = M::NU, NV = M::NV;
    __shared__ __attribute__((aligned(16))) double lds[(kCmBlock / 64) * 64 * 16];
    const int lane = threadIdx.x & 63;
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kCmBlock / 64) + (threadIdx.x >> 6);
    if (q >= d.n_chunks) return;
    const int c = __builtin_amdgcn_readfirstlane(d.chunk_cam[q]);
    const int k0 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q]);
    const int k1 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q + 1]);
    const typename M::Cam cam = M::load(d.cams, c);
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;
    // (9-number camera, two wavefronts per SIMD: the point index of the next round is requested
    // while this round's gather is in flight)
    constexpr bool kAhead = ND > 6;
    int pn = kAhead && k0 + lane < k1 ? d.cm_pt[k0 + lane] : 0;
    for (int k = k0 + lane; k < k1; k += 64) {
        int p;
        if constexpr (kAhead) { p = pn; if (k + 64 < k1) pn = d.cm_pt[k + 64]; }
        else p = d.cm_pt[k];
        double4 X = d.X[p];
        const Seed s = M::seed(cam, X.x, X.y, X.z);
        const typename M::Obs ob = M::obs(s, cam, d.K);
        double2 uv = d.cm_uv[k];
        double ru, rv;
        M::residual(ob, uv.x, uv.y, ru, rv);
        if (!kGrad) { d.cm_sx[k] = s.x; d.cm_sy[k] = s.y; d.cm_siz[k] = s.iz; }
        double jc[2 * ND];
        M::jac_cam(ob, jc);
        if (!kGrad) {
            // (pinhole model: jc[4] and jc[ND + 3] are zero, products with them are left out)
            constexpr bool kZ = M::kDiagonalPixelMap;
            int n = 0;
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int j = i; j < ND; ++j) {
                    const bool no_u = kZ && (i == 4 || j == 4), no_v = kZ && (i == 3 || j == 3);
                    if (!(no_u && no_v))
                        acc[n] += no_u ? jc[ND + i] * jc[ND + j] : no_v ? jc[i] * jc[j] : jc[i] * jc[j] + jc[ND + i] * jc[ND + j];
                    ++n;
                }
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) acc[NU + i] += jc[i] * ru + jc[ND + i] * rv;
    }
    if (kGrad) {
        double g[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) g[i] = wave_sum(acc[NU + i]);
        double* o = d.chunk_part + NV * (size_t)q;
        for (int i = lane; i < NU; i += 64) o[i] = 0.0;
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < ND; ++i) o[NU + i] = g[i];
        }
    } else {
        chunk_store<NV>(d, q, acc, lds, lane);
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

// Camera-major: chunk partials of T_c = sum W Vinv W' (packed upper) and sum W s0,
// W = Jc' Jp (ND x 3), formed as Jc' (Jp Vinv Jp') Jc without the ND x 3 intermediates.
template <class M>
__global__ __launch_bounds__(kCmBlock) void k_sdiag_cm(Dev d) {
    constexpr int ND = M::ND, NU = M::NU, NV = M::NV;
    __shared__ __attribute__((aligned(16))) double lds[(kCmBlock / 64) * 64 * 16];
    const int lane = threadIdx.x & 63;
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kCmBlock / 64) + (threadIdx.x >> 6);
    if (q >= d.n_chunks) return;
    const int c = __builtin_amdgcn_readfirstlane(d.chunk_cam[q]);
    const int k0 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q]);
    const int k1 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q + 1]);
    const typename M::Cam cam = M::load(d.cams, c);
    double acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.0;
    constexpr bool kAhead = ND > 6;         // as in k_lin_cm
    int pn = kAhead && k0 + lane < k1 ? d.cm_pt[k0 + lane] : 0;
    for (int k = k0 + lane; k < k1; k += 64) {
        int p;
        if constexpr (kAhead) { p = pn; if (k + 64 < k1) pn = d.cm_pt[k + 64]; }
        else p = d.cm_pt[k];
        Seed s; s.x = d.cm_sx[k]; s.y = d.cm_sy[k]; s.iz = d.cm_siz[k];      // (stored by k_lin_cm)
        const typename M::Obs ob = M::obs(s, cam, d.K);
        const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)p);
        double2 v01 = vi[0], v23 = vi[1], v45 = vi[2];
        double4 s0 = d.s0[p];
        double jc[2 * ND], a[3], b[3];
        M::jac_cam(ob, jc);
        M::jac_pt(ob, cam.q, a, b);
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
        // (pinhole model: jc[4] and jc[ND + 3] are zero, products with them are left out)
        constexpr bool kZ = M::kDiagonalPixelMap;
        int n = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const double l0 = kZ && i == 3 ? jc[i] * m00 : kZ && i == 4 ? jc[ND + i] * m01 : jc[i] * m00 + jc[ND + i] * m01;
            const double l1 = kZ && i == 3 ? jc[i] * m01 : kZ && i == 4 ? jc[ND + i] * m11 : jc[i] * m01 + jc[ND + i] * m11;
#pragma unroll
            for (int j = i; j < ND; ++j)
                acc[n++] += kZ && j == 3 ? l0 * jc[j] : kZ && j == 4 ? l1 * jc[ND + j] : l0 * jc[j] + l1 * jc[ND + j];
        }
        // Jc' (Jp (Vinv g_p)): the right-hand-side contribution
        const double w0 = a[0] * s0.x + a[1] * s0.y + a[2] * s0.z;
        const double w1 = b[0] * s0.x + b[1] * s0.y + b[2] * s0.z;
#pragma unroll
        for (int i = 0; i < ND; ++i) acc[NU + i] += jc[i] * w0 + jc[ND + i] * w1;
    }
    chunk_store<NV>(d, q, acc, lds, lane);
}

// Both camera-major passes in one, for a linearisation whose damping is already known (the points
// were damped by k_lin_pm in the same pass, so Vinv and s0 are current) and whose reduced matrix is
// built explicitly.  What the diagonal block needs of U_c and T_c is their difference, formed per
// observation as Jc' (I - Jp Vinv Jp') Jc, plus the diagonal of U_c for the damping term; the
// chunk row holds NU + 3 ND sums:
//     [0, NU)            upper triangle: (U - T)_ij off the diagonal, U_ii on it
//     [NU, NU + ND)      g_c
//     [NU + ND, +ND)     T_ii
//     [NU + 2 ND, +ND)   sum W s0 (the point part of the reduced right-hand side)
// (198 registers, two wavefronts per SIMD; held to 168 and three wavefronts it runs no faster: the
// gathers of X, Vinv and s0 bound it, 150 us against 80 + 88 us for the two passes it replaces.)
// k_chunk_reduce_fused spreads them over UG and the T array so that every consumer of the two
// separate passes (k_finish_diag: S_cc = U* - T - B) finds what it expects: UG = (upper, g_c) as
// above, T = (T_ii on the diagonal, zero off it | sum W s0).
template <class M>
__global__ __launch_bounds__(kCmBlock) void k_lin_sdiag_cm(Dev d) {
    constexpr int ND = M::ND, NU = M::NU, NF = NU + 3 * ND;
    __shared__ __attribute__((aligned(16))) double lds[(kCmBlock / 64) * 64 * 16];
    const int lane = threadIdx.x & 63;
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kCmBlock / 64) + (threadIdx.x >> 6);
    if (q >= d.n_chunks) return;
    const int c = __builtin_amdgcn_readfirstlane(d.chunk_cam[q]);
    const int k0 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q]);
    const int k1 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q + 1]);
    const typename M::Cam cam = M::load(d.cams, c);
    double acc[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i] = 0.0;
    // software pipeline: the point index is requested two rounds ahead, the point's X, Vinv, s0 and
    // the observation one round ahead, so that a round's arithmetic overlaps the next round's
    // gathers (two wavefronts per SIMD leave the registers for it)
    const int kf = k0 + lane;
    int p1 = kf < k1 ? d.cm_pt[kf] : 0;
    int p2 = kf + 64 < k1 ? d.cm_pt[kf + 64] : 0;
    double4 nX = make_double4(0.0, 0.0, 0.0, 0.0), ns0 = nX;
    double2 n01 = make_double2(0.0, 0.0), n23 = n01, n45 = n01, nuv = n01;
    if (kf < k1) {
        nX = d.X[p1];
        const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)p1);
        n01 = vi[0]; n23 = vi[1]; n45 = vi[2];
        ns0 = d.s0[p1];
        nuv = d.cm_uv[kf];
    }
    for (int k = kf; k < k1; k += 64) {
        const double4 X = nX, s0 = ns0;
        const double2 v01 = n01, v23 = n23, v45 = n45, uv = nuv;
        p1 = p2;
        if (k + 128 < k1) p2 = d.cm_pt[k + 128];
        if (k + 64 < k1) {
            nX = d.X[p1];
            const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)p1);
            n01 = vi[0]; n23 = vi[1]; n45 = vi[2];
            ns0 = d.s0[p1];
            nuv = d.cm_uv[k + 64];
        }
        const Seed s = M::seed(cam, X.x, X.y, X.z);
        const typename M::Obs ob = M::obs(s, cam, d.K);
        double ru, rv;
        M::residual(ob, uv.x, uv.y, ru, rv);
        double jc[2 * ND], a[3], b[3];
        M::jac_cam(ob, jc);
        M::jac_pt(ob, cam.q, a, b);
        const double ta0 = v01.x * a[0] + v01.y * a[1] + v23.x * a[2];
        const double ta1 = v01.y * a[0] + v23.y * a[1] + v45.x * a[2];
        const double ta2 = v23.x * a[0] + v45.x * a[1] + v45.y * a[2];
        const double tb0 = v01.x * b[0] + v01.y * b[1] + v23.x * b[2];
        const double tb1 = v01.y * b[0] + v23.y * b[1] + v45.x * b[2];
        const double tb2 = v23.x * b[0] + v45.x * b[1] + v45.y * b[2];
        const double m00 = a[0] * ta0 + a[1] * ta1 + a[2] * ta2;
        const double m01 = a[0] * tb0 + a[1] * tb1 + a[2] * tb2;
        const double m11 = b[0] * tb0 + b[1] * tb1 + b[2] * tb2;
        const double p00 = 1.0 - m00, p01 = -m01, p11 = 1.0 - m11;
        // (pinhole model: jc[4] and jc[ND + 3] are zero, products with them are left out)
        constexpr bool kZ = M::kDiagonalPixelMap;
        int n = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const double l0 = kZ && i == 3 ? jc[i] * p00 : kZ && i == 4 ? jc[ND + i] * p01 : jc[i] * p00 + jc[ND + i] * p01;
            const double l1 = kZ && i == 3 ? jc[i] * p01 : kZ && i == 4 ? jc[ND + i] * p11 : jc[i] * p01 + jc[ND + i] * p11;
            const double uii = kZ && i == 3 ? jc[i] * jc[i] : kZ && i == 4 ? jc[ND + i] * jc[ND + i] : jc[i] * jc[i] + jc[ND + i] * jc[ND + i];
            const double tii = kZ && i == 3 ? jc[i] * jc[i] * m00 : kZ && i == 4 ? jc[ND + i] * jc[ND + i] * m11
                               : jc[i] * (jc[i] * m00 + jc[ND + i] * m01) + jc[ND + i] * (jc[i] * m01 + jc[ND + i] * m11);
            acc[n++] += uii;
            acc[NU + ND + i] += tii;
#pragma unroll
            for (int j = i + 1; j < ND; ++j)
                acc[n++] += kZ && j == 3 ? l0 * jc[j] : kZ && j == 4 ? l1 * jc[ND + j] : l0 * jc[j] + l1 * jc[ND + j];
        }
        const double w0 = a[0] * s0.x + a[1] * s0.y + a[2] * s0.z;
        const double w1 = b[0] * s0.x + b[1] * s0.y + b[2] * s0.z;
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            acc[NU + i] += jc[i] * ru + jc[ND + i] * rv;
            acc[NU + 2 * ND + i] += jc[i] * w0 + jc[ND + i] * w1;
        }
    }
    double tot[(NF + 13) / 14];
    wave_sum_lds<NF, 14>(acc, lds + (threadIdx.x >> 6) * (64 * 16), lane, tot);
    if (lane < 14) {
        double* o = d.chunk_part + NF * (size_t)q;
#pragma unroll
        for (int p = 0; p < (NF + 13) / 14; ++p)
            if (14 * p + lane < NF) o[14 * p + lane] = tot[p];
    }
}

// thread = (camera, entry of the UG row followed by the T row): chunk sums in fixed order
template <class M>
__global__ __launch_bounds__(kBlock) void k_chunk_reduce_fused(Dev d, double* __restrict__ tpart) {
    constexpr int ND = M::ND, NU = M::NU, NV = M::NV, NF = NU + 3 * ND;
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= d.n_cam * 2 * NV) return;
    const int c = i / (2 * NV), v = i - c * (2 * NV);
    int src = -1;                                   // column of the chunk rows, -1: a zero of the T row
    if (v < NV) src = v;
    else if (v - NV >= NU) src = NU + 2 * ND + (v - NV - NU);
    else {
#pragma unroll
        for (int r = 0; r < ND; ++r)
            if (v - NV == upidx(ND, r, r)) src = NU + ND + r;
    }
    double s = 0.0;
    if (src >= 0)
        for (int q = d.cam_chunk_ptr[c]; q < d.cam_chunk_ptr[c + 1]; ++q) s += d.chunk_part[(size_t)q * NF + src];
    if (v < NV) d.UG[NV * (size_t)c + v] = s;
    else tpart[NV * (size_t)c + (v - NV)] = s;
}

// Per camera: Minv = (U + lambda D - T)^-1, rhs = -g_c + T[NU..].
template <class M>
__global__ __launch_bounds__(kBlock) void k_precond(Dev d, double lambda) {
    constexpr int ND = M::ND, NU = M::NU, NV = M::NV;
    int c = blockIdx.x * kBlock + threadIdx.x;
    if (c < d.n_cam) {
        const double* U = d.UG + NV * (size_t)c;
        const double* T = d.T + NV * (size_t)c;
        double A[ND][ND], Ai[ND][ND], Ud[ND];
        int n = 0;
#pragma unroll
        for (int i = 0; i < ND; ++i)
#pragma unroll
            for (int j = i; j < ND; ++j) {
                double v = U[n] - T[n];
                if (i == j) { Ud[i] = U[n] + lambda * fmin(fmax(U[n], 1e-6), 1e32); v = Ud[i] - T[n]; }
                A[i][j] = v; A[j][i] = v;
                ++n;
            }
        bool ok = spd_inverse(A, Ai);
        double* o = d.Minv + ND * ND * (size_t)c;
        if (ok) {
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int j = 0; j < ND; ++j) o[ND * i + j] = Ai[i][j];
        } else {   // fall back to the damped diagonal of U
#pragma unroll
            for (int i = 0; i < ND; ++i)
#pragma unroll
                for (int j = 0; j < ND; ++j) o[ND * i + j] = (i == j) ? 1.0 / Ud[i] : 0.0;
        }
        const double* g = d.UG + NV * (size_t)c + NU;
        double* r = d.rhs + ND * (size_t)c;
#pragma unroll
        for (int i = 0; i < ND; ++i) r[i] = -g[i] + T[NU + i];
    }
}

// ------------------------------------------------------------------------------------------
// Schur matvec, point-major half: s_p = Vinv_p * sum_o Jp' (Jc x_c).
template <class M>
__global__ __launch_bounds__(kBlock) void k_spmv_pm(Dev d, const double* __restrict__ x, const PcgState* __restrict__ state) {
    constexpr int ND = M::ND;
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
        double dd[ND];
        load_tangent<ND>(x, c, dd);
        const typename M::Cam k = M::load_rot(d.cams, c);
        const typename M::Obs ob = M::obs(s, k, d.K);
        double w0, w1;
        M::jc_times(ob, dd, w0, w1);
        M::jpT_times_acc(ob, k.q, w0, w1, e);
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

// Schur matvec, camera-major half: chunk partials of sum_o Jc' (Jp s_p)  (ND values).
template <class M>
__global__ __launch_bounds__(kCmBlock) void k_spmv_cm(Dev d, const double4* __restrict__ svec, const PcgState* __restrict__ state) {
    constexpr int ND = M::ND;
    if (state && state->done) return;
    const int lane = threadIdx.x & 63;
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kCmBlock / 64) + (threadIdx.x >> 6);
    if (q >= d.n_chunks) return;
    const int c = __builtin_amdgcn_readfirstlane(d.chunk_cam[q]);
    const int k0 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q]);
    const int k1 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q + 1]);
    const typename M::Cam cam = M::load(d.cams, c);
    double acc[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) acc[i] = 0.0;
    for (int k = k0 + lane; k < k1; k += 64) {
        Seed s; s.x = d.cm_sx[k]; s.y = d.cm_sy[k]; s.iz = d.cm_siz[k];
        const typename M::Obs ob = M::obs(s, cam, d.K);
        double4 sp = svec[d.cm_pt[k]];
        double dd[3] = {sp.x, sp.y, sp.z};
        double w0, w1;
        M::jp_times(ob, cam.q, dd, w0, w1);
        M::jcT_times_acc(ob, w0, w1, acc);
    }
#pragma unroll
    for (int i = 0; i < ND; ++i) acc[i] = wave_sum(acc[i]);
    if (lane == 0) {
        double* o = d.chunk_part + ND * (size_t)q;
#pragma unroll
        for (int i = 0; i < ND; ++i) o[i] = acc[i];
    }
}

// ------------------------------------------------------------------------------------------
// Small dense helpers for the camera-sized vectors (the PCG kernels live in sfmba_pcg.hip).

// y = U* x - qS  (diagnostic: one application of the reduced operator).
template <class M>
__global__ __launch_bounds__(kBlock) void k_schur_apply(Dev d, const double* __restrict__ x,
                                                        double lambda, double* __restrict__ y) {
    constexpr int ND = M::ND;
    int c = blockIdx.x * kBlock + threadIdx.x;
    if (c >= d.n_cam) return;
    double p[ND], q[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) p[i] = x[ND * (size_t)c + i];
    ustar_times<ND>(d.UG + M::NV * (size_t)c, lambda, p, q);
#pragma unroll
    for (int i = 0; i < ND; ++i) y[ND * (size_t)c + i] = q[i] - d.qS[ND * (size_t)c + i];
}

// ------------------------------------------------------------------------------------------
// Back-substitution per point: dp = -Vinv (g_p + sum Jp' Jc dc); X_trial = X + dp; the cost of
// the linear model at the step; |dp|^2 and |X|^2; and, in the same pass, the cost at the trial
// point (X_trial with the retracted cameras, which k_cam_update must have written): every lane
// holds its observation, its point's step and gathers its camera anyway.  Seeds and residuals at
// the linearisation point are rebuilt from X and the camera rather than read back.
template <class M>
struct BackObs { typename M::Obs s; double ru, rv, w0, w1; typename M::Cam k; double2 uv; int c; };
template <class M>
__device__ __forceinline__ void back_finish(const Dev& d, BackObs<M>& o, const double (&dd)[M::ND], const double4& X) {
    o.s = M::obs(M::seed(o.k, X.x, X.y, X.z), o.k, d.K);
    M::residual(o.s, o.uv.x, o.uv.y, o.ru, o.rv);
    M::jc_times(o.s, dd, o.w0, o.w1);
}
template <class M>
__device__ __forceinline__ BackObs<M> back_load(const Dev& d, const double* __restrict__ dc, int j, const double4& X) {
    BackObs<M> o;
    o.c = d.pm_cam[j];
    o.uv = d.pm_uv[j];
    o.k = M::load(d.cams, o.c);
    double dd[M::ND];
    load_tangent<M::ND>(dc, o.c, dd);
    back_finish<M>(d, o, dd, X);
    return o;
}

// squared residual of observation o at the trial point; non-finite values are counted instead
template <class M>
__device__ __forceinline__ void trial_cost(const Dev& d, const BackObs<M>& o, const typename M::Cam& kt, const double4& X, const double* dp,
                                           double& cost, double& bad) {
    const typename M::Obs st = M::obs(M::seed(kt, X.x + dp[0], X.y + dp[1], X.z + dp[2]), kt, d.K);
    double ru, rv;
    M::residual(st, o.uv.x, o.uv.y, ru, rv);
    const double e = ru * ru + rv * rv;
    if (isfinite(e)) cost += e; else bad += 1.0;
}

// One long tile (a single point with more than 64 observations) of k_back_pm_pipe: the wavefront
// strides over the observations twice, as k_back_pm does.
template <class M>
__device__ __forceinline__ void back_tile_long(const Dev& d, const double* __restrict__ dc, double* __restrict__ dp_out,
                                               int beg, int end, int lane,
                                               double& model, double& step2, double& x2, double& trial, double& bad) {
    double e[3] = {0, 0, 0};
    const int key = __builtin_amdgcn_readfirstlane(d.pm_pt[beg]);
    const double4 X = d.X[key];
    for (int j = beg + lane; j < end; j += 64) {
        const BackObs<M> o = back_load<M>(d, dc, j, X);
        M::jpT_times_acc(o.s, o.k.q, o.w0, o.w1, e);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) e[i] = wave_sum(e[i]);
    const double4 g = d.s0[key];
    const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)key);
    const double2 v01 = vi[0], v23 = vi[1], v45 = vi[2];
    double dp[3];
    dp[0] = -(v01.x * e[0] + v01.y * e[1] + v23.x * e[2] + g.x);
    dp[1] = -(v01.y * e[0] + v23.y * e[1] + v45.x * e[2] + g.y);
    dp[2] = -(v23.x * e[0] + v45.x * e[1] + v45.y * e[2] + g.z);
    if (lane == 0) {
        d.X_trial[key] = make_double4(X.x + dp[0], X.y + dp[1], X.z + dp[2], 0.0);
        if (dp_out) { dp_out[3 * (size_t)key] = dp[0]; dp_out[3 * (size_t)key + 1] = dp[1]; dp_out[3 * (size_t)key + 2] = dp[2]; }
        step2 += dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
        x2 += X.x * X.x + X.y * X.y + X.z * X.z;
    }
    for (int j = beg + lane; j < end; j += 64) {
        const BackObs<M> b = back_load<M>(d, dc, j, X);
        const typename M::Cam kt = M::load(d.cams_trial, b.c);
        double u0, u1;
        M::jp_times(b.s, b.k.q, dp, u0, u1);
        const double l0 = b.ru + b.w0 + u0, l1 = b.rv + b.w1 + u1;
        model += l0 * l0 + l1 * l1;
        trial_cost<M>(d, b, kt, X, dp, trial, bad);
    }
}

// ---- k_back_pm as a pipeline over the tiles of a resident wavefront -----------------------------
// What an ordinary tile (one observation per lane) needs arrives in two steps, each requested a
// tile ahead of its use and by all lanes on clamped indices (see k_lin_pm): the streams of the
// observations, then what they point at - the point, its damped inverse block and s0, and the
// tile's cameras [c0, c0 + nst) with their trial versions and steps, which pass through registers
// into the wavefront's LDS stage.
struct BackStreams {
    int beg, end;
    bool is_long;
    int key, c;
    double2 uv;
};
template <class M>
struct BackGather {
    double4 X, g;
    double2 v01, v23, v45;
    double2 cam[2], trial[2];
    double dcr[3];
    int c0, nst;
};
static_assert(kStageCams * 9 <= 192 && kStageCams * 8 <= 128, "stage pieces per lane");
__device__ __forceinline__ BackStreams back_streams_fetch(const Dev& d, int q, int q_end, int lane, int ranges, int rbase) {
    BackStreams t;
    t.beg = t.end = 0;
    if (q < q_end) {
        t.beg = __builtin_amdgcn_readfirstlane(__shfl(ranges, q - rbase, 64));
        t.end = __builtin_amdgcn_readfirstlane(__shfl(ranges, q + 1 - rbase, 64));
    }
    t.is_long = t.end - t.beg > 64;
    const int j = t.beg + lane;
    const int jc = min(j, d.n_obs - 1);
    const int key = d.pm_pt[jc], c = d.pm_cam[jc];
    t.uv = d.pm_uv[jc];
    const bool live = !t.is_long && j < t.end;
    t.key = live ? key : -1 - lane;
    t.c = live ? c : 0x7fffffff;
    return t;
}
template <class M>
__device__ __forceinline__ BackGather<M> back_gather_fetch(const Dev& d, const double* __restrict__ dc, const BackStreams& t, int lane) {
    BackGather<M> g;
    const int c0 = wave_min_int(t.c);
    const bool any = c0 != 0x7fffffff;
    g.c0 = any ? c0 : 0;
    g.nst = any ? min(kStageCams, d.n_cam - c0) : 0;
    const int ncw = max(g.nst, 1) * M::CW, nd = max(g.nst, 1) * M::ND;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = min(lane + 64 * q, ncw - 1);
        g.cam[q] = d.cams[(size_t)g.c0 * M::CW + i];
        g.trial[q] = d.cams_trial[(size_t)g.c0 * M::CW + i];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) g.dcr[q] = dc[(size_t)g.c0 * M::ND + min(lane + 64 * q, nd - 1)];
    const int p = max(t.key, 0);
    g.X = d.X[p];
    g.g = d.s0[p];
    const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)p);
    g.v01 = vi[0]; g.v23 = vi[1]; g.v45 = vi[2];
    return g;
}
// tiles of one register of boundaries (63) per stretch; grid: what the chip holds at once
constexpr int kBackMaxBlocks = SFMBA_BACK_BLOCKS;

template <class M>
__global__ __launch_bounds__(kBlock) void k_back_pm_pipe(Dev d, const double* __restrict__ dc, double* __restrict__ dp_out) {
    __shared__ double red[block_reduce_scratch<5>()];
    __shared__ __attribute__((aligned(16))) CamStage<M> stage[kBlock / 64];
    const int lane = threadIdx.x & 63;
    double model = 0.0, step2 = 0.0, x2 = 0.0, trial = 0.0, bad = 0.0;
    const int w = xcd_block(blockIdx.x, gridDim.x) * (kBlock / 64) + (threadIdx.x >> 6);
    const int W = gridDim.x * (kBlock / 64);
    const int t0 = (int)((long long)w * d.n_wtiles / W), t1_all = (int)((long long)(w + 1) * d.n_wtiles / W);
    CamStage<M>& sg = stage[threadIdx.x >> 6];
#pragma unroll 1
    for (int rbase = t0; rbase < t1_all; rbase += 63) {
    const int t1 = min(rbase + 63, t1_all);
    const int ranges = d.wtile_beg[min(rbase + lane, d.n_wtiles)];
    BackStreams s1 = back_streams_fetch(d, rbase, t1, lane, ranges, rbase);
    BackStreams s2 = back_streams_fetch(d, rbase + 1, t1, lane, ranges, rbase);
    BackGather<M> g1 = back_gather_fetch<M>(d, dc, s1, lane);
#pragma unroll 1
    for (int q = rbase; q < t1; ++q) {
        const BackStreams cu = s1;
        const BackGather<M> gc = g1;
        s1 = s2;
        s2 = back_streams_fetch(d, q + 2, t1, lane, ranges, rbase);
        g1 = back_gather_fetch<M>(d, dc, s1, lane);
        if (cu.is_long) {
            back_tile_long<M>(d, dc, dp_out, cu.beg, cu.end, lane, model, step2, x2, trial, bad);
            continue;
        }
        // the tile's cameras into the stage
        const int c0 = gc.c0, nst = gc.nst;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (lane + 64 * k < nst * M::CW) { sg.cam[lane + 64 * k] = gc.cam[k]; sg.trial[lane + 64 * k] = gc.trial[k]; }
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (lane + 64 * k < nst * M::ND) sg.dc[lane + 64 * k] = gc.dcr[k];
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
        __builtin_amdgcn_wave_barrier();
        const int key = cu.key;
        const bool live = key >= 0;
        const bool outside = __any(live && cu.c - c0 >= nst);
        BackObs<M> o{};
        typename M::Cam kt{};
        const double4 X = gc.X;
        double e[3] = {0, 0, 0};
        if (live) {
            o.uv = cu.uv;
            o.c = cu.c;
            const int rel = min(cu.c - c0, nst - 1);
            double dd[M::ND];
            o.k = M::load(sg.cam, rel);
            kt = M::load(sg.trial, rel);
#pragma unroll
            for (int i = 0; i < M::ND; ++i) dd[i] = sg.dc[M::ND * rel + i];
            if (outside) {                      // (wave-uniform: a tile whose cameras all lie in the piece has no load here)
                if (cu.c - c0 >= nst) {
                    o.k = M::load(d.cams, cu.c);
                    kt = M::load(d.cams_trial, cu.c);
                    load_tangent<M::ND>(dc, cu.c, dd);
                }
            }
            back_finish<M>(d, o, dd, X);
            M::jpT_times_acc(o.s, o.k.q, o.w0, o.w1, e);
        }
        seg_scan(e, key, lane);
        const int last = seg_last(key, lane);
#pragma unroll
        for (int i = 0; i < 3; ++i) e[i] = __shfl(e[i], last, 64);
        const bool writer = live && last == lane;
        double dp[3] = {0, 0, 0};
        if (live) {
            // g holds V*^-1 g_p
            dp[0] = -(gc.v01.x * e[0] + gc.v01.y * e[1] + gc.v23.x * e[2] + gc.g.x);
            dp[1] = -(gc.v01.y * e[0] + gc.v23.y * e[1] + gc.v45.x * e[2] + gc.g.y);
            dp[2] = -(gc.v23.x * e[0] + gc.v45.x * e[1] + gc.v45.y * e[2] + gc.g.z);
        }
        if (writer) {
            d.X_trial[key] = make_double4(X.x + dp[0], X.y + dp[1], X.z + dp[2], 0.0);
            if (dp_out) { dp_out[3 * (size_t)key] = dp[0]; dp_out[3 * (size_t)key + 1] = dp[1]; dp_out[3 * (size_t)key + 2] = dp[2]; }
            step2 += dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
            x2 += X.x * X.x + X.y * X.y + X.z * X.z;
        }
        if (live) {
            double u0, u1;
            M::jp_times(o.s, o.k.q, dp, u0, u1);
            const double l0 = o.ru + o.w0 + u0, l1 = o.rv + o.w1 + u1;
            model += l0 * l0 + l1 * l1;
            trial_cost<M>(d, o, kt, X, dp, trial, bad);
        }
        __builtin_amdgcn_wave_barrier();          // the next tile's stores to the stage stay behind these reads
    }
    }
    const double vals[5] = {model, step2, x2, trial, bad};
    const double r = block_reduce_n<5>(vals, 0u, red);
    if (threadIdx.x < 5) put_partial(d, SLOT_MODEL + threadIdx.x, r);
}

template <class M, bool kStage>
__global__ __launch_bounds__(kBlock) void k_back_pm(Dev d, const double* __restrict__ dc,
                                                    double* __restrict__ dp_out) {
    __shared__ double red[block_reduce_scratch<5>()];
    __shared__ __attribute__((aligned(16))) CamStage<M> stage[kStage ? kBlock / 64 : 1];
    const int lane = threadIdx.x & 63;
    double model = 0.0, step2 = 0.0, x2 = 0.0, trial = 0.0, bad = 0.0;
    const Tile t = wave_tile(d);        // (several tiles per wavefront, as in k_lin_pm, measured: 3 % slower here)
    double e[3] = {0, 0, 0};
    int key = -1 - lane;
    BackObs<M> o{};
    double4 X = make_double4(0.0, 0.0, 0.0, 0.0);
    // everything the lane will need after the segmented sum is requested up front, so that the
    // scan does not sit between two rounds of memory latency
    double4 g = make_double4(0.0, 0.0, 0.0, 0.0);
    double2 v01 = make_double2(0.0, 0.0), v23 = v01, v45 = v01;
    typename M::Cam kt{};
    if (kStage && !t.is_long) {
        // ordinary tile: one observation per lane, cameras through the wavefront's LDS stage
        CamStage<M>& sg = stage[threadIdx.x >> 6];
        const int j = t.beg + lane;
        const bool live = j < t.end;
        int c = 0x7fffffff;
        if (live) {
            key = d.pm_pt[j];
            c = d.pm_cam[j];
            o.uv = d.pm_uv[j];
            X = d.X[key];
            g = d.s0[key];
            const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)key);
            v01 = vi[0]; v23 = vi[1]; v45 = vi[2];
        }
        const int c0 = wave_min_int(c);
        const int nst = t.end > t.beg ? min(kStageCams, d.n_cam - c0) : 0;       // cameras [c0, c0 + nst)
        for (int i = lane; i < nst * M::CW; i += 64) {
            sg.cam[i] = d.cams[(size_t)c0 * M::CW + i];
            sg.trial[i] = d.cams_trial[(size_t)c0 * M::CW + i];
        }
        for (int i = lane; i < nst * M::ND; i += 64) sg.dc[i] = dc[(size_t)c0 * M::ND + i];
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if (live) {
            o.c = c;
            const int rel = c - c0;
            double dd[M::ND];
            if (rel < nst) {
                o.k = M::load(sg.cam, rel);
                kt = M::load(sg.trial, rel);
#pragma unroll
                for (int i = 0; i < M::ND; ++i) dd[i] = sg.dc[M::ND * rel + i];
            } else {
                o.k = M::load(d.cams, c);
                kt = M::load(d.cams_trial, c);
                load_tangent<M::ND>(dc, c, dd);
            }
            back_finish<M>(d, o, dd, X);
            M::jpT_times_acc(o.s, o.k.q, o.w0, o.w1, e);
        }
    } else
    for (int j = t.beg + lane; j < t.end; j += 64) {
        key = d.pm_pt[j];
        X = d.X[key];
        if (!t.is_long) {
            g = d.s0[key];
            const double2* vi = reinterpret_cast<const double2*>(d.Vinv + 6 * (size_t)key);
            v01 = vi[0]; v23 = vi[1]; v45 = vi[2];
        }
        o = back_load<M>(d, dc, j, X);
        if (!t.is_long) kt = M::load(d.cams_trial, o.c);
        M::jpT_times_acc(o.s, o.k.q, o.w0, o.w1, e);
    }
    bool writer;
    if (t.is_long) {
#pragma unroll
        for (int i = 0; i < 3; ++i) e[i] = wave_sum(e[i]);
        key = __shfl(key, 0, 64);
        writer = lane == 0;
        if (key >= 0) {
            X = d.X[key];
            g = d.s0[key];
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
        // g holds V*^-1 g_p
        dp[0] = -(v01.x * e[0] + v01.y * e[1] + v23.x * e[2] + g.x);
        dp[1] = -(v01.y * e[0] + v23.y * e[1] + v45.x * e[2] + g.y);
        dp[2] = -(v23.x * e[0] + v45.x * e[1] + v45.y * e[2] + g.z);
    }
    if (writer) {
        d.X_trial[key] = make_double4(X.x + dp[0], X.y + dp[1], X.z + dp[2], 0.0);
        if (dp_out) { dp_out[3 * (size_t)key] = dp[0]; dp_out[3 * (size_t)key + 1] = dp[1]; dp_out[3 * (size_t)key + 2] = dp[2]; }
        step2 += dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
        x2 += X.x * X.x + X.y * X.y + X.z * X.z;
    }
    if (t.is_long) {
        for (int j = t.beg + lane; j < t.end; j += 64) {
            const BackObs<M> b = back_load<M>(d, dc, j, X);
            const typename M::Cam kt = M::load(d.cams_trial, b.c);
            double u0, u1;
            M::jp_times(b.s, b.k.q, dp, u0, u1);
            const double l0 = b.ru + b.w0 + u0, l1 = b.rv + b.w1 + u1;
            model += l0 * l0 + l1 * l1;
            trial_cost<M>(d, b, kt, X, dp, trial, bad);
        }
    } else if (key >= 0) {
        double u0, u1;
        M::jp_times(o.s, o.k.q, dp, u0, u1);
        const double l0 = o.ru + o.w0 + u0, l1 = o.rv + o.w1 + u1;
        model += l0 * l0 + l1 * l1;
        trial_cost<M>(d, o, kt, X, dp, trial, bad);
    }
    static_assert(SLOT_STEP2_PT == SLOT_MODEL + 1 && SLOT_X2_PT == SLOT_MODEL + 2 && SLOT_TRIAL == SLOT_MODEL + 3 &&
                  SLOT_NONFINITE == SLOT_MODEL + 4, "slot order");
    const double vals[5] = {model, step2, x2, trial, bad};
    const double r = block_reduce_n<5>(vals, 0u, red);
    if (threadIdx.x < 5) put_partial(d, SLOT_MODEL + threadIdx.x, r);
}

// Camera retraction: Q' = exp([dw]x) Q, t' = exp([dw]x) t + dt.
template <class M>
__global__ __launch_bounds__(kBlock) void k_cam_update(Dev d, const double* __restrict__ dc) {
    constexpr int ND = M::ND;
    __shared__ double red[16];
    int c = blockIdx.x * kBlock + threadIdx.x;
    double step2 = 0.0, x2 = 0.0;
    if (c < d.n_cam) {
        const typename M::Cam k = M::load(d.cams, c);
        double w[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) { w[i] = dc[ND * (size_t)c + i]; step2 += w[i] * w[i]; }
        M::store(d.cams_trial, c, M::retract(k, w, x2));
    }
    double a = block_sum(step2, red);
    double b = block_sum(x2, red);
    if (threadIdx.x == 0) { put_partial(d, SLOT_STEP2_CAM, a); put_partial(d, SLOT_X2_CAM, b); }
}

// scalars[slot] = sum (or max) of the block partials of that slot; one workgroup per job, so that
// everything the host wants to read after a wait can be finished by ONE launch: the sums of the
// trial step, the two slots k_cam_update left as block partials and, when the linearisation before
// it left them unfinished (its values are only read by the host, together with these), its cost
// and gradient maxima.
struct FinalizeJobs {
    int n;
    int slot[12];
    int nblocks[12];    // partials of the slot (kind 0, 1)
    int kind[12];       // 0 sum, 1 max of block partials, 2 max |g_c| over the cameras out of UG
};
template <class M>
__global__ __launch_bounds__(1024) void k_scal_finalize(Dev d, FinalizeJobs jobs) {
    __shared__ double red[16];
    const int slot = jobs.slot[blockIdx.x], kind = jobs.kind[blockIdx.x];
    const bool is_max = kind != 0;
    const int nblocks = kind == 2 ? d.n_cam * M::ND : jobs.nblocks[blockIdx.x];
    const double* part = d.scal_part + (size_t)slot * kMaxScalBlocks;
    // eight loads in flight per thread (the order of the additions is fixed all the same)
    double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i0 = threadIdx.x; i0 < nblocks; i0 += 8 * 1024) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 1024;
            if (kind == 2) x[u] = i < nblocks ? fabs(d.UG[M::NV * (size_t)(i / M::ND) + M::NU + (i % M::ND)]) : 0.0;
            else x[u] = i < nblocks ? part[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = is_max ? fmax(a[u], x[u]) : a[u] + x[u];
    }
    double v = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) v = is_max ? fmax(v, a[u]) : v + a[u];
    double r = is_max ? block_max(v, red) : block_sum(v, red);
    if (threadIdx.x == 0) {
        d.scalars[slot] = r;
        if (d.scalars_host) d.scalars_host[slot] = r;
    }
}
static void add_job(FinalizeJobs& j, int slot, int nblocks, int kind) {
    j.slot[j.n] = slot; j.nblocks[j.n] = nblocks; j.kind[j.n] = kind; ++j.n;
}
static void add_lin_jobs(const Dev& d, FinalizeJobs& j);
static void run_jobs(const Dev& d, const FinalizeJobs& j, hipStream_t st) {
    if (j.n == 0) return;
    SFMBA_LAUNCH_T(d, k_scal_finalize, dim3(j.n), dim3(1024), st, d, j);
}

// Per-camera sum of reprojection errors |projection - observation| at the current estimate,
// camera-major: chunk partials (sum, count).  The per-view quantity the reference prints after
// every bundle adjustment (/root/reference/view.py:103-119, utils.py:337-352).
template <class M>
__global__ __launch_bounds__(kCmBlock) void k_reproj_cm(Dev d) {
    const int lane = threadIdx.x & 63;
    const int q = xcd_block(blockIdx.x, gridDim.x) * (kCmBlock / 64) + (threadIdx.x >> 6);
    if (q >= d.n_chunks) return;
    const int c = __builtin_amdgcn_readfirstlane(d.chunk_cam[q]);
    const int k0 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q]);
    const int k1 = __builtin_amdgcn_readfirstlane(d.chunk_beg[q + 1]);
    const typename M::Cam cam = M::load(d.cams, c);
    double sum = 0.0;
    for (int k = k0 + lane; k < k1; k += 64) {
        const double4 X = d.X[d.cm_pt[k]];
        const typename M::Obs ob = M::obs(M::seed(cam, X.x, X.y, X.z), cam, d.K);
        const double2 uv = d.cm_uv[k];
        double ru, rv;
        M::residual(ob, uv.x, uv.y, ru, rv);
        sum += sqrt(ru * ru + rv * rv);
    }
    sum = wave_sum(sum);
    if (lane == 0) {
        d.chunk_part[2 * (size_t)q] = sum;
        d.chunk_part[2 * (size_t)q + 1] = (double)(k1 - k0);
    }
}
void launch_reproj_cm(const Dev& d, double* out2, hipStream_t st) {
    if (d.n_chunks > 0) SFMBA_LAUNCH_T(d, k_reproj_cm, dim3(blocks_chunk(d)), dim3(kCmBlock), st, d);
    launch_chunk_reduce(d, 2, out2, st);
}

// max |g_c| over the cameras (after the all-reduce of UG), straight into its scalar slot.
void launch_gmax_cam(const Dev& d, hipStream_t st) {
    FinalizeJobs j{};
    add_job(j, SLOT_GMAX_CAM, 0, 2);
    run_jobs(d, j, st);
}

// a += b (in-process all-reduce of several handles)
__global__ __launch_bounds__(kBlock) void k_vec_add(double* __restrict__ a, const double* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) a[i] += b[i];
}
void launch_vec_add(double* a, const double* b, size_t n, hipStream_t st) {
    hipLaunchKernelGGL(k_vec_add, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, a, b, n);
}

// Points between the caller's numbering (n x 3 doubles) and the internal order (double4 records).
__global__ __launch_bounds__(kBlock) void k_points_in(const double* __restrict__ in, const int* __restrict__ new2old, int n,
                                                      double4* __restrict__ X) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const double* x = in + 3 * (size_t)new2old[p];
    X[p] = make_double4(x[0], x[1], x[2], 0.0);
}
__global__ __launch_bounds__(kBlock) void k_points_out(const double4* __restrict__ X, const int* __restrict__ new2old, int n,
                                                       double* __restrict__ out) {
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const double4 x = X[p];
    double* o = out + 3 * (size_t)new2old[p];
    o[0] = x.x; o[1] = x.y; o[2] = x.z;
}
void launch_points_in(const double* in, const int* new2old, int n, double4* X, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_points_in, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, st, in, new2old, n, X);
}
void launch_points_out(const double4* X, const int* new2old, int n, double* out, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_points_out, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, st, X, new2old, n, out);
}

// ------------------------------------------------------------------------------------------
int scal_blocks_pt(const Dev& d) { return cdiv(d.n_pts, kBlock); }
static int blocks_wtile(const Dev& d) { return cdiv(d.n_wtiles, kBlock / 64); }
int blocks_cam(const Dev& d) { return cdiv(d.n_cam, kBlock); }


void launch_residuals(const Dev& d, double* out, hipStream_t st) {
    if (d.n_obs == 0) return;
    SFMBA_LAUNCH(d, k_residuals, dim3(cdiv(d.n_obs, kBlock)), st, d, out);
}
void launch_jacobian_blocks(const Dev& d, double* jc, double* jp, hipStream_t st) {
    if (d.n_obs == 0) return;
    SFMBA_LAUNCH(d, k_jacobian_blocks, dim3(cdiv(d.n_obs, kBlock)), st, d, jc, jp);
}
// Grid of k_lin_pm: one wavefront per job while the chip holds them all at once (two workgroups per
// compute unit, what the LDS stage admits), beyond that the resident wavefronts share the jobs.
constexpr int kLinPmMaxBlocks = 512;
int lin_pm_blocks(const Dev& d) {
    // (SFMBA_LIN_PM_BLOCKS: another cap, for tests of wavefronts with many jobs and for tuning)
    static const int cap = getenv("SFMBA_LIN_PM_BLOCKS") ? std::max(1, atoi(getenv("SFMBA_LIN_PM_BLOCKS"))) : kLinPmMaxBlocks;
    return std::min(cdiv(cdiv(d.n_wtiles, kTilesPerWave), kBlock / 64), cap);
}
void launch_lin_pm(const Dev& d, bool store_seeds, double lambda, hipStream_t st, bool grad_only, bool keep_v, bool defer_scalars) {
    const int kv = keep_v || !(lambda > 0.0) ? 1 : 0;     // undamped: V and g_p are the only products
    // (the gradient-only variant has no LDS stage and fills the chip with a wavefront per job; it is
    // never deferred, so the count of its block partials is only needed here)
    const int nb = grad_only ? cdiv(cdiv(d.n_wtiles, kTilesPerWave), kBlock / 64) : lin_pm_blocks(d);
    if (nb > 0 && grad_only) {
        if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_lin_pm<Bal, false, true>), dim3(nb), dim3(kBlock), 0, st, d, 0.0, 0);
        else hipLaunchKernelGGL((k_lin_pm<Pinhole, false, true>), dim3(nb), dim3(kBlock), 0, st, d, 0.0, 0);
    } else if (nb > 0) {
        if (d.model == MODEL_BAL && store_seeds) hipLaunchKernelGGL((k_lin_pm<Bal, true>), dim3(nb), dim3(kBlock), 0, st, d, lambda, kv);
        else if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_lin_pm<Bal, false>), dim3(nb), dim3(kBlock), 0, st, d, lambda, kv);
        else if (store_seeds) hipLaunchKernelGGL((k_lin_pm<Pinhole, true>), dim3(nb), dim3(kBlock), 0, st, d, lambda, kv);
        else hipLaunchKernelGGL((k_lin_pm<Pinhole, false>), dim3(nb), dim3(kBlock), 0, st, d, lambda, kv);
    }
    if (!defer_scalars) {
        FinalizeJobs j{};
        add_job(j, SLOT_COST, nb, 0);
        add_job(j, SLOT_GMAX_PT, nb, 1);
        run_jobs(d, j, st);
    }
}
void launch_lin_cm(const Dev& d, hipStream_t st, bool grad_only) {
    if (d.n_chunks == 0) return;
    const dim3 grid(blocks_chunk(d));
    if (grad_only) {
        if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_lin_cm<Bal, true>), grid, dim3(kCmBlock), 0, st, d);
        else hipLaunchKernelGGL((k_lin_cm<Pinhole, true>), grid, dim3(kCmBlock), 0, st, d);
    } else {
        if (d.model == MODEL_BAL) hipLaunchKernelGGL((k_lin_cm<Bal, false>), grid, dim3(kCmBlock), 0, st, d);
        else hipLaunchKernelGGL((k_lin_cm<Pinhole, false>), grid, dim3(kCmBlock), 0, st, d);
    }
}
void launch_chunk_reduce(const Dev& d, int nvals, double* out, hipStream_t st) {
    hipLaunchKernelGGL(k_chunk_reduce, dim3(cdiv(d.n_cam * nvals, kBlock)), dim3(kBlock), 0, st, d, nvals, out);
}
int lin_sdiag_values(int model) { return model == MODEL_BAL ? Bal::NU + 3 * Bal::ND : Pinhole::NU + 3 * Pinhole::ND; }
void launch_lin_sdiag_cm(const Dev& d, double* tpart, hipStream_t st) {
    if (d.n_chunks > 0) SFMBA_LAUNCH_T(d, k_lin_sdiag_cm, dim3(blocks_chunk(d)), dim3(kCmBlock), st, d);
    SFMBA_LAUNCH(d, k_chunk_reduce_fused, dim3(cdiv(d.n_cam * 2 * cam_sums(d.model), kBlock)), st, d, tpart);
}
void launch_damp_pm(const Dev& d, double lambda, hipStream_t st) {
    if (d.n_pts == 0) return;
    hipLaunchKernelGGL(k_damp_pm, dim3(scal_blocks_pt(d)), dim3(kBlock), 0, st, d, lambda);
}
void launch_sdiag_cm(const Dev& d, hipStream_t st) {
    if (d.n_chunks == 0) return;
    SFMBA_LAUNCH_T(d, k_sdiag_cm, dim3(blocks_chunk(d)), dim3(kCmBlock), st, d);
}
void launch_precond(const Dev& d, double lambda, hipStream_t st) {
    SFMBA_LAUNCH(d, k_precond, dim3(blocks_cam(d)), st, d, lambda);
}
void launch_spmv_pm(const Dev& d, const double* x, const PcgState* state, hipStream_t st) {
    if (d.n_wtiles == 0) return;
    SFMBA_LAUNCH(d, k_spmv_pm, dim3(blocks_wtile(d)), st, d, x, state);
}
void launch_spmv_cm(const Dev& d, const double4* svec, const PcgState* state, hipStream_t st) {
    if (d.n_chunks == 0) return;
    SFMBA_LAUNCH_T(d, k_spmv_cm, dim3(blocks_chunk(d)), dim3(kCmBlock), st, d, svec, state);
}
void launch_schur_apply(const Dev& d, const double* x, double lambda, double* y, hipStream_t st) {
    SFMBA_LAUNCH(d, k_schur_apply, dim3(blocks_cam(d)), st, d, x, lambda, y);
}
void launch_back_pm(const Dev& d, const double* dc, double* dp_out, hipStream_t st, bool with_lin_scalars) {
    // SFMBA_BACK_PM: 2 (default) the pipeline over resident wavefronts, 1 a wavefront per tile with
    // the cameras staged in LDS, 0 a wavefront per tile with every camera gathered
    static const int mode = getenv("SFMBA_BACK_PM") ? atoi(getenv("SFMBA_BACK_PM")) : 2;
    static const int cap = getenv("SFMBA_BACK_PM_BLOCKS") ? std::max(1, atoi(getenv("SFMBA_BACK_PM_BLOCKS"))) : kBackMaxBlocks;
    const int nb = mode == 2 ? std::min(blocks_wtile(d), cap) : blocks_wtile(d);
    if (nb <= 0) return;
    if (mode == 2) {
        SFMBA_LAUNCH(d, k_back_pm_pipe, dim3(nb), st, d, dc, dp_out);
    } else if (d.model == MODEL_BAL) {
        if (mode == 1) hipLaunchKernelGGL((k_back_pm<Bal, true>), dim3(nb), dim3(kBlock), 0, st, d, dc, dp_out);
        else hipLaunchKernelGGL((k_back_pm<Bal, false>), dim3(nb), dim3(kBlock), 0, st, d, dc, dp_out);
    } else {
        if (mode == 1) hipLaunchKernelGGL((k_back_pm<Pinhole, true>), dim3(nb), dim3(kBlock), 0, st, d, dc, dp_out);
        else hipLaunchKernelGGL((k_back_pm<Pinhole, false>), dim3(nb), dim3(kBlock), 0, st, d, dc, dp_out);
    }
    // MODEL .. NONFINITE, and with them the two slots k_cam_update left as block partials
    FinalizeJobs j{};
    for (int i = 0; i < 5; ++i) add_job(j, SLOT_MODEL + i, nb, 0);
    add_job(j, SLOT_STEP2_CAM, blocks_cam(d), 0);
    add_job(j, SLOT_X2_CAM, blocks_cam(d), 0);
    if (with_lin_scalars) add_lin_jobs(d, j);
    run_jobs(d, j, st);
}
// The scalars a linearisation whose launch_lin_pm deferred them still owes: cost, max |g_p|, max |g_c|.
static void add_lin_jobs(const Dev& d, FinalizeJobs& j) {
    add_job(j, SLOT_COST, lin_pm_blocks(d), 0);
    add_job(j, SLOT_GMAX_PT, lin_pm_blocks(d), 1);
    add_job(j, SLOT_GMAX_CAM, 0, 2);
}
void launch_lin_scalars(const Dev& d, hipStream_t st) {
    FinalizeJobs j{};
    add_lin_jobs(d, j);
    run_jobs(d, j, st);
}
void launch_cam_update(const Dev& d, const double* dc, hipStream_t st) {
    // (its two scalar slots are finalised together with those of k_back_pm, which always follows)
    SFMBA_LAUNCH(d, k_cam_update, dim3(blocks_cam(d)), st, d, dc);
}

}  // namespace sfmba
