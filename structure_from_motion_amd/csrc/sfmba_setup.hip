// Problem set-up on the GPU: everything sfmba_set_problem derives from the caller's index arrays
// (camera_indices, point_indices of /root/reference/bundle_adjustment.py:107-122) - the index
// check NumPy does at bundle_adjustment.py:49, the point partition of this rank, the renumbering
// of the points, the point-major and camera-major orderings, camera chunks, wave tiles, and for
// the explicit reduced matrix the observation-pair lists per upper block and the block-CSR
// pattern - as radix sorts, scans and small kernels on the handle's stream.  The reference calls
// bundle adjustment on a new problem after every image (main.py:115-116), so this is paid per
// optimize() call.
//
// The arrays produced are the ones host_setup (sfmba_host.cpp) produces, element for element:
// every sort is a stable LSD radix sort keyed exactly like the host's counting sorts, and the
// sequential rules (wave tiles, chunk cuts) are evaluated in closed form.  Solves are therefore
// bit-identical between the two paths (tests/test_gpu_setup.py).
#include <string.h>      // (rocPRIM's texture iterator calls the host memset without including it)

#include <rocprim/rocprim.hpp>

#include <algorithm>

#include "sfmba_handle.h"

namespace sfmba {

namespace {

constexpr int kT = 256;
inline unsigned grid_for(size_t n) { return (unsigned)std::max<size_t>((n + kT - 1) / kT, 1); }   // (every kernel checks its bound)
inline unsigned bits_for(uint64_t max_value) {      // bits needed to hold values 0..max_value
    unsigned b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

// Bump allocator over one grow-only device buffer of the handle.
struct Arena {
    DBuf<unsigned char>& buf;
    size_t used = 0;
    explicit Arena(DBuf<unsigned char>& b) : buf(b) {}
    static size_t pad(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
    void reserve(size_t bytes) { buf.alloc(bytes); used = 0; }
    template <class T>
    T* get(size_t count) {
        T* p = reinterpret_cast<T*>(buf.p + used);
        used += pad(count * sizeof(T));
        if (used > buf.n) fail(SFMBA_ERR_INVALID_ARGUMENT, "set-up workspace overrun (%zu of %zu bytes)", used, buf.n);
        return p;
    }
};

// rocPRIM calls: query the temporary storage, grow the handle's buffer for it, run.
#define PRIM_TRY(expr)                                             \
    do {                                                           \
        hipError_t e_ = (expr);                                    \
        if (e_ != hipSuccess) throw HipError{e_, #expr};           \
    } while (0)

template <class K, class VIn>
void sort_pairs(H* h, const K* keys_in, K* keys_out, VIn vals_in, int* vals_out, size_t n, unsigned bits) {
    if (n == 0) return;
    size_t bytes = 0;
    PRIM_TRY(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, h->stream));
    h->ws[3].alloc(std::max<size_t>(bytes, 256));
    PRIM_TRY(rocprim::radix_sort_pairs(h->ws[3].p, bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, bits, h->stream));
}
template <class T>
void exclusive_scan(H* h, const T* in, T* out, size_t n) {
    if (n == 0) return;
    size_t bytes = 0;
    PRIM_TRY(rocprim::exclusive_scan(nullptr, bytes, in, out, T(0), n, rocprim::plus<T>(), h->stream));
    h->ws[3].alloc(std::max<size_t>(bytes, 256));
    PRIM_TRY(rocprim::exclusive_scan(h->ws[3].p, bytes, in, out, T(0), n, rocprim::plus<T>(), h->stream));
}
// out = the in[i] with flag[i] != 0, in order; *count_dev receives how many.
void select_flagged(H* h, const int* in, const unsigned char* flags, int* out, int* count_dev, size_t n) {
    if (n == 0) { HIP_TRY(hipMemsetAsync(count_dev, 0, sizeof(int), h->stream)); return; }
    size_t bytes = 0;
    PRIM_TRY(rocprim::select(nullptr, bytes, in, flags, out, count_dev, n, h->stream));
    h->ws[3].alloc(std::max<size_t>(bytes, 256));
    PRIM_TRY(rocprim::select(h->ws[3].p, bytes, in, flags, out, count_dev, n, h->stream));
}

// ---- kernels ------------------------------------------------------------------------------

// Index check and narrowing: the first observation (smallest i) with an index out of range.
__global__ __launch_bounds__(kT) void k_check_narrow(const int64_t* __restrict__ cam, const int64_t* __restrict__ pt, int n,
                                                     int64_t n_cam, int64_t n_pts, unsigned* __restrict__ cam32,
                                                     unsigned* __restrict__ pt32, int* __restrict__ first_bad) {
    const int i = blockIdx.x * kT + threadIdx.x;
    if (i >= n) return;
    const int64_t c = cam[i], p = pt[i];
    const bool bad = c < 0 || c >= n_cam || p < 0 || p >= n_pts;
    cam32[i] = bad ? 0u : (unsigned)c;
    pt32[i] = bad ? 0u : (unsigned)p;
    if (bad) atomicMin(first_bad, i);
}

// Offsets of every key's run in a sorted key array: ptr[k] = first position with key >= k, for
// k = 0..n_keys (ptr[n_keys] = n).  Thread s fills the keys between its predecessor's and its own.
__global__ __launch_bounds__(kT) void k_run_offsets(const unsigned* __restrict__ keys, int n, int n_keys, int* __restrict__ ptr) {
    const int s = blockIdx.x * kT + threadIdx.x;
    if (s > n) return;
    const int lo = s == 0 ? 0 : (int)keys[s - 1] + 1;
    const int hi = s == n ? n_keys : (int)keys[s];
    for (int k = lo; k <= hi; ++k) ptr[k] = s;
}

// Partition boundaries of ranks rk and rk + 1: the first point whose prefix count reaches
// (int64)((double)n_obs * r / nranks), as std::lower_bound over the prefix sums.
__global__ void k_partition(const int* __restrict__ pre, int n_pts, int n_obs, int rk, int nr, int* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int b[2];
    for (int j = 0; j < 2; ++j) {
        const int r = rk + j;
        if (r <= 0) { b[j] = 0; continue; }
        if (r >= nr) { b[j] = n_pts; continue; }
        const long long target = (long long)((double)n_obs * r / nr);
        int lo = 0, hi = n_pts + 1;                         // first index with pre[idx] >= target
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long long)pre[mid] < target) lo = mid + 1; else hi = mid; }
        b[j] = lo;
    }
    const int lo = min(b[0], n_pts), hi = min(max(b[1], lo), n_pts);
    out[0] = lo; out[1] = hi; out[2] = pre[lo]; out[3] = pre[hi];
}

__global__ __launch_bounds__(kT) void k_gather_u32(const unsigned* __restrict__ src, const int* __restrict__ idx, int n, unsigned* __restrict__ dst) {
    const int s = blockIdx.x * kT + threadIdx.x;
    if (s < n) dst[s] = src[idx[s]];
}

// Smallest observing camera of every owned point (n_cam when nobody observes it) and its
// observation count.
__global__ __launch_bounds__(kT) void k_point_keys(const int* __restrict__ pre, const unsigned* __restrict__ scam, int pt_lo, int np,
                                                   unsigned n_cam, unsigned* __restrict__ min_cam) {
    const int p = blockIdx.x * kT + threadIdx.x;
    if (p >= np) return;
    unsigned m = n_cam;
    for (int s = pre[pt_lo + p]; s < pre[pt_lo + p + 1]; ++s) m = min(m, scam[s]);
    min_cam[p] = m;
}

__global__ __launch_bounds__(kT) void k_renumber(const int* __restrict__ new2old, const int* __restrict__ pre, int pt_lo, int np,
                                                 int* __restrict__ old2new, int* __restrict__ cnt_new) {
    const int q = blockIdx.x * kT + threadIdx.x;
    if (q > np) return;
    if (q == np) { cnt_new[q] = 0; return; }
    const int p = new2old[q];
    old2new[p] = q;
    cnt_new[q] = pre[pt_lo + p + 1] - pre[pt_lo + p];
}

// Point-major arrays: sorted position s (by the caller's point id, stable) -> its place in the
// renumbered order, observations of a point keeping their order.
__global__ __launch_bounds__(kT) void k_point_major(const unsigned* __restrict__ spt, const int* __restrict__ sidx,
                                                    const unsigned* __restrict__ scam,
                                                    const int* __restrict__ pre, const int* __restrict__ old2new,
                                                    const int* __restrict__ pt_ptr, int pt_lo, int o_lo, int no,
                                                    int* __restrict__ pm_cam, int* __restrict__ pm_pt, int* __restrict__ pm_orig) {
    const int t = blockIdx.x * kT + threadIdx.x;
    if (t >= no) return;
    const int s = o_lo + t;
    const int p = (int)spt[s];
    const int q = old2new[p - pt_lo];
    const int j = pt_ptr[q] + (s - pre[p]);
    const int i = sidx[s];
    pm_cam[j] = (int)scam[s];
    pm_pt[j] = q;
    pm_orig[j] = i;
}

__global__ __launch_bounds__(kT) void k_camera_major(const int* __restrict__ cm2pm, const int* __restrict__ pm_pt, int no,
                                                     int* __restrict__ pm2cm, int* __restrict__ cm_pt) {
    const int k = blockIdx.x * kT + threadIdx.x;
    if (k >= no) return;
    const int j = cm2pm[k];
    pm2cm[j] = k;
    cm_pt[k] = pm_pt[j];
}
// The image points in both orders.  Apart from the index arrays this is all the structure needs
// of the caller's data, and it is applied last, so that the upload of points_2d (half of the
// bytes sfmba_set_problem moves over PCIe) can run while the index work above is on the GPU.
__global__ __launch_bounds__(kT) void k_place_uv(const double2* __restrict__ uv, const int* __restrict__ pm_orig, const int* __restrict__ pm2cm,
                                                 int no, double2* __restrict__ pm_uv, double2* __restrict__ cm_uv) {
    const int j = blockIdx.x * kT + threadIdx.x;
    if (j >= no) return;
    const double2 v = uv[pm_orig[j]];
    pm_uv[j] = v;
    cm_uv[pm2cm[j]] = v;
}

// Pieces of at most `chunk` items per run: counts, then (after a scan) the pieces themselves.
__global__ __launch_bounds__(kT) void k_piece_counts(const int* __restrict__ ptr, int n_runs, int piece, int* __restrict__ cnt) {
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c > n_runs) return;
    cnt[c] = c == n_runs ? 0 : (ptr[c + 1] - ptr[c] + piece - 1) / piece;
}
__global__ __launch_bounds__(kT) void k_piece_fill(const int* __restrict__ ptr, const int* __restrict__ piece_ptr, int n_runs, int piece,
                                                   int total, int* __restrict__ piece_run, int* __restrict__ piece_beg) {
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c > n_runs) return;
    if (c == n_runs) { piece_beg[piece_ptr[n_runs]] = total; return; }
    int q = piece_ptr[c];
    for (int b = ptr[c]; b < ptr[c + 1]; b += piece, ++q) { piece_run[q] = c; piece_beg[q] = b; }
}

// Wave tiles.  A tile is a maximal run of whole points with at most 64 observations; a point with
// more than 64 is a tile of its own.  For the point q that starts a tile, nxt[q] is the point
// that starts the next one (np = end).  Points without observations never start a tile.
__device__ __forceinline__ int last_index_le(const int* __restrict__ a, int n, int v) {   // largest idx in [0, n) with a[idx] <= v
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= v) lo = mid + 1; else hi = mid; }
    return lo - 1;
}
__global__ __launch_bounds__(kT) void k_tile_next(const int* __restrict__ pt_ptr, int np, int* __restrict__ nxt, unsigned char* __restrict__ mark) {
    const int q = blockIdx.x * kT + threadIdx.x;
    if (q > np) return;
    mark[q] = 0;
    if (q == np) { nxt[q] = np; return; }
    const int b = pt_ptr[q];
    int e = last_index_le(pt_ptr, np + 1, b + 64);
    if (pt_ptr[e] == b) e = last_index_le(pt_ptr, np + 1, pt_ptr[q + 1]);     // the point alone (more than 64, or none at all)
    nxt[q] = e;
}
__global__ void k_tile_first(const int* __restrict__ pt_ptr, int np, int no, unsigned char* __restrict__ mark) {
    if (threadIdx.x != 0 || blockIdx.x != 0 || no == 0) return;
    mark[last_index_le(pt_ptr, np + 1, 0)] = 1;           // the first point that has observations
}
// One round of pointer jumping: whatever is marked marks its 2^k-th successor; the jump doubles.
__global__ __launch_bounds__(kT) void k_tile_jump(const int* __restrict__ jin, int* __restrict__ jout, int np, unsigned char* __restrict__ mark) {
    const int q = blockIdx.x * kT + threadIdx.x;
    if (q > np) return;
    const int j = jin[q];
    if (mark[q] && j < np) mark[j] = 1;
    jout[q] = jin[j];
}

// Observation pairs of one point, in the order host_setup enumerates them (a < b over the
// point's observations): a pair of two cameras goes to the upper block (min, max) with the
// observation of the smaller camera first; two observations of the same camera give two entries
// (a, b) and (b, a) of that camera's diagonal block.
// One thread per observation a: its pairs (a, b) with the later observations b of the same point.
// `pt` gives the point of every observation, `ptr` the observation range of every point; points
// in [skip_lo, skip_hi) are left out.
__global__ __launch_bounds__(kT) void k_pair_counts(const int* __restrict__ ptr, const int* __restrict__ pt, const int* __restrict__ cam, int n,
                                                    int skip_lo, int skip_hi, long long* __restrict__ cnt) {
    const int a = blockIdx.x * kT + threadIdx.x;
    if (a > n) return;
    long long c = 0;
    if (a < n) {
        const int p = pt[a];
        if (!(p >= skip_lo && p < skip_hi)) {
            const int b1 = ptr[p + 1], ca = cam[a];
            c = b1 - 1 - a;
            for (int b = a + 1; b < b1; ++b) c += cam[b] == ca;
        }
    }
    cnt[a] = c;
}
template <bool kOwned>
__global__ __launch_bounds__(kT) void k_pair_fill(const int* __restrict__ ptr, const int* __restrict__ pt, const int* __restrict__ cam, int n,
                                                  int skip_lo, int skip_hi, const long long* __restrict__ off, unsigned long long n_cam,
                                                  unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int a = blockIdx.x * kT + threadIdx.x;
    if (a >= n) return;
    const int p = pt[a];
    if (p >= skip_lo && p < skip_hi) return;
    const int b1 = ptr[p + 1], ca = cam[a];
    long long k = off[a];
    for (int b = a + 1; b < b1; ++b) {
        const int cb = cam[b];
        const int lo = min(ca, cb), hi = max(ca, cb);
        const int reps = ca == cb ? 2 : 1;
        for (int r = 0; r < reps; ++r, ++k) {
            keys[k] = (unsigned long long)lo * n_cam + (unsigned long long)hi;
            vals[k] = kOwned ? p : -1;
        }
    }
}
__global__ __launch_bounds__(kT) void k_diag_keys(int n_cam, unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int c = blockIdx.x * kT + threadIdx.x;
    if (c >= n_cam) return;
    keys[c] = (unsigned long long)c * (unsigned)n_cam + (unsigned)c;
    vals[c] = -1;
}

// After the sort by block key: where a new block starts and which entries are pairs of this rank.
__global__ __launch_bounds__(kT) void k_block_flags(const unsigned long long* __restrict__ keys, const int* __restrict__ vals, size_t m,
                                                    int* __restrict__ head, int* __restrict__ keep) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i > m) return;
    if (i == m) { head[i] = 0; keep[i] = 0; return; }
    head[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
    keep[i] = vals[i] >= 0 ? 1 : 0;
}
// Block table and pair lists from the two scans (ublk = blocks before i, kpos = kept before i).
__global__ __launch_bounds__(kT) void k_block_table(const unsigned long long* __restrict__ keys, const int* __restrict__ vals, size_t m,
                                                    const int* __restrict__ head, const int* __restrict__ ublk, const int* __restrict__ kpos,
                                                    unsigned long long n_cam, int* __restrict__ ur, int* __restrict__ uc,
                                                    int* __restrict__ bcount, int* __restrict__ udiag, int* __restrict__ pp) {
    const size_t i = (size_t)blockIdx.x * kT + threadIdx.x;
    if (i > m) return;
    if (i == m) { bcount[ublk[m]] = kpos[m]; return; }
    if (head[i]) {
        const int u = ublk[i];
        const int r = (int)(keys[i] / n_cam), c = (int)(keys[i] % n_cam);
        ur[u] = r; uc[u] = c;
        bcount[u] = kpos[i];
        if (r == c) udiag[r] = u;
    }
    const int v = vals[i];
    if (v >= 0) pp[kpos[i]] = v;
}

// Full block pattern: entry (r, c) of every upper block and, off the diagonal, (c, r); the
// transposed entry of a diagonal block gets a key beyond every real one and is dropped.
__global__ __launch_bounds__(kT) void k_pattern_keys(const int* __restrict__ ur, const int* __restrict__ uc, int n_ublk, unsigned long long n_cam,
                                                     unsigned long long* __restrict__ keys, int* __restrict__ vals) {
    const int u = blockIdx.x * kT + threadIdx.x;
    if (u >= n_ublk) return;
    const unsigned long long r = (unsigned)ur[u], c = (unsigned)uc[u];
    keys[2 * (size_t)u] = r * n_cam + c;
    vals[2 * (size_t)u] = 2 * u;
    keys[2 * (size_t)u + 1] = r == c ? n_cam * n_cam : c * n_cam + r;
    vals[2 * (size_t)u + 1] = 2 * u + 1;
}
__global__ __launch_bounds__(kT) void k_pattern_fill(const unsigned long long* __restrict__ keys, const int* __restrict__ vals, int nnzb,
                                                     unsigned long long n_cam, unsigned* __restrict__ rows, int* __restrict__ bsr_col,
                                                     int* __restrict__ bsr_diag, int* __restrict__ upos, int* __restrict__ upos_t) {
    const int b = blockIdx.x * kT + threadIdx.x;
    if (b >= nnzb) return;
    const int r = (int)(keys[b] / n_cam), c = (int)(keys[b] % n_cam);
    rows[b] = (unsigned)r;
    bsr_col[b] = c;
    const int u = vals[b] >> 1;
    if (vals[b] & 1) upos_t[u] = b;
    else { upos[u] = b; if (r == c) { upos_t[u] = b; bsr_diag[r] = b; } }
}

// The decision the host path takes before building the explicit structure.
bool explicit_affordable(int mode, int64_t n_pairs, int no) {
    return !(mode == 0 && (n_pairs > 24 * (int64_t)std::max(no, 1) || n_pairs > 1500000000LL));
}

}  // namespace

void device_place_uv(H* h, const double2* uv) {
    if (h->n_obs > 0)
        hipLaunchKernelGGL(k_place_uv, dim3(grid_for((size_t)h->n_obs)), dim3(kT), 0, h->stream, uv, h->pm_orig.p, h->setup_pm2cm, h->n_obs,
                           h->pm_uv.p, h->cm_uv.p);
    HIP_TRY(hipGetLastError());
}

void device_setup(H* h, const int64_t* cam_idx, const int64_t* pt_idx) {
    hipStream_t st = h->stream;
    const int N = (int)h->n_obs_all, P = (int)h->n_pts_all, C = (int)h->n_cam;
    const int nr = std::max(1, h->opt.nranks), rk = nr > 1 ? h->opt.rank : 0;
    SetupTrace tr;
    h->device_built = true;
    h->new2old.clear();

    // ---- observation level -------------------------------------------------------------
    Arena A(h->ws[0]);
    {
        const size_t n = (size_t)N + 1, p = (size_t)P + 2, c = (size_t)C + 2;
        A.reserve(6 * Arena::pad(4 * n) + Arena::pad(4 * p) + 8 * Arena::pad(4 * p) + Arena::pad(p) +
                  4 * Arena::pad(4 * n) + 2 * Arena::pad(4 * c) + 16 * 256);
    }
    int* info = A.get<int>(16);                       // first bad index, partition, counts for the host
    unsigned* cam32 = A.get<unsigned>(N);
    unsigned* pt32 = A.get<unsigned>(N);
    unsigned* spt = A.get<unsigned>(N);
    int* sidx = A.get<int>(N);
    unsigned* scam = A.get<unsigned>(N);
    int* pre = A.get<int>((size_t)P + 1);
    int host_info[16];
    {
        const int big = 0x7fffffff;
        HIP_TRY(hipMemcpyAsync(info, &big, sizeof(int), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_check_narrow, dim3(grid_for(N)), dim3(kT), 0, st, cam_idx, pt_idx, N, (int64_t)C, (int64_t)P, cam32, pt32, info);
        // stable sort of the observations by the caller's point id
        sort_pairs(h, pt32, spt, rocprim::counting_iterator<int>(0), sidx, (size_t)N, bits_for((uint64_t)std::max(P - 1, 1)));
        hipLaunchKernelGGL(k_run_offsets, dim3(grid_for((size_t)N + 1)), dim3(kT), 0, st, spt, N, P, pre);
        hipLaunchKernelGGL(k_gather_u32, dim3(grid_for(N)), dim3(kT), 0, st, cam32, sidx, N, scam);
        hipLaunchKernelGGL(k_partition, dim3(1), dim3(64), 0, st, pre, P, N, rk, nr, info + 1);
        HIP_TRY(hipMemcpyAsync(host_info, info, 5 * sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (host_info[0] < N) {
        // the message NumPy gives for the first offending index (bundle_adjustment.py:49)
        int64_t bad[2];
        HIP_TRY(hipMemcpy(&bad[0], cam_idx + host_info[0], sizeof(int64_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&bad[1], pt_idx + host_info[0], sizeof(int64_t), hipMemcpyDeviceToHost));
        if (bad[0] < 0 || bad[0] >= C)
            fail(SFMBA_ERR_INDEX_RANGE, "index %lld is out of bounds for axis 0 with size %lld", (long long)bad[0], (long long)C);
        fail(SFMBA_ERR_INDEX_RANGE, "index %lld is out of bounds for axis 0 with size %lld", (long long)bad[1], (long long)P);
    }
    h->pt_lo = host_info[1]; h->pt_hi = host_info[2];
    const int o_lo = host_info[3], o_hi = host_info[4];
    const int np = (int)(h->pt_hi - h->pt_lo), no = o_hi - o_lo, pt_lo = (int)h->pt_lo;
    h->n_pts = np; h->n_obs = no;
    tr.lap("dev: check + sort by point");

    // ---- renumbering, point-major, camera-major --------------------------------------------
    unsigned* min_cam = A.get<unsigned>((size_t)np + 1);
    unsigned* min_sorted = A.get<unsigned>((size_t)np + 1);
    int* old2new = A.get<int>((size_t)np + 1);
    int* cnt_new = A.get<int>((size_t)np + 1);
    int* pm2cm = A.get<int>((size_t)no + 1);
    unsigned* cmk = A.get<unsigned>((size_t)no + 1);
    int* cm2pm = A.get<int>((size_t)no + 1);
    int* cam_ptr = A.get<int>((size_t)C + 1);
    int* nch = A.get<int>((size_t)C + 1);
    int* nxt[2] = {A.get<int>((size_t)np + 1), A.get<int>((size_t)np + 1)};
    unsigned char* mark = A.get<unsigned char>((size_t)np + 1);
    int* tile_src = A.get<int>((size_t)np + 1);

    h->new2old_dev.alloc((size_t)np);
    h->pt_ptr.alloc((size_t)np + 1);
    h->pm_cam.alloc((size_t)no); h->pm_pt.alloc((size_t)no); h->pm_orig.alloc((size_t)no); h->pm_uv.alloc((size_t)no);
    h->cm_pt.alloc((size_t)no); h->cm_uv.alloc((size_t)no);
    h->cam_chunk_ptr.alloc((size_t)C + 1);
    hipLaunchKernelGGL(k_point_keys, dim3(grid_for(np)), dim3(kT), 0, st, pre, scam, pt_lo, np, (unsigned)C, min_cam);
    // points by their smallest observing camera, stable in the caller's id
    sort_pairs(h, min_cam, min_sorted, rocprim::counting_iterator<int>(0), h->new2old_dev.p, (size_t)np, bits_for((uint64_t)C));
    hipLaunchKernelGGL(k_renumber, dim3(grid_for((size_t)np + 1)), dim3(kT), 0, st, h->new2old_dev.p, pre, pt_lo, np, old2new, cnt_new);
    exclusive_scan(h, cnt_new, h->pt_ptr.p, (size_t)np + 1);
    hipLaunchKernelGGL(k_point_major, dim3(grid_for(no)), dim3(kT), 0, st, spt, sidx, scam, pre, old2new, h->pt_ptr.p,
                               pt_lo, o_lo, no, h->pm_cam.p, h->pm_pt.p, h->pm_orig.p);
    // camera-major: stable sort of the point-major list by camera
    sort_pairs(h, reinterpret_cast<const unsigned*>(h->pm_cam.p), cmk, rocprim::counting_iterator<int>(0), cm2pm, (size_t)no,
               bits_for((uint64_t)std::max(C - 1, 1)));
    hipLaunchKernelGGL(k_camera_major, dim3(grid_for(no)), dim3(kT), 0, st, cm2pm, h->pm_pt.p, no, pm2cm, h->cm_pt.p);
    h->setup_pm2cm = pm2cm;          // (lives in ws[0] until the next set-up)
    hipLaunchKernelGGL(k_run_offsets, dim3(grid_for((size_t)no + 1)), dim3(kT), 0, st, cmk, no, C, cam_ptr);
    hipLaunchKernelGGL(k_piece_counts, dim3(grid_for((size_t)C + 1)), dim3(kT), 0, st, cam_ptr, C, h->chunk, nch);
    exclusive_scan(h, nch, h->cam_chunk_ptr.p, (size_t)C + 1);
    // wave tiles: successor of every tile start, then the starts reachable from the first one
    hipLaunchKernelGGL(k_tile_next, dim3(grid_for((size_t)np + 1)), dim3(kT), 0, st, h->pt_ptr.p, np, nxt[0], mark);
    hipLaunchKernelGGL(k_tile_first, dim3(1), dim3(64), 0, st, h->pt_ptr.p, np, no, mark);
    {
        int cur = 0;
        for (unsigned round = 0; round <= bits_for((uint64_t)std::max(no, 1)); ++round, cur ^= 1)
            hipLaunchKernelGGL(k_tile_jump, dim3(grid_for((size_t)np + 1)), dim3(kT), 0, st, nxt[cur], nxt[cur ^ 1], np, mark);
    }
    // (tile starts are the offsets pt_ptr[q] of the marked points, in order)
    select_flagged(h, h->pt_ptr.p, mark, tile_src, info + 8, (size_t)np);
    HIP_TRY(hipMemcpyAsync(info + 9, h->cam_chunk_ptr.p + C, sizeof(int), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(host_info, info + 8, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    h->n_wtiles = host_info[0];
    h->n_chunks = host_info[1];
    h->wtile_beg.alloc((size_t)h->n_wtiles + 1);
    if (h->n_wtiles) HIP_TRY(hipMemcpyAsync(h->wtile_beg.p, tile_src, (size_t)h->n_wtiles * sizeof(int), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpyAsync(h->wtile_beg.p + h->n_wtiles, &no, sizeof(int), hipMemcpyHostToDevice, st));
    h->chunk_cam.alloc((size_t)h->n_chunks); h->chunk_beg.alloc((size_t)h->n_chunks + 1);
    hipLaunchKernelGGL(k_piece_fill, dim3(grid_for((size_t)C + 1)), dim3(kT), 0, st, cam_ptr, h->cam_chunk_ptr.p, C, h->chunk, no,
                       h->chunk_cam.p, h->chunk_beg.p);
    HIP_TRY(hipGetLastError());
    tr.lap("dev: orderings, chunks, wave tiles");

    // ---- explicit reduced matrix: pair lists and block pattern ---------------------------
    const SchurPlan plan = plan_schur(h);
    if (plan.mode == 1) { HIP_TRY(hipStreamSynchronize(st)); return; }
    Arena B(h->ws[1]);
    B.reserve(2 * Arena::pad(8 * ((size_t)no + 2)) + 2 * Arena::pad(8 * ((size_t)N + 2)) + 16 * 256);
    long long* pcnt = B.get<long long>((size_t)no + 1);
    long long* poff = B.get<long long>((size_t)no + 1);
    const bool others = no != N && !h->split;          // other ranks' points: only their camera pairs matter (replicated matrix)
    long long* ocnt = B.get<long long>(others ? (size_t)N + 1 : 1);
    long long* ooff = B.get<long long>(others ? (size_t)N + 1 : 1);
    hipLaunchKernelGGL(k_pair_counts, dim3(grid_for((size_t)no + 1)), dim3(kT), 0, st, h->pt_ptr.p, h->pm_pt.p, h->pm_cam.p, no, 0, 0, pcnt);
    exclusive_scan(h, pcnt, poff, (size_t)no + 1);
    long long totals[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&totals[0], poff + no, sizeof(long long), hipMemcpyDeviceToHost, st));
    if (others) {
        hipLaunchKernelGGL(k_pair_counts, dim3(grid_for((size_t)N + 1)), dim3(kT), 0, st, pre, reinterpret_cast<const int*>(spt),
                           reinterpret_cast<const int*>(scam), N, pt_lo, pt_lo + np, ocnt);
        exclusive_scan(h, ocnt, ooff, (size_t)N + 1);
        HIP_TRY(hipMemcpyAsync(&totals[1], ooff + N, sizeof(long long), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));
    const int64_t n_pairs = totals[0], n_other = totals[1];
    if (!explicit_affordable(plan.mode, n_pairs, no)) return;
    if (n_pairs > 2000000000LL || n_pairs + n_other + C > 2100000000LL)
        fail(SFMBA_ERR_INVALID_ARGUMENT, "explicit Schur complement: too many observation pairs");
    tr.lap("dev: count pairs");

    const size_t M = (size_t)C + (size_t)n_pairs + (size_t)n_other;
    Arena E(h->ws[2]);
    E.reserve(2 * Arena::pad(8 * (M + 1)) + 4 * Arena::pad(4 * (M + 1)) + 16 * 256);
    unsigned long long* keys[2] = {E.get<unsigned long long>(M + 1), E.get<unsigned long long>(M + 1)};
    int* vals[2] = {E.get<int>(M + 1), E.get<int>(M + 1)};
    // entries: the diagonal of every camera first (so that every diagonal block exists), this
    // rank's pairs in point order (value: the point), then the other ranks' pairs (value -1)
    hipLaunchKernelGGL(k_diag_keys, dim3(grid_for(C)), dim3(kT), 0, st, C, keys[0], vals[0]);
    hipLaunchKernelGGL((k_pair_fill<true>), dim3(grid_for(no)), dim3(kT), 0, st, h->pt_ptr.p, h->pm_pt.p, h->pm_cam.p, no, 0, 0, poff,
                       (unsigned long long)C, keys[0] + C, vals[0] + C);
    if (others) hipLaunchKernelGGL((k_pair_fill<false>), dim3(grid_for(N)), dim3(kT), 0, st, pre, reinterpret_cast<const int*>(spt),
                                   reinterpret_cast<const int*>(scam), N, pt_lo, pt_lo + np, ooff, (unsigned long long)C,
                                   keys[0] + C + n_pairs, vals[0] + C + n_pairs);
    sort_pairs(h, keys[0], keys[1], vals[0], vals[1], M, bits_for((uint64_t)C * (uint64_t)C));
    // block starts and kept entries -> block index and pair position of every entry
    int* head = vals[0];                                   // (the unsorted values are no longer needed)
    int* keep = E.get<int>(M + 1);
    int* ublk = E.get<int>(M + 1);
    int* kpos = reinterpret_cast<int*>(keys[0]);           // (nor are the unsorted keys)
    hipLaunchKernelGGL(k_block_flags, dim3(grid_for(M + 1)), dim3(kT), 0, st, keys[1], vals[1], M, head, keep);
    exclusive_scan(h, head, ublk, M + 1);
    exclusive_scan(h, keep, kpos, M + 1);
    // (exclusive scan of the heads counts the blocks BEFORE an entry: the block of a head at i is ublk[i])
    int n_ublk = 0;
    HIP_TRY(hipMemcpyAsync(&n_ublk, ublk + M, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (plan.mode == 0 && (int64_t)n_ublk * 2 * 288 > (8LL << 30)) return;
    tr.lap("dev: pair keys sorted");

    h->ublk_r.alloc((size_t)n_ublk); h->ublk_c.alloc((size_t)n_ublk); h->ublk_diag.alloc((size_t)C);
    h->ublk_chunk_ptr.alloc((size_t)n_ublk + 1); h->ublk_pos.alloc((size_t)n_ublk); h->ublk_pos_t.alloc((size_t)n_ublk);
    h->pair_p.alloc((size_t)n_pairs);
    const int nnzb = 2 * n_ublk - C;
    h->bsr_ptr.alloc((size_t)C + 1); h->bsr_col.alloc((size_t)nnzb); h->bsr_diag.alloc((size_t)C);
    Arena F(h->ws[4]);
    F.reserve(2 * Arena::pad(4 * ((size_t)n_ublk + 2)) + 2 * Arena::pad(16 * ((size_t)n_ublk + 1)) + 2 * Arena::pad(8 * ((size_t)n_ublk + 1)) +
              Arena::pad(8 * ((size_t)n_ublk + 1)) + 16 * 256);
    int* bcount = F.get<int>((size_t)n_ublk + 1);
    int* pch = F.get<int>((size_t)n_ublk + 1);
    unsigned long long* bkeys[2] = {F.get<unsigned long long>(2 * (size_t)n_ublk), F.get<unsigned long long>(2 * (size_t)n_ublk)};
    int* bvals[2] = {F.get<int>(2 * (size_t)n_ublk), F.get<int>(2 * (size_t)n_ublk)};
    unsigned* rows = F.get<unsigned>(2 * (size_t)n_ublk);
    hipLaunchKernelGGL(k_block_table, dim3(grid_for(M + 1)), dim3(kT), 0, st, keys[1], vals[1], M, head, ublk, kpos, (unsigned long long)C,
                       h->ublk_r.p, h->ublk_c.p, bcount, h->ublk_diag.p, h->pair_p.p);
    // pair chunks of every block
    int pchunk = 512;
    if (const char* e = getenv("SFMBA_PAIR_CHUNK")) { int v = atoi(e); if (v >= 64) pchunk = v; }
    hipLaunchKernelGGL(k_piece_counts, dim3(grid_for((size_t)n_ublk + 1)), dim3(kT), 0, st, bcount, n_ublk, pchunk, pch);
    exclusive_scan(h, pch, h->ublk_chunk_ptr.p, (size_t)n_ublk + 1);
    int n_pchunks = 0;
    HIP_TRY(hipMemcpyAsync(&n_pchunks, h->ublk_chunk_ptr.p + n_ublk, sizeof(int), hipMemcpyDeviceToHost, st));
    // full pattern, sorted by (row, column): the slot of every block and of its transpose
    hipLaunchKernelGGL(k_pattern_keys, dim3(grid_for(n_ublk)), dim3(kT), 0, st, h->ublk_r.p, h->ublk_c.p, n_ublk, (unsigned long long)C, bkeys[0], bvals[0]);
    sort_pairs(h, bkeys[0], bkeys[1], bvals[0], bvals[1], 2 * (size_t)n_ublk, bits_for((uint64_t)C * (uint64_t)C));
    hipLaunchKernelGGL(k_pattern_fill, dim3(grid_for(nnzb)), dim3(kT), 0, st, bkeys[1], bvals[1], nnzb, (unsigned long long)C, rows,
                       h->bsr_col.p, h->bsr_diag.p, h->ublk_pos.p, h->ublk_pos_t.p);
    hipLaunchKernelGGL(k_run_offsets, dim3(grid_for((size_t)nnzb + 1)), dim3(kT), 0, st, rows, nnzb, C, h->bsr_ptr.p);
    HIP_TRY(hipStreamSynchronize(st));
    h->pchunk_blk.alloc((size_t)n_pchunks); h->pchunk_beg.alloc((size_t)n_pchunks + 1);
    hipLaunchKernelGGL(k_piece_fill, dim3(grid_for((size_t)n_ublk + 1)), dim3(kT), 0, st, bcount, h->ublk_chunk_ptr.p, n_ublk, pchunk, (int)n_pairs,
                       h->pchunk_blk.p, h->pchunk_beg.p);
    // the pattern on the host, for the coarse-space set-up
    h->bsr_ptr_host.resize((size_t)C + 1); h->bsr_col_host.resize((size_t)nnzb);
    HIP_TRY(hipMemcpyAsync(h->bsr_ptr_host.data(), h->bsr_ptr.p, ((size_t)C + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(h->bsr_col_host.data(), h->bsr_col.p, (size_t)nnzb * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
    tr.lap("dev: blocks, chunks, pattern");
    h->n_ublk = n_ublk; h->n_pchunks = n_pchunks; h->nnzb = nnzb; h->n_pairs = n_pairs;
    finish_schur_setup(h, plan.want_dense);
    tr.lap("dev: solver set-up");
}

}  // namespace sfmba
