// What bounds the product of the reduced camera matrix (6x6 blocks, block-CSR, both orientations
// stored) with a direction z + beta p at the Final shape: 13 682 block rows of 23 blocks (91 MB).
// Variants of the production kernel's inner part, each timed over back-to-back launches:
//   rows        one wavefront per block row, lane = (block slot, row of the block), as shipped
//   rows_noS    the same without the matrix loads          rows_noV   without the vector gathers
//   stream      the matrix bytes alone, same work split    stream_gs  grid-stride over 2 048 workgroups
//   flat        one wavefront per block row, lanes walk the row's 36 n numbers in 16-byte pieces
//   flat1       the same with a single, ready-made direction vector
//   hipcc --offload-arch=gfx950 -O3 -o profiles/micro/bsr_spmv profiles/micro/bsr_spmv.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct M { const int* ptr; const int* col; const double* S; const double* z; const double* p; double* q; int n; double beta; };

template <bool kS, bool kV>
__global__ __launch_bounds__(256) void k_rows(M m) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + w;
    const int sub = lane / 6, k = lane - 6 * sub;
    __shared__ double part[256];
    double acc = 0.0;
    if (c < m.n && sub < 10) {
        const int b1 = m.ptr[c + 1];
        for (int b = m.ptr[c] + sub; b < b1; b += 10) {
            const size_t col = 6 * (size_t)m.col[b];
            double2 s0 = {1, 1}, s1 = {1, 1}, s2 = {1, 1}, z0 = {1, 1}, z1 = {1, 1}, z2 = {1, 1}, p0 = {1, 1}, p1 = {1, 1}, p2 = {1, 1};
            if (kS) { const double2* s = reinterpret_cast<const double2*>(m.S + 36 * (size_t)b + 6 * k); s0 = s[0]; s1 = s[1]; s2 = s[2]; }
            if (kV) {
                const double2* zv = reinterpret_cast<const double2*>(m.z + col);
                const double2* pv = reinterpret_cast<const double2*>(m.p + col);
                z0 = zv[0]; z1 = zv[1]; z2 = zv[2]; p0 = pv[0]; p1 = pv[1]; p2 = pv[2];
            }
            acc += s0.x * (z0.x + m.beta * p0.x) + s0.y * (z0.y + m.beta * p0.y) + s1.x * (z1.x + m.beta * p1.x) +
                   s1.y * (z1.y + m.beta * p1.y) + s2.x * (z2.x + m.beta * p2.x) + s2.y * (z2.y + m.beta * p2.y);
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (c < m.n && lane < 6) {
        double q = 0.0;
        for (int s = 0; s < 10; ++s) q += part[64 * w + lane + 6 * s];
        m.q[6 * (size_t)c + lane] = q;
    }
}


// as k_rows, but a lane loads only its own entry of z and p of the block's column; the six entries
// of the direction meet in LDS
__global__ __launch_bounds__(256) void k_rows_lds(M m) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + w;
    const int sub = lane / 6, k = lane - 6 * sub;
    __shared__ double part[256];
    __shared__ double2 dir[4][32];
    double* mine = reinterpret_cast<double*>(dir[w]);
    double acc = 0.0;
    if (c < m.n) {
        const int b0 = m.ptr[c], b1 = m.ptr[c + 1];
        for (int base = b0; base < b1; base += 10) {
            const int b = base + sub;
            const bool ok = sub < 10 && b < b1;
            double2 s0 = {0, 0}, s1 = {0, 0}, s2 = {0, 0};
            double dk = 0.0;
            if (ok) {
                const size_t col = 6 * (size_t)m.col[b];
                const double2* s = reinterpret_cast<const double2*>(m.S + 36 * (size_t)b + 6 * k);
                s0 = s[0]; s1 = s[1]; s2 = s[2];
                dk = m.z[col + k] + m.beta * m.p[col + k];
            }
            mine[lane] = dk;
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
            if (ok) {
                const double2* dv = dir[w] + 3 * sub;
                const double2 d0 = dv[0], d1 = dv[1], d2 = dv[2];
                acc += s0.x * d0.x + s0.y * d0.y + s1.x * d1.x + s1.y * d1.y + s2.x * d2.x + s2.y * d2.y;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
            __builtin_amdgcn_wave_barrier();
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (c < m.n && lane < 6) {
        double q = 0.0;
        for (int s = 0; s < 10; ++s) q += part[64 * w + lane + 6 * s];
        m.q[6 * (size_t)c + lane] = q;
    }
}

__global__ __launch_bounds__(256) void k_stream(M m) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + w;
    double acc = 0.0;
    if (c < m.n) {
        const int e0 = 18 * m.ptr[c], e1 = 18 * m.ptr[c + 1];
        const double2* s = reinterpret_cast<const double2*>(m.S);
        for (int e = e0 + lane; e < e1; e += 64) { const double2 v = s[e]; acc += v.x + v.y; }
    }
    if (acc == 12345.678) m.q[0] = acc;
}
__global__ __launch_bounds__(256) void k_stream_gs(M m, size_t n2) {
    const double2* s = reinterpret_cast<const double2*>(m.S);
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (size_t)gridDim.x * 256) { const double2 v = s[i]; acc += v.x + v.y; }
    if (acc == 12345.678) m.q[0] = acc;
}


// the head of the production kernel: a state record, then the sum of n_part partials by the whole
// workgroup (two dependent memory round trips and four barriers before the first matrix load)
__device__ __forceinline__ double head(const double* state, const double* part, int n_part, double* red) {
    const double s0 = state[0];
    if (s0 == 12345.678) return 0.0;
    double v = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 256) v += part[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const double t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t * 1e-30 + s0;
}

// production form: head first, then one row per wavefront
__global__ __launch_bounds__(256) void k_rows_head(M m, const double* state, const double* partials, int n_part) {
    __shared__ double red[4];
    __shared__ double part[256];
    const double beta = head(state, partials, n_part, red);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 4 + w;
    const int sub = lane / 6, k = lane - 6 * sub;
    double acc = 0.0;
    if (c < m.n && sub < 10) {
        const int b1 = m.ptr[c + 1];
        for (int b = m.ptr[c] + sub; b < b1; b += 10) {
            const size_t col = 6 * (size_t)m.col[b];
            const double2* s = reinterpret_cast<const double2*>(m.S + 36 * (size_t)b + 6 * k);
            const double2* zv = reinterpret_cast<const double2*>(m.z + col);
            const double2* pv = reinterpret_cast<const double2*>(m.p + col);
            const double2 s0 = s[0], s1 = s[1], s2 = s[2], z0 = zv[0], z1 = zv[1], z2 = zv[2], p0 = pv[0], p1 = pv[1], p2 = pv[2];
            acc += s0.x * (z0.x + beta * p0.x) + s0.y * (z0.y + beta * p0.y) + s1.x * (z1.x + beta * p1.x) +
                   s1.y * (z1.y + beta * p1.y) + s2.x * (z2.x + beta * p2.x) + s2.y * (z2.y + beta * p2.y);
        }
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    if (c < m.n && lane < 6) {
        double q = 0.0;
        for (int s = 0; s < 10; ++s) q += part[64 * w + lane + 6 * s];
        m.q[6 * (size_t)c + lane] = q;
    }
}

// resident form: as many workgroups as the chip holds at once, a wavefront walks every
// (number of wavefronts)-th row and requests the loads of its next row before it works on the
// current one; the head is paid once per workgroup and overlaps the first row's loads
template <int kWaves>
__global__ __launch_bounds__(64 * kWaves) void k_rows_resident(M m, const double* state, const double* partials, int n_part) {
    __shared__ double red[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sub = lane / 6, k = lane - 6 * sub;
    const int stride = gridDim.x * kWaves;
    int c = blockIdx.x * kWaves + w;
    double2 sm[3][3], zv[3][3], pv[3][3];
    bool ok[3];
    auto request = [&](int row) {
        const int b0 = row < m.n && sub < 10 ? m.ptr[row] : 0, b1 = row < m.n && sub < 10 ? m.ptr[row + 1] : 0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int b = b0 + sub + 10 * t;
            ok[t] = b < b1;
            if (ok[t]) {
                const size_t col = 6 * (size_t)m.col[b];
                const double2* s = reinterpret_cast<const double2*>(m.S + 36 * (size_t)b + 6 * k);
                sm[t][0] = s[0]; sm[t][1] = s[1]; sm[t][2] = s[2];
                const double2* a = reinterpret_cast<const double2*>(m.z + col);
                const double2* bb = reinterpret_cast<const double2*>(m.p + col);
                zv[t][0] = a[0]; zv[t][1] = a[1]; zv[t][2] = a[2];
                pv[t][0] = bb[0]; pv[t][1] = bb[1]; pv[t][2] = bb[2];
            }
        }
    };
    request(c);
    const double beta = head(state, partials, n_part, red);
    for (; c < m.n; c += stride) {
        double acc = 0.0;
#pragma unroll
        for (int t = 0; t < 3; ++t)
            if (ok[t])
#pragma unroll
                for (int i = 0; i < 3; ++i)
                    acc += sm[t][i].x * (zv[t][i].x + beta * pv[t][i].x) + sm[t][i].y * (zv[t][i].y + beta * pv[t][i].y);
        request(c + stride);
        // sum over the ten slots of a row: lanes k, k + 6, ..., k + 54
        double q = 0.0;
#pragma unroll
        for (int s = 0; s < 10; ++s) q += __shfl(acc, (lane % 6) + 6 * s);
        if (lane < 6) m.q[6 * (size_t)c + lane] = q;
    }
}


// a launch of the production grid that finds the solve finished: one state record, then nothing
struct State { double rz, rz0, tol2, tol2_loose, q_abs, dq1, dq2; int iter, max_iter, done, pad; long long ticks; };
__global__ __launch_bounds__(256) void k_done(const State* st, State* out) {
    const State s = st[0];
    if (s.done) { if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = s; return; }
    out[1] = s;
}

__global__ __launch_bounds__(256) void k_done_field(const State* st, State* out) {
    if (st[0].done) { if (blockIdx.x == 0 && threadIdx.x == 0) out[0].done = 1; return; }
    out[1] = st[0];
}
__global__ __launch_bounds__(256) void k_done_nostore(const State* st, State* out) {
    const State s = st[0];
    if (s.done) return;
    out[1] = s;
}
__global__ __launch_bounds__(256) void k_done_restrict(const State* __restrict__ st, State* __restrict__ out) {
    const State s = st[0];
    if (s.done) { if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = s; return; }
    out[1] = s;
}

__global__ __launch_bounds__(256) void k_vload(const int* st, int* out) {
    const int v = __builtin_nontemporal_load(st + 16 + (threadIdx.x & 1));       // a vector load (the address differs between lanes)
    if (v) return;
    out[40] = v;
}
__global__ __launch_bounds__(256) void k_vload_lds(const int* st, int* out) {
    __shared__ int keep[256];
    const int v = st[16 + (threadIdx.x & 1)];
    keep[threadIdx.x] = v;
    __syncthreads();
    if (keep[threadIdx.x ^ 1]) return;
    out[40] = v;
}
struct Big { double* a[180]; const State* st; State* out; };
__global__ __launch_bounds__(256) void k_done_big(Big b) {
    const State s = b.st[0];
    if (s.done) { if (blockIdx.x == 0 && threadIdx.x == 0) b.out[0] = s; return; }
    b.out[1] = s;
}
__global__ __launch_bounds__(256) void k_empty() {}

// lanes walk the row's numbers: piece e (16 bytes) of the row's blocks belongs to block e / 18, row
// (e % 18) / 3, columns 2 (e % 3) and the next
template <bool kTwo, int kRowsPerWave>
__global__ __launch_bounds__(256) void k_flat(M m) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double2* s = reinterpret_cast<const double2*>(m.S);
    for (int rr = 0; rr < kRowsPerWave; ++rr) {
        const int c = (blockIdx.x * 4 + w) * kRowsPerWave + rr;
        if (c >= m.n) return;
        const int b0 = m.ptr[c], nb = m.ptr[c + 1] - b0;
        double acc[6] = {0, 0, 0, 0, 0, 0};
        for (int e = lane; e < 18 * nb; e += 64) {
            const int blk = e / 18, rem = e - 18 * blk, r = rem / 3, cc = rem - 3 * r;
            const double2 sv = s[18 * (size_t)b0 + e];
            const size_t col = 3 * (size_t)m.col[b0 + blk] + cc;
            double2 d = reinterpret_cast<const double2*>(m.z)[col];
            if (kTwo) { const double2 pv = reinterpret_cast<const double2*>(m.p)[col]; d.x += m.beta * pv.x; d.y += m.beta * pv.y; }
            const double t = sv.x * d.x + sv.y * d.y;
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] += r == i ? t : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc[i] += __shfl_xor(acc[i], o);
        }
        if (lane < 6) {
            double q = acc[0];
#pragma unroll
            for (int i = 1; i < 6; ++i) q = lane == i ? acc[i] : q;
            m.q[6 * (size_t)c + lane] = q;
        }
    }
}

int main() {
    const int n = 13682, half = 11, per = 2 * half + 1;
    std::vector<int> ptr(n + 1), col((size_t)n * per);
    for (int i = 0; i < n; ++i) {
        ptr[i] = i * per;
        std::vector<int> cs;
        for (int o = -half; o <= half; ++o) cs.push_back(((i + o) % n + n) % n);
        std::sort(cs.begin(), cs.end());
        std::copy(cs.begin(), cs.end(), col.begin() + (size_t)i * per);
    }
    ptr[n] = n * per;
    const size_t nb = (size_t)n * per;
    std::vector<double> S(nb * 36), z((size_t)n * 6), p((size_t)n * 6);
    for (size_t i = 0; i < S.size(); ++i) S[i] = ((i * 2654435761u) % 1000) * 1e-3 - 0.5;
    for (size_t i = 0; i < z.size(); ++i) { z[i] = ((i * 40503u) % 997) * 1e-3; p[i] = ((i * 9176u) % 991) * 1e-3; }
    int *dptr, *dcol; double *dS, *dz, *dp, *dq;
    CHECK(hipMalloc(&dptr, ptr.size() * 4)); CHECK(hipMalloc(&dcol, col.size() * 4)); CHECK(hipMalloc(&dS, S.size() * 8));
    CHECK(hipMalloc(&dz, z.size() * 8)); CHECK(hipMalloc(&dp, p.size() * 8)); CHECK(hipMalloc(&dq, z.size() * 8));
    CHECK(hipMemcpy(dptr, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dcol, col.data(), col.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dS, S.data(), S.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dz, z.data(), z.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dp, p.data(), p.size() * 8, hipMemcpyHostToDevice));
    M m{dptr, dcol, dS, dz, dp, dq, n, 0.37};
    // reference product on the host
    std::vector<double> ref((size_t)n * 6), got((size_t)n * 6);
    for (int i = 0; i < n; ++i)
        for (int r = 0; r < 6; ++r) {
            double a = 0.0;
            for (int b = ptr[i]; b < ptr[i + 1]; ++b)
                for (int c = 0; c < 6; ++c) a += S[36 * (size_t)b + 6 * r + c] * (z[6 * (size_t)col[b] + c] + 0.37 * p[6 * (size_t)col[b] + c]);
            ref[6 * (size_t)i + r] = a;
        }
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double moved = (double)S.size() * 8 + col.size() * 4;
    auto time = [&](const char* name, bool check, auto launch) {
        CHECK(hipMemset(dq, 0, z.size() * 8));
        for (int w = 0; w < 3; ++w) launch();
        CHECK(hipEventRecord(e0));
        const int reps = 50;
        for (int r = 0; r < reps; ++r) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double err = -1.0;
        if (check) {
            CHECK(hipMemcpy(got.data(), dq, got.size() * 8, hipMemcpyDeviceToHost));
            err = 0.0;
            for (size_t i = 0; i < got.size(); ++i) err = std::max(err, std::abs(got[i] - ref[i]));
        }
        printf("%-12s %8.2f us  %6.2f TB/s  max err %.2e\n", name, 1e3 * ms / reps, moved * reps / (ms * 1e-3) / 1e12, err);
    };
    const int wgs = (n + 3) / 4;
    time("rows", true, [&] { hipLaunchKernelGGL((k_rows<true, true>), dim3(wgs), dim3(256), 0, 0, m); });
    time("rows_lds", true, [&] { hipLaunchKernelGGL(k_rows_lds, dim3(wgs), dim3(256), 0, 0, m); });
    time("rows_noS", false, [&] { hipLaunchKernelGGL((k_rows<false, true>), dim3(wgs), dim3(256), 0, 0, m); });
    time("rows_noV", false, [&] { hipLaunchKernelGGL((k_rows<true, false>), dim3(wgs), dim3(256), 0, 0, m); });
    time("stream", false, [&] { hipLaunchKernelGGL(k_stream, dim3(wgs), dim3(256), 0, 0, m); });
    time("stream_gs", false, [&] { hipLaunchKernelGGL(k_stream_gs, dim3(2048), dim3(256), 0, 0, m, S.size() / 2); });
    double *dstate, *dpart;
    CHECK(hipMalloc(&dstate, 64)); CHECK(hipMalloc(&dpart, 4096 * 8));
    { std::vector<double> st(8, 0.37), pp(4096, 1.0); CHECK(hipMemcpy(dstate, st.data(), 64, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dpart, pp.data(), 4096 * 8, hipMemcpyHostToDevice)); }
    time("rows+head", true, [&] { hipLaunchKernelGGL(k_rows_head, dim3(wgs), dim3(256), 0, 0, m, dstate, dpart, 326); });
    for (int g : {512, 1024, 2048})
        time(g == 512 ? "res4 512" : g == 1024 ? "res4 1024" : "res4 2048", true, [&] { hipLaunchKernelGGL((k_rows_resident<4>), dim3(g), dim3(256), 0, 0, m, dstate, dpart, 326); });
    for (int g : {256, 512, 1024})
        time(g == 256 ? "res8 256" : g == 512 ? "res8 512" : "res8 1024", true, [&] { hipLaunchKernelGGL((k_rows_resident<8>), dim3(g), dim3(512), 0, 0, m, dstate, dpart, 326); });
    State hs{}; hs.done = 1; State* dst;
    CHECK(hipMalloc(&dst, 4 * sizeof(State))); CHECK(hipMemcpy(dst, &hs, sizeof(State), hipMemcpyHostToDevice));
    Big big{}; big.st = dst; big.out = dst + 2;
    time("empty 3420", false, [&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(256), 0, 0); });
    time("empty 326", false, [&] { hipLaunchKernelGGL(k_empty, dim3(326), dim3(256), 0, 0); });
    time("done 3420", false, [&] { hipLaunchKernelGGL(k_done, dim3(wgs), dim3(256), 0, 0, dst, dst + 2); });
    time("done field", false, [&] { hipLaunchKernelGGL(k_done_field, dim3(wgs), dim3(256), 0, 0, dst, dst + 2); });
    time("done nostore", false, [&] { hipLaunchKernelGGL(k_done_nostore, dim3(wgs), dim3(256), 0, 0, dst, dst + 2); });
    time("done restr", false, [&] { hipLaunchKernelGGL(k_done_restrict, dim3(wgs), dim3(256), 0, 0, dst, dst + 2); });
    time("vload", false, [&] { hipLaunchKernelGGL(k_vload, dim3(wgs), dim3(256), 0, 0, (const int*)dst, (int*)(dst + 2)); });
    time("vload lds", false, [&] { hipLaunchKernelGGL(k_vload_lds, dim3(wgs), dim3(256), 0, 0, (const int*)dst, (int*)(dst + 2)); });
    time("done big", false, [&] { hipLaunchKernelGGL(k_done_big, dim3(wgs), dim3(256), 0, 0, big); });
    time("done 326", false, [&] { hipLaunchKernelGGL(k_done, dim3(326), dim3(256), 0, 0, dst, dst + 2); });
    time("flat", true, [&] { hipLaunchKernelGGL((k_flat<true, 1>), dim3(wgs), dim3(256), 0, 0, m); });
    time("flat1", false, [&] { hipLaunchKernelGGL((k_flat<false, 1>), dim3(wgs), dim3(256), 0, 0, m); });
    time("flat x2", true, [&] { hipLaunchKernelGGL((k_flat<true, 2>), dim3((n + 7) / 8), dim3(256), 0, 0, m); });
    time("flat x4", true, [&] { hipLaunchKernelGGL((k_flat<true, 4>), dim3((n + 15) / 16), dim3(256), 0, 0, m); });
    return 0;
}
