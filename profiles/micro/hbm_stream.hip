// What this GPU's HBM delivers to plain streaming kernels, as a yardstick for the roofline
// fractions of the solver's kernels: a read-only sum, a copy and a 24-byte gather-free triad over
// 2 GiB arrays, double2 / double4 accesses, one element per thread and grid-stride variants.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/micro/hbm_stream profiles/micro/hbm_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_read(const double4* __restrict__ a, size_t n, double* __restrict__ out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double4 v = a[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678) out[0] = s;       // never true: keeps the loads alive
}
__global__ __launch_bounds__(256) void k_copy(const double4* __restrict__ a, double4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_copy_flat(const double4* __restrict__ a, double4* __restrict__ b, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}

int main() {
    const size_t bytes = (size_t)2 << 30, n = bytes / sizeof(double4);
    double4 *a, *b; double* out;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&out, 8));
    CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 0, bytes));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time = [&](const char* name, double moved, auto launch) {
        for (int w = 0; w < 2; ++w) launch();
        CHECK(hipEventRecord(e0));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %8.1f us  %7.2f TB/s\n", name, 1e3 * ms / reps, moved * reps / (ms * 1e-3) / 1e12);
    };
    for (int blocks : {2048, 8192, 32768})  {
        char nm[64];
        snprintf(nm, sizeof nm, "read  grid-stride %d", blocks);
        time(nm, (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, a, n, out); });
        snprintf(nm, sizeof nm, "copy  grid-stride %d", blocks);
        time(nm, 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, n); });
    }
    time("copy  one element/thread", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n); });
    time("hipMemcpyAsync d2d", 2.0 * bytes, [&] { CHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
    CHECK(hipGetLastError());
    return 0;
}
