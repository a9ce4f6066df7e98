// How long the host takes to notice that a kernel has finished, and to get the next one running:
// a loop of (kernel, wait, kernel, wait ...) with the wait done by hipStreamSynchronize, by
// hipEventSynchronize, or by spinning on a word the kernel writes into pinned host memory.
// The kernel spins for a given time first (a stand-in for the trial-step kernel of ~100 us), so
// that the runtime's wait sees a kernel that is still running when the wait begins.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/micro/host_sync profiles/micro/host_sync.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_work(volatile unsigned long long* ticket, unsigned long long value, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        __threadfence_system();
        *ticket = value;
    }
}


// Host time of a launch: what hipLaunchKernelGGL costs the calling thread, by stream and argument size.
struct BigArgs { double* p[170]; int n; };
__global__ void k_big(BigArgs a) { if (a.n == 12345) a.p[0][0] = 1.0; }
__global__ void k_small(int n, double* p) { if (n == 12345) p[0] = 1.0; }
template <class F>
static double host_us_per_launch(int reps, F launch, hipStream_t a, hipStream_t b) {
    for (int w = 0; w < 50; ++w) launch(w);
    CHECK(hipStreamSynchronize(a)); if (b) CHECK(hipStreamSynchronize(b));
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; ++r) launch(r);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
    CHECK(hipStreamSynchronize(a)); if (b) CHECK(hipStreamSynchronize(b));
    return us;
}
static void launch_costs() {
    hipStream_t s1, s2; CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    BigArgs big{}; big.n = 0;
    const int reps = 2000;
    printf("host time per launch: small args, one stream            %6.2f us\n", host_us_per_launch(reps, [&](int) { hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s1, 0, (double*)nullptr); }, s1, nullptr));
    printf("host time per launch: 1.4 KB of arguments, one stream   %6.2f us\n", host_us_per_launch(reps, [&](int) { hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s1, big); }, s1, nullptr));
    printf("host time per launch: 1.4 KB, two streams alternating   %6.2f us\n", host_us_per_launch(reps, [&](int r) { hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, (r & 1) ? s1 : s2, big); }, s1, s2));
    printf("host time per launch: 1.4 KB, 14 on one stream then 14 on the other %6.2f us\n", host_us_per_launch(reps, [&](int r) { hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, ((r / 14) & 1) ? s1 : s2, big); }, s1, s2));
    // a busy device: a long kernel on s1 while launches go to s2
    unsigned long long* t; CHECK(hipHostMalloc((void**)&t, 64, hipHostMallocDefault));
    hipLaunchKernelGGL(k_work, dim3(1024), dim3(256), 0, s1, t, 1ull, 100LL * 40000);     // 40 ms
    printf("host time per launch: 1.4 KB on a second stream beside a running kernel %6.2f us\n", host_us_per_launch(reps, [&](int) { hipLaunchKernelGGL(k_big, dim3(64), dim3(256), 0, s2, big); }, s2, nullptr));
    CHECK(hipStreamSynchronize(s1));
}

int main() {
    launch_costs();
    unsigned long long* ticket;
    CHECK(hipHostMalloc((void**)&ticket, 64, hipHostMallocDefault));
    *ticket = 0;
    hipStream_t st; CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t ev; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const int reps = 300;
    for (long long us : {2LL, 100LL}) {
        const long long ticks = us * 100;      // 100 MHz wall clock
        for (int mode = 0; mode < 3; ++mode) {
            unsigned long long v = *ticket;
            for (int w = 0; w < 20; ++w) { hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, st, ticket, ++v, ticks); CHECK(hipStreamSynchronize(st)); }
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < reps; ++r) {
                hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, st, ticket, ++v, ticks);
                if (mode == 0) CHECK(hipStreamSynchronize(st));
                else if (mode == 1) { CHECK(hipEventRecord(ev, st)); CHECK(hipEventSynchronize(ev)); }
                else { while (*(volatile unsigned long long*)ticket != v) {} }
            }
            const double per = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            CHECK(hipStreamSynchronize(st));
            printf("kernel %3lld us, wait by %-22s %7.2f us per round (%.2f us beyond the kernel)\n", us,
                   mode == 0 ? "hipStreamSynchronize:" : mode == 1 ? "hipEventSynchronize:" : "spinning on pinned word:", per, per - us);
        }
    }
    return 0;
}
