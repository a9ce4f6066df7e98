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

int main() {
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
