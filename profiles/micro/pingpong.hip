// Latency probe: two workgroups on different dies bounce a counter through device memory with
// agent-scope relaxed loads and stores.  Prints the round-trip time per bounce for several
// allocation kinds.  Build: hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void pingpong(unsigned long long* buf, int n, int partner_block, long long* ticks, long long limit) {
    if (threadIdx.x != 0) return;
    if (blockIdx.x != 0 && blockIdx.x != partner_block) return;
    const bool first = blockIdx.x == 0;
    unsigned long long* mine = buf + (first ? 0 : 32);
    unsigned long long* theirs = buf + (first ? 32 : 0);
    const long long t0 = wall_clock64();
    for (int i = 1; i <= n; ++i) {
        if (first) __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)i)
            if (wall_clock64() - t0 > limit) { ticks[blockIdx.x ? 1 : 0] = -1; return; }
        if (!first) __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    ticks[first ? 0 : 1] = wall_clock64() - t0;
}

// one workgroup: store then load back (same thread), n times: store-to-load turnaround
__global__ void selfloop(unsigned long long* buf, int n, long long* ticks) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long t0 = wall_clock64();
    unsigned long long acc = 0;
    for (int i = 1; i <= n; ++i) {
        __hip_atomic_store(buf, (unsigned long long)i + acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        acc += __hip_atomic_load(buf + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // dependent chain of loads
    }
    ticks[0] = wall_clock64() - t0;
    buf[1] = acc;
}

int main() {
    const int n = 2000;
    long long* ticks;
    CHECK(hipHostMalloc((void**)&ticks, 2 * sizeof(long long), hipHostMallocDefault));
    const char* names[] = {"hipMalloc", "finegrained", "uncached"};
    for (int kind = 0; kind < 3; ++kind) {
        unsigned long long* buf = nullptr;
        hipError_t e = hipSuccess;
        if (kind == 0) e = hipMalloc((void**)&buf, 4096);
        else if (kind == 1) e = hipExtMallocWithFlags((void**)&buf, 4096, hipDeviceMallocFinegrained);
        else e = hipExtMallocWithFlags((void**)&buf, 4096, hipDeviceMallocUncached);
        if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", names[kind], hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        for (int partner : {1, 8, 9}) {       // block 1: next die; block 8: same die as block 0 (round-robin)
            CHECK(hipMemset(buf, 0, 4096));
            ticks[0] = ticks[1] = 0;
            void* args[] = {&buf, (void*)&n, &partner, &ticks, nullptr};
            long long limit = 100000000LL;
            args[4] = &limit;
            CHECK(hipLaunchCooperativeKernel((const void*)pingpong, dim3(16), dim3(64), args, 0, 0));
            CHECK(hipDeviceSynchronize());
            printf("%-12s partner block %d: %.3f us per round trip (two hops)\n", names[kind], partner, ticks[0] * 0.01 / n);
        }
        CHECK(hipMemset(buf, 0, 4096));
        hipLaunchKernelGGL(selfloop, dim3(1), dim3(64), 0, 0, buf, n, ticks);
        CHECK(hipDeviceSynchronize());
        printf("%-12s dependent load + store per step: %.3f us\n", names[kind], ticks[0] * 0.01 / n);
        CHECK(hipFree(buf));
    }
    return 0;
}
