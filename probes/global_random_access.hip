// Probe: random 4-byte accesses to per-wavefront private regions in global memory (the shape of Cheetah/Lion tables:
// 768 KiB .. 1.75 MiB per stream).  Reports accesses/s for atomic exchange (returning), dependent and independent,
// as a function of the number of streams (working set) — decides between lane-per-stream and wave-per-stream designs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// mode 0: independent atomic exchanges (address from a counter hash), 1: dependent chain (next address from returned value),
// mode 2: independent plain load (glc) + store, 3: lane-per-stream layout (each LANE has its own region), independent xchg
template <int MODE>
__global__ __launch_bounds__(64) void hammer(uint32_t* __restrict__ mem, uint64_t region_words, uint32_t iters, uint32_t* __restrict__ sink) {
    const uint32_t lane = threadIdx.x;
    const uint64_t stream = MODE == 3 ? (uint64_t)blockIdx.x * 64 + lane : blockIdx.x;
    uint32_t* base = mem + stream * region_words;
    uint32_t x = (uint32_t)(stream * 2654435761u) ^ (lane * 40503u) ^ 0x9e3779b9u;
    uint32_t acc = 0;
    const uint32_t mask = (uint32_t)region_words - 1u;
    for (uint32_t i = 0; i < iters; ++i) {
        x = x * 1664525u + 1013904223u;
        uint32_t a = (x >> 8) & mask;
        if (MODE == 1) a = (a + acc) & mask;
        if (MODE == 2) {
            const uint32_t v = __builtin_nontemporal_load(base + a);
            base[a] = v + i;
            acc += v;
        } else {
            const uint32_t old = __hip_atomic_exchange(base + a, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc += old;
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}

template <int MODE>
static int run(const char* name, uint32_t* mem, uint64_t region_words, uint32_t streams, uint32_t iters, uint32_t* sink) {
    const uint32_t blocks = MODE == 3 ? streams / 64 : streams;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(hammer<MODE>, dim3(blocks), dim3(64), 0, 0, mem, region_words, iters / 4, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(hammer<MODE>, dim3(blocks), dim3(64), 0, 0, mem, region_words, iters, sink);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    const double n = (double)blocks * 64.0 * iters;
    printf("%-28s streams %6u region %5llu KiB set %8.1f MiB  %8.3f ms  %8.2f G acc/s  (%.1f ns per wave-step)\n", name, streams,
           (unsigned long long)(region_words * 4 / 1024), (double)streams * region_words * 4 / 1048576.0, ms, n / ms * 1e-6, ms * 1e6 / iters);
    return 0;
}

int main() {
    const uint64_t region_words = 768 * 1024 / 4 / 3 * 4;   // 1 MiB regions would alias nicely; use 256 Ki words = 1 MiB
    (void)region_words;
    const uint64_t words = 262144;                           // 1 MiB per stream (power of two for the mask)
    const uint32_t max_streams = 16384;
    uint32_t *mem, *sink;
    CK(hipMalloc(&mem, (size_t)max_streams * words * 4));    // 16 GiB
    CK(hipMemset(mem, 0, (size_t)max_streams * words * 4));
    CK(hipMalloc(&sink, 64));
    for (uint32_t s : {64u, 256u, 512u, 1024u, 4096u}) {
        if (run<0>("xchg independent (wave)", mem, words, s, 2000, sink)) return 1;
        if (run<1>("xchg dependent (wave)", mem, words, s, 500, sink)) return 1;
        if (run<2>("load+store (wave)", mem, words, s, 2000, sink)) return 1;
    }
    for (uint32_t s : {1024u, 4096u, 16384u}) {
        if (run<3>("xchg independent (lane-per-stream)", mem, words, s, 500, sink)) return 1;
    }
    return 0;
}
