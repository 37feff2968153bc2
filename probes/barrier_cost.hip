// Probe: cost of a work-group barrier round on gfx950 with W waves, and whether per-wave work overlaps across waves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint64_t* out, int rounds, int busy_waves, int work) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t acc = lane + wave;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rounds; ++r) {
        if ((int)wave < busy_waves) {
            for (int i = 0; i < work; ++i) { acc = acc * 3u + 1u; asm volatile("" : "+v"(acc)); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 0x12345) out[1] = acc;
}
int main() {
    uint64_t* d; hipMalloc(&d, 64);
    const int rounds = 2000;
    for (int waves : {4, 8, 16}) for (int busy : {0, 1, 4, 16}) for (int work : {50, 200}) {
        if (busy > waves) continue;
        hipLaunchKernelGGL(k, dim3(1), dim3(waves * 64), 0, 0, d, rounds, busy, work);
        hipDeviceSynchronize();
        uint64_t h; hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
        printf("waves %2d busy %2d work %3d (x2 valu): %.0f cycles/round\n", waves, busy, work, (double)h / rounds);
    }
    return 0;
}
