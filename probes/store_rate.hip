// Probe: what a wave pays to ISSUE global stores — cycles per store instruction per wave when 1 / 2 / 8 waves of every CU store at once:
// 64 lanes x dword contiguous and aligned, the same 2 bytes off alignment, 64 lanes x 2 bytes, and a record-like pattern (items of 2 or 4 bytes
// packed at 2-byte granularity: what the Chameleon encoder's emit writes).  Up to 32 stores of a wave are in flight (vmcnt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 400
template <int mode>
__global__ void k(uint8_t* buf, uint64_t stride_wg, uint64_t* out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* base = buf + blockIdx.x * stride_wg + wave * (stride_wg / 16);
    // record-like: lane offsets from a pseudo signature (60 % of the lanes 2 bytes, the others 4)
    const uint64_t sig = 0xB6DB5B6DEDB6B5DBull;
    const uint32_t below = __builtin_popcountll(sig & ((1ull << lane) - 1ull));
    const uint32_t rec_off = 4u * lane - 2u * below;
    uint32_t off = mode == 0 ? 4u * lane : mode == 1 ? 4u * lane + 2u : mode == 2 ? 2u * lane : rec_off;
    const uint32_t step = mode == 2 ? 128u : mode == 3 ? 176u : 256u;
    uint32_t v = lane * 0x01010101u;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int rep = 0; rep < 16; ++rep) {
            uint8_t* a = base + off;
            if (mode == 2) asm volatile("global_store_short %0, %1, off" ::"v"(a), "v"(v) : "memory");
            else asm volatile("global_store_dword %0, %1, off" ::"v"(a), "v"(v) : "memory");
            off += step;
        }
        asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        if ((it & 15) == 15) off -= 16u * 16u * step;          // stay inside ~64-700 KiB per wave
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) out[mode * 16 + wave] = t1 - t0;
}
int main() {
    const uint64_t stride = 16ull << 20;                        // 16 MiB per work-group, 1 MiB per wave
    uint8_t* buf; hipMalloc(&buf, 256 * stride);
    uint64_t* d; hipMalloc(&d, 2048);
    const char* names[4] = {"dword x 64 lanes, aligned", "dword x 64 lanes, +2 bytes", "short x 64 lanes", "record-like (2 / 4 byte items packed)"};
    for (int waves : {1, 2, 8}) {
        hipMemset(d, 0, 2048);
        hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, buf, stride, d); hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, buf, stride, d);
        hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, buf, stride, d); hipLaunchKernelGGL(k<3>, dim3(256), dim3(64 * waves), 0, 0, buf, stride, d);
        hipDeviceSynchronize();
        uint64_t h[256]; hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
        printf("%d wave(s) per CU storing, 256 CUs:\n", waves);
        for (int m = 0; m < 4; ++m) {
            double worst = 0;
            for (int w = 0; w < waves; ++w) { double v = (double)h[m * 16 + w] / (ITERS * 16.0); if (v > worst) worst = v; }
            printf("  %-40s %.1f cycles per store instruction per wave -> one per %.1f cycles per CU\n", names[m], worst, worst / waves);
        }
    }
    return 0;
}
