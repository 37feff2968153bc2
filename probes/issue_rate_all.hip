// Probe: aggregate instruction issue rate of ALL waves of a work-group (cycles per instruction per wave, and instructions per cycle per SIMD)
// for VALU-only, SALU-only, mixed VALU/SALU and VALU/LDS streams at 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2000
template <int mode>
__global__ void k(uint64_t* out) {
    __shared__ uint32_t buf[4096];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t a = lane, b = lane * 3, c = lane + 7, d = lane ^ 5, e = lane + 11, f = lane * 5, g = lane + 13, h = lane ^ 9;
    buf[threadIdx.x] = lane;
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep) {
        if (mode == 0) {        // 8 independent VALU
            asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else if (mode == 1) { // 8 independent SALU
            asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" ::: "scc", "s20", "s21", "s22", "s23");
        } else if (mode == 2) { // alternating VALU / SALU
            asm volatile("v_add_u32 %0, %0, 1\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, 1\n s_add_u32 s21, s21, 1\n v_add_u32 %2, %2, 1\n s_add_u32 s20, s20, 1\n v_add_u32 %3, %3, 1\n s_add_u32 s21, s21, 1"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "scc", "s20", "s21");
        } else if (mode == 3) { // 6 VALU + 2 quarter-rate multiplies
            asm volatile("v_add_u32 %0, %0, 1\n v_mul_lo_u32 %1, %1, %1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_mul_lo_u32 %5, %5, %5\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        }
      }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) out[mode * 16 + wave] = (t1 - t0);
    if (a + b + c + d + e + f + g + h == 0x12345678) out[255] = a + buf[lane];
}
int main() {
    uint64_t* d; hipMalloc(&d, 2048);
    const char* names[4] = {"8 independent VALU", "8 independent SALU", "VALU/SALU alternating", "6 VALU + 2 v_mul_lo_u32"};
    for (int threads : {256, 512, 768, 1024}) {
        hipMemset(d, 0, 2048);
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, d); hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, d);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, d); hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, d);
        hipDeviceSynchronize();
        uint64_t h[256]; hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
        printf("work-group of %d threads (%d waves per SIMD):\n", threads, threads / 256);
        for (int m = 0; m < 4; ++m) {
            printf("  %-26s cycles per instruction, by wave:", names[m]);
            double worst = 0;
            for (int w = 0; w < threads / 64; ++w) { double v = (double)h[m * 16 + w] / (ITERS * 64.0); printf(" %.2f", v); if (v > worst) worst = v; }
            printf("  -> %.2f instructions per cycle per SIMD\n", (threads / 256) / worst);
        }
    }
    return 0;
}
