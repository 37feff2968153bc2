// Probe: instruction issue rate of one wavefront (cycles per instruction) for dependent / independent VALU and SALU streams,
// alone on its SIMD and with 1..4 busy waves per SIMD (work-group sizes 64, 256, 512, 1024).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 4000
template <int mode>
__global__ void k(uint64_t* out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t a = lane, b = lane * 3, c = lane + 7, d = lane ^ 5, e = lane + 11, f = lane * 5, g = lane + 13, h = lane ^ 9;
        __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
      for (int rep = 0; rep < 8; ++rep) {
        if (mode == 0) {        // 8 dependent VALU (one chain)
            asm volatile("v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %0\n v_xor_b32 %0, %0, %1" : "+v"(a) : "v"(b));
        } else if (mode == 1) { // 8 independent VALU
            asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else if (mode == 2) { // 8 dependent SALU
            asm volatile("s_add_u32 s20, s20, s20\n s_xor_b32 s20, s20, s21\n s_add_u32 s20, s20, s20\n s_xor_b32 s20, s20, s21\n s_add_u32 s20, s20, s20\n s_xor_b32 s20, s20, s21\n s_add_u32 s20, s20, s20\n s_xor_b32 s20, s20, s21" ::: "scc", "s20", "s21");
        } else if (mode == 3) { // 8 independent SALU
            asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1" ::: "scc", "s20", "s21", "s22", "s23");
        } else if (mode == 4) { // alternating VALU / SALU, independent
            asm volatile("v_add_u32 %0, %0, 1\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, 1\n s_add_u32 s21, s21, 1\n v_add_u32 %2, %2, 1\n s_add_u32 s20, s20, 1\n v_add_u32 %3, %3, 1\n s_add_u32 s21, s21, 1"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "scc", "s20", "s21");
        } else if (mode == 5) { // VALU -> SGPR -> VALU dependency (v_cmp + s_bcnt1 + v_add with sgpr), x2 + 2 more
            asm volatile("v_cmp_eq_u32 vcc, %0, %1\n s_bcnt1_i32_b64 s20, vcc\n v_add_u32 %0, %0, s20\n v_xor_b32 %0, %0, %1\n v_cmp_eq_u32 vcc, %0, %1\n s_bcnt1_i32_b64 s20, vcc\n v_add_u32 %0, %0, s20\n v_xor_b32 %0, %0, %1"
                         : "+v"(a), "+v"(b) :: "vcc", "scc", "s20");
        } else if (mode == 6) { // 8 v_mul_lo_u32 independent
            asm volatile("v_mul_lo_u32 %0, %0, %0\n v_mul_lo_u32 %1, %1, %1\n v_mul_lo_u32 %2, %2, %2\n v_mul_lo_u32 %3, %3, %3\n v_mul_lo_u32 %4, %4, %4\n v_mul_lo_u32 %5, %5, %5\n v_mul_lo_u32 %6, %6, %6\n v_mul_lo_u32 %7, %7, %7"
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        } else if (mode == 7) { // 8 s_nop 0
            asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
        }
      }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0 && wave == 0) out[mode] = (t1 - t0);
    if (a + b + c + d + e + f + g + h == 0x12345678) out[31] = a;
}
int main() {
    uint64_t* d; hipMalloc(&d, 256);
    const char* names[8] = {"8 dependent VALU", "8 independent VALU", "8 dependent SALU", "8 independent SALU", "VALU/SALU alternating", "v_cmp->s_bcnt->v_add chain", "8 independent v_mul_lo_u32", "8 s_nop"};
    for (int threads : {64, 256, 512, 1024}) {
        hipMemset(d, 0, 256);
        hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, d); hipLaunchKernelGGL(k<1>, dim3(1), dim3(threads), 0, 0, d);
        hipLaunchKernelGGL(k<2>, dim3(1), dim3(threads), 0, 0, d); hipLaunchKernelGGL(k<3>, dim3(1), dim3(threads), 0, 0, d);
        hipLaunchKernelGGL(k<4>, dim3(1), dim3(threads), 0, 0, d); hipLaunchKernelGGL(k<5>, dim3(1), dim3(threads), 0, 0, d);
        hipLaunchKernelGGL(k<6>, dim3(1), dim3(threads), 0, 0, d); hipLaunchKernelGGL(k<7>, dim3(1), dim3(threads), 0, 0, d);
        hipDeviceSynchronize();
        uint64_t h[32]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        printf("work-group of %d threads (%d waves per SIMD):\n", threads, threads / 256 ? threads / 256 : 1);
        for (int m = 0; m < 8; ++m) printf("  %-30s %6.2f cycles per instruction (wave 0)\n", names[m], (double)h[m] / (ITERS * 64.0));
    }
    return 0;
}
