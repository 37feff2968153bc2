// Probe: cost of branches for one wavefront (cycles per branch), straight-line s_nop filler subtracted.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define ITERS 4000
template <int mode>
__global__ void k(uint64_t* out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            if (mode == 0) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");
            if (mode == 1) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_branch 1f\n s_nop 0\n1:");                 // taken, skips 1
            if (mode == 2) asm volatile("s_cmp_eq_u32 0, 1\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_cbranch_scc1 1f\n s_nop 0\n1:" ::: "scc");   // not taken
            if (mode == 3) asm volatile("s_cmp_eq_u32 0, 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_cbranch_scc1 1f\n s_nop 0\n1:" ::: "scc");   // taken
            if (mode == 4) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_cbranch_execz 1f\n s_nop 0\n1:");             // not taken
            if (mode == 5) asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_branch 1f\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n1:");   // taken, skips 32 (128 bytes)
            if (mode == 6) asm volatile("s_mov_b64 s[20:21], exec\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_and_b64 exec, exec, s[20:21]\n s_nop 0\n s_mov_b64 exec, s[20:21]" ::: "s20", "s21");   // exec writes
            if (mode == 7) asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 0\n s_nop 0\n s_nop 0\n s_waitcnt vmcnt(0)\n s_nop 0\n s_nop 0\n s_nop 0");             // idle waitcnts
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0 && wave == 0) out[mode] = (t1 - t0);
}
template <int m>
static void run(uint64_t* d, int threads, const char* name) {
    (void)hipMemset(d, 0, 256);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(threads), 0, 0, d);
    hipLaunchKernelGGL(k<m>, dim3(1), dim3(threads), 0, 0, d);
    (void)hipDeviceSynchronize();
    uint64_t h[32]; (void)hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    printf("threads %4d  %-36s %7.1f cycles per group of 8 (baseline %.1f)\n", threads, name, (double)h[m] / (ITERS * 8.0), (double)h[0] / (ITERS * 8.0));
    fflush(stdout);
}
int main(int argc, char** argv) {
    uint64_t* d; (void)hipMalloc(&d, 256);
    const int m = argc > 1 ? atoi(argv[1]) : 1, threads = argc > 2 ? atoi(argv[2]) : 64;
    switch (m) {
        case 1: run<1>(d, threads, "7 nop + s_branch taken (skip 1)"); break;
        case 2: run<2>(d, threads, "cmp + 6 nop + cbranch not taken"); break;
        case 3: run<3>(d, threads, "cmp + 6 nop + cbranch taken"); break;
        case 4: run<4>(d, threads, "7 nop + cbranch_execz not taken"); break;
        case 5: run<5>(d, threads, "7 nop + s_branch taken (skip 32)"); break;
        case 6: run<6>(d, threads, "3 exec writes + 5 nop"); break;
        case 7: run<7>(d, threads, "2 idle s_waitcnt + 6 nop"); break;
    }
    return 0;
}
