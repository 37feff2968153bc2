// Probe: latency of a dependent LDS look-up chain on a lone wave (the Cheetah context walk's hop: ds_read_u16 -> s_waitcnt -> v_readfirstlane
// -> address), with and without the walk's bookkeeping around it, and of a 64-lane exec-masked ds_write_b16 between hops.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
extern __shared__ uint16_t tab[];
template <int mode>
__global__ void k(uint64_t* out, int hops) {
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 65536; i += 64) tab[i] = (uint16_t)((i * 40503u + 12345u) & 0xffffu);
    __syncthreads();
    uint32_t c = 1;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int i = 0; i < hops; ++i) {
        uint32_t nx;
        if (mode == 0) {          // bare hop
            asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nx) : "v"(2u * c) : "memory");
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
        } else if (mode == 1) {   // hop + one masked 16-bit store of 4 lanes before it
            asm volatile("s_mov_b64 exec, 0xf0\n\tds_write_b16 %0, %1\n\ts_mov_b64 exec, -1" :: "v"(2u * (c + lane)), "v"(lane) : "memory");
            asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nx) : "v"(2u * c) : "memory");
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
        } else {                  // hop + ten scalar instructions (the walk's bookkeeping)
            asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1" ::: "scc", "s20", "s21");
            asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nx) : "v"(2u * c) : "memory");
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)nx);
        }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[mode] = t1 - t0; out[8 + mode] = c; }
}
int main() {
    uint64_t* d; hipMalloc(&d, 256); hipMemset(d, 0, 256);
    const int hops = 20000;
    hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 131072, 0, d, hops);
    hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 131072, 0, d, hops);
    hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 131072, 0, d, hops);
    hipDeviceSynchronize();
    uint64_t h[32]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    const char* names[3] = {"bare hop (ds_read_u16, wait, readfirstlane)", "hop + a masked 16-bit store", "hop + ten scalar instructions"};
    for (int m = 0; m < 3; ++m) printf("%-48s %7.1f cycles per hop\n", names[m], (double)h[m] / hops);
    return 0;
}
