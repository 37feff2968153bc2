// Probe: does ds_read_b32 work at addresses that are not 4-byte aligned (unaligned access mode of the LDS)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[2048];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 2048; i += 64) buf[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)buf;
    uint32_t bad = 0;
    for (uint32_t off = 0; off < 16; ++off) {
        const uint32_t a = 16u * lane + off;                   // every alignment 0..15
        uint32_t r;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(base + a) : "memory");
        const uint32_t want = (uint32_t)buf[a] | ((uint32_t)buf[a + 1] << 8) | ((uint32_t)buf[a + 2] << 16) | ((uint32_t)buf[a + 3] << 24);
        if (r != want) bad |= 1u << off;
    }
    if (bad) atomicOr(out, bad);
    if (lane == 0) out[1] = 1;
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 8); (void)hipMemset(d, 0, 8);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipDeviceSynchronize();
    uint32_t h[2]; (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("ran %u, mismatch mask by (address %% 16): 0x%04x  (0 = ds_read_b32 works at every byte alignment)\n", h[1], h[0]);
    return 0;
}
