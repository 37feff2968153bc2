// Probe: what do the building blocks of the feeder / dictionary waves cost for ONE wave alone on its SIMD (cycles per op)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 2000
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
__global__ void k(uint64_t* out, const uint8_t* g, int mode) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t acc = lane, x = lane * 7u + 1u;
    uint32_t* l32 = reinterpret_cast<uint32_t*>(smem);
    for (int i = lane; i < 4096; i += 64) l32[i] = i;
    __syncthreads();
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        switch (mode) {
            case 0: acc = acc * 3u + 1u; break;                                                       // 1 dependent VALU pair
            case 1: acc += (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane + 1) & 63) << 2), (int)acc); break;   // bpermute round trip
            case 2: acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x111, 0xf, 0xf, true); break;         // dpp add
            case 3: acc += (uint32_t)__builtin_amdgcn_readlane((int)acc, 5); break;                    // readlane -> salu -> valu
            case 4: { uint64_t b = __builtin_amdgcn_ballot_w64(acc & 1); acc += (uint32_t)__builtin_ctzll(b | 0x100); } break;   // ballot + ctz
            case 5: acc += l32[(acc + lane) & 4095]; break;                                            // dependent LDS read
            case 6: if (lane < 8) l32[lane] = acc; acc += 1; break;                                    // masked LDS store
            case 7: { uint32_t s = rfl(acc); s = (s >> 3) + 5; s = s * 3; acc += s; } break;           // rfl + 2 SALU + back
            case 8: acc += *reinterpret_cast<const volatile uint32_t*>(g + ((acc & 1023) << 2)); break;   // dependent global load (L2 hit)
            case 9: { uint32_t a = acc * 5u, b = acc ^ 77u, c = acc + 9u, d = acc >> 1; acc = (a ^ b) + (c ^ d); } break;   // 4 independent + 3
        }
        asm volatile("" : "+v"(acc));
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) out[mode] = (t1 - t0);
    if (acc == 0x12345678) out[31] = acc;
}
int main() {
    uint64_t* d; hipMalloc(&d, 256); hipMemset(d, 0, 256);
    uint8_t* g; hipMalloc(&g, 1 << 20); hipMemset(g, 1, 1 << 20);
    for (int m = 0; m < 10; ++m) hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d, g, m);
    hipDeviceSynchronize();
    uint64_t h[32]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    const char* names[10] = {"dependent mul+add", "bpermute round trip", "dpp add", "readlane+add", "ballot+ctz+add", "dependent LDS read", "masked LDS store", "rfl+2salu+add", "dependent global load (hot)", "7 valu (4 indep)"};
    for (int m = 0; m < 10; ++m) printf("%-32s %.1f cycles/iter\n", names[m], (double)h[m] / ITERS);
    return 0;
}
