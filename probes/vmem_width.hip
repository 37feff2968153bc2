// Probe: what the texture-addresser path charges per wave-wide memory INSTRUCTION by access width.  8 waves of every CU stream loads (or stores) of
// 4 / 8 / 16 bytes per lane over a per-wave window that stays in L2 (so the answer is the CU's memory pipeline, not HBM): cycles per instruction
// per CU and bytes per cycle per CU.  Background for DESIGN.md 4.5 (round 4): rocprofv3 shows the TA of a CU busy 65-73 % of the Chameleon kernels'
// time at ~2.2 dword-per-lane instructions per 256-byte block.     hipcc --offload-arch=gfx950 -O3 -o vmem_width vmem_width.hip && ./vmem_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITERS 256
template <int BYTES, bool STORE>
__global__ __launch_bounds__(512) void k(uint8_t* buf, uint64_t* out) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint8_t* base = buf + ((uint64_t)blockIdx.x * 8 + wave) * (256u << 10) + lane * BYTES;     // 256 KiB per wave
    uint32_t off = 0;
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v4 = {lane, lane + 1, lane + 2, lane + 3};
    const u32x2 v2 = {lane, lane + 1};
    uint32_t acc = 0;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        // 16 instructions per batch, all of them in flight together, landed before anything else touches their registers (one asm statement per
        // batch: a load left in flight across statements lands in whatever the compiler has put there since)
        uint8_t* a = base + off;
        if (STORE) {
#pragma unroll
            for (int rep = 0; rep < 16; ++rep) {
                uint8_t* p = a + rep * 64 * BYTES;
                if (BYTES == 4) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v2.x) : "memory");
                else if (BYTES == 8) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v2) : "memory");
                else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v4) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else {
#define L16(OP, T, STEP) { T r0, r1, r2, r3, r4, r5, r6, r7; \
            asm volatile(OP " %0, %8, off\n\t" OP " %1, %8, off offset:" #STEP "\n\t" OP " %2, %8, off offset:2*" #STEP "\n\t" OP " %3, %8, off offset:3*" #STEP "\n\t" \
                         OP " %4, %9, off\n\t" OP " %5, %9, off offset:" #STEP "\n\t" OP " %6, %9, off offset:2*" #STEP "\n\t" OP " %7, %9, off offset:3*" #STEP "\n\t" \
                         "s_waitcnt vmcnt(0)" : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(a), "v"(a + 4 * STEP) : "memory"); \
            acc ^= *reinterpret_cast<uint32_t*>(&r0) ^ *reinterpret_cast<uint32_t*>(&r7); }
            if (BYTES == 4) { L16("global_load_dword", uint32_t, 256) L16("global_load_dword", uint32_t, 256) }
            else if (BYTES == 8) { L16("global_load_dwordx2", u32x2, 512) L16("global_load_dwordx2", u32x2, 512) }
            else { L16("global_load_dwordx4", u32x4, 1024) L16("global_load_dwordx4", u32x4, 1024) }
        }
        off = (off + 16u * 64u * BYTES) & ((256u << 10) - 1u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) out[blockIdx.x * 8 + wave] = (t1 - t0) + (acc == 0x12345u);
}
template <int BYTES, bool STORE>
void run(uint8_t* buf, uint64_t* d, const char* what) {
    hipMemset(d, 0, 2048 * 8);
    hipLaunchKernelGGL((k<BYTES, STORE>), dim3(256), dim3(512), 0, 0, buf, d);
    hipDeviceSynchronize();
    static uint64_t h[2048];
    hipMemcpy(h, d, 2048 * 8, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int i = 0; i < 2048; ++i) worst = h[i] > worst ? (double)h[i] : worst;
    const double per_cu = worst / (ITERS * 16.0) / 8.0;                                          // cycles per instruction per CU (8 waves at once)
    printf("  %-28s %6.1f cycles per wave-instruction per CU   %6.1f bytes per cycle per CU\n", what, per_cu, 64.0 * BYTES / per_cu);
}
int main() {
    uint8_t* buf; hipMalloc(&buf, 2048ull * (256u << 10));
    uint64_t* d; hipMalloc(&d, 2048 * 8);
    hipMemset(buf, 1, 2048ull * (256u << 10));
    printf("8 waves per CU, 256 CUs, 256 KiB window per wave:\n");
    run<4, false>(buf, d, "load  4 bytes per lane"); run<8, false>(buf, d, "load  8 bytes per lane"); run<16, false>(buf, d, "load 16 bytes per lane");
    run<4, true>(buf, d, "store 4 bytes per lane"); run<8, true>(buf, d, "store 8 bytes per lane"); run<16, true>(buf, d, "store 16 bytes per lane");
    return 0;
}
