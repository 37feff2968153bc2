// Probe: serialisation order of same-address LDS atomics inside ONE wave64 instruction on gfx950.
// Not product code. Build: hipcc --offload-arch=gfx950 -O3 -o lds_order lds_order.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>

typedef __attribute__((address_space(3))) uint32_t lds_u32;

__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ uint32_t mskor_rtn(uint32_t addr, uint32_t mask, uint32_t val) {
    uint32_t r;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(mask), "v"(val) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t xchg_rtn(uint32_t addr, uint32_t val) {
    uint32_t r;
    asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(val) : "memory");
    return r;
}
__device__ __forceinline__ uint32_t or_rtn(uint32_t addr, uint32_t val) {
    uint32_t r;
    asm volatile("ds_or_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(addr), "v"(val) : "memory");
    return r;
}

extern __shared__ __attribute__((aligned(16))) uint32_t smem[];

// test 0: all lanes same dword. out[0..63]=xchg returns, [64..127]=mskor (same half), [128..191]=mskor alternating halves,
// [192..255] = or_rtn returns, [256..319] = final value after plain ds_write_b16 from all lanes / b32
__global__ void k_same(uint32_t* out) {
    int lane = threadIdx.x;
    uint32_t base = lds_addr(smem);
    if (lane == 0) { smem[0] = 0; smem[1] = 0; smem[2] = 0; smem[3] = 0; smem[4]=0; smem[5]=0; }
    __syncthreads();
    out[lane] = xchg_rtn(base, lane + 1);
    out[64 + lane] = mskor_rtn(base + 4, 0xFFFFu, lane + 1);
    uint32_t sh = (lane & 1) * 16;
    out[128 + lane] = mskor_rtn(base + 8, 0xFFFFu << sh, (uint32_t)(lane + 1) << sh);
    out[192 + lane] = or_rtn(base + 12, lane < 32 ? (1u << lane) : 0u);
    __syncthreads();
    ((volatile uint16_t*)smem)[8] = (uint16_t)(lane + 1);   // smem[4] low half
    smem[5] = lane + 1;
    __syncthreads();
    if (lane == 0) { out[256] = smem[0]; out[257] = smem[1]; out[258] = smem[2]; out[259] = smem[3]; out[260] = smem[4]; out[261] = smem[5]; }
}

// test 1: randomised Chameleon-like step. Each wave (block of 64) gets 64 (h,e) pairs, runs ITER blocks against a
// 64Ki x u16 table + 64Ki-bit bitmap in LDS using mskor_rtn/or_rtn, and records returned old entry + old valid bit.
// Host emulates sequentially (ascending lane order) and compares.
__global__ void k_rand(const uint32_t* __restrict__ hs, const uint32_t* __restrict__ es, uint32_t* __restrict__ ret, int iters) {
    int lane = threadIdx.x;
    uint32_t tbl = lds_addr(smem);            // 128 KiB table
    uint32_t bmp = tbl + 131072;               // 8 KiB bitmap
    for (int i = lane; i < (131072 + 8192) / 4; i += 64) smem[i] = 0;
    __syncthreads();
    size_t off = (size_t)blockIdx.x * iters * 64;
    for (int it = 0; it < iters; ++it) {
        uint32_t h = hs[off + it * 64 + lane], e = es[off + it * 64 + lane];
        uint32_t sh = (h & 1) * 16;
        uint32_t r = mskor_rtn(tbl + (h >> 1) * 4, 0xFFFFu << sh, e << sh);
        uint32_t v = or_rtn(bmp + (h >> 5) * 4, 1u << (h & 31));
        ret[off + it * 64 + lane] = ((r >> sh) & 0xFFFF) | (((v >> (h & 31)) & 1) << 16);
    }
}

// test 2: throughput / latency of mskor_rtn with random addresses. mode 0: dependent wait each; mode 1: 8 in flight.
__global__ void k_tput(const uint32_t* __restrict__ hs, uint32_t* __restrict__ out, long long* cyc, int iters, int mode) {
    int lane = threadIdx.x & 63;
    uint32_t tbl = lds_addr(smem);
    for (int i = threadIdx.x; i < 131072 / 4; i += blockDim.x) smem[i] = 0;
    __syncthreads();
    uint32_t acc = 0;
    uint32_t h0 = hs[threadIdx.x];
    long long t0 = clock64();
    if (mode == 0) {
        for (int it = 0; it < iters; ++it) {
            uint32_t h = (h0 * 2654435761u + it * 40503u) >> 16; h0 += 0x9E3779B9u;
            uint32_t sh = (h & 1) * 16;
            acc += mskor_rtn(tbl + (h >> 1) * 4, 0xFFFFu << sh, (h ^ it) << sh);
        }
    } else {
        for (int it = 0; it < iters; it += 8) {
            uint32_t r[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                uint32_t h = (h0 * 2654435761u + (it + u) * 40503u) >> 16; h0 += 0x9E3779B9u;
                uint32_t sh = (h & 1) * 16;
                asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3" : "=&v"(r[u]) : "v"(tbl + (h >> 1) * 4), "v"(0xFFFFu << sh), "v"((h ^ it) << sh) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) :: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += r[u];
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    uint32_t* d_out; CK(hipMalloc(&d_out, 4096));
    CK(hipFuncSetAttribute((const void*)k_same, hipFuncAttributeMaxDynamicSharedMemorySize, 1024));
    hipLaunchKernelGGL(k_same, dim3(1), dim3(64), 1024, 0, d_out);
    CK(hipDeviceSynchronize());
    uint32_t h_out[320]; CK(hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost));
    const char* names[4] = {"xchg_rtn same dword", "mskor_rtn same half", "mskor_rtn alternating halves", "or_rtn"};
    for (int t = 0; t < 4; ++t) {
        printf("%s:", names[t]);
        for (int i = 0; i < 64; ++i) printf(" %x", h_out[t * 64 + i]);
        printf("\n");
    }
    printf("finals: %x %x %x %x  write_b16 winner=%x write_b32 winner=%x\n", h_out[256], h_out[257], h_out[258], h_out[259], h_out[260], h_out[261]);
    bool asc = true;
    for (int i = 0; i < 64; ++i) { if (h_out[i] != (uint32_t)i) asc = false; if (h_out[64 + i] != (uint32_t)i) asc = false; }
    printf("ASCENDING_LANE_ORDER=%d\n", (int)asc);

    // randomised test
    const int waves = 512, iters = 256; size_t n = (size_t)waves * iters * 64;
    std::vector<uint32_t> hs(n), es(n), ret(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return (uint32_t)((s * 0x2545F4914F6CDD1Dull) >> 32); };
    for (int w = 0; w < waves; ++w) {
        int hbits = 1 + (w % 16);   // 2..65536 distinct hashes -> from extreme to mild collision rates
        int ebits = 1 + ((w / 16) % 16);
        for (size_t i = 0; i < (size_t)iters * 64; ++i) {
            hs[(size_t)w * iters * 64 + i] = rnd() & ((1u << hbits) - 1);
            es[(size_t)w * iters * 64 + i] = rnd() & ((1u << ebits) - 1);
        }
    }
    uint32_t *d_hs, *d_es, *d_ret;
    CK(hipMalloc(&d_hs, n * 4)); CK(hipMalloc(&d_es, n * 4)); CK(hipMalloc(&d_ret, n * 4));
    CK(hipMemcpy(d_hs, hs.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_es, es.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipFuncSetAttribute((const void*)k_rand, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 8192));
    hipLaunchKernelGGL(k_rand, dim3(waves), dim3(64), 131072 + 8192, 0, d_hs, d_es, d_ret, iters);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ret.data(), d_ret, n * 4, hipMemcpyDeviceToHost));
    size_t bad = 0;
    std::vector<uint16_t> tbl(65536); std::vector<uint8_t> val(65536);
    for (int w = 0; w < waves; ++w) {
        std::fill(tbl.begin(), tbl.end(), 0); std::fill(val.begin(), val.end(), 0);
        for (size_t i = 0; i < (size_t)iters * 64; ++i) {
            size_t k = (size_t)w * iters * 64 + i;
            uint32_t h = hs[k], e = es[k];
            uint32_t expect = tbl[h] | ((uint32_t)val[h] << 16);
            tbl[h] = (uint16_t)e; val[h] = 1;
            if (ret[k] != expect) { if (bad < 5) printf("mismatch w=%d i=%zu h=%x e=%x got=%x expect=%x\n", w, i, h, e, ret[k], expect); ++bad; }
        }
    }
    printf("RANDOM_SEQUENTIAL_EQUIV mismatches=%zu of %zu\n", bad, n);

    // throughput
    long long* d_cyc; CK(hipMalloc(&d_cyc, 8 * 1024));
    uint32_t* d_o2; CK(hipMalloc(&d_o2, 4 * 1024 * 1024));
    CK(hipFuncSetAttribute((const void*)k_tput, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int mode = 0; mode < 2; ++mode) for (int threads = 64; threads <= 1024; threads *= 2) {
        int it = 4096;
        hipLaunchKernelGGL(k_tput, dim3(256), dim3(threads), 131072, 0, d_hs, d_o2, d_cyc, it, mode);
        CK(hipDeviceSynchronize());
        long long c[256]; CK(hipMemcpy(c, d_cyc, sizeof(c), hipMemcpyDeviceToHost));
        double avg = 0; for (int i = 0; i < 256; ++i) avg += c[i]; avg /= 256;
        printf("tput mode=%d waves/CU=%d: %.1f clk per wave-instr, %.2f clk per CU-wide instr\n", mode, threads / 64, avg / it, avg / it / (threads / 64));
    }
    return 0;
}
