// rotor_hop.hip — what a round hand-off of the wave-rotation kernels (density_amd/csrc/rotor.hip) costs at best.
// One work-group of 16 waves, 128 KiB table; every wave does nothing but: wait for the token, 8 ordered exchanges from prepared
// registers, token for the next round.  Reports cycles per round for several polling styles and address patterns, and the cost of
// the 8 exchanges alone on a lone wave.   hipcc --offload-arch=gfx950 -O3 -o rotor_hop rotor_hop.hip && ./rotor_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
constexpr uint32_t kTable = 131072, kSync = kTable, kSink = kTable + 64;

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint32_t peek1(uint32_t a) { uint32_t v; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(a) : "memory"); return v; }

#define XCHG8 \
    "ds_mskor_rtn_b32 %0, %8, %16, %24\n\tds_mskor_rtn_b32 %1, %9, %17, %25\n\tds_mskor_rtn_b32 %2, %10, %18, %26\n\tds_mskor_rtn_b32 %3, %11, %19, %27\n\t" \
    "ds_mskor_rtn_b32 %4, %12, %20, %28\n\tds_mskor_rtn_b32 %5, %13, %21, %29\n\tds_mskor_rtn_b32 %6, %14, %22, %30\n\tds_mskor_rtn_b32 %7, %15, %23, %31\n\t"
#define OPS(ret, addr, mask, val, ta, tv) \
    : "=&v"(ret[0]), "=&v"(ret[1]), "=&v"(ret[2]), "=&v"(ret[3]), "=&v"(ret[4]), "=&v"(ret[5]), "=&v"(ret[6]), "=&v"(ret[7]) \
    : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]), \
      "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]), "v"(mask[7]), \
      "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]), "v"(ta), "v"(tv) : "memory"

// mode bits: 1 = pipelined polling (2 reads in flight), 2 = s_setprio 3 while polling, 4 = no exchanges (token only), 8 = text-like addresses
// (a quarter of the lanes of a block share 4 slots), 16 = all lanes one slot, 32 = token after the answers
__global__ __launch_bounds__(1024) void hop_kernel(uint64_t* out, uint32_t rounds, uint32_t mode, uint32_t nwaves) {
    const uint32_t lane = threadIdx.x & 63u, wave = rfl(threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < kTable / 4; i += 1024) reinterpret_cast<uint32_t*>(smem)[i] = 0;
    if (threadIdx.x == 0) *reinterpret_cast<uint32_t*>(smem + kSync) = 0;
    __syncthreads();
    if (wave >= nwaves) return;
    const uint32_t tok = kSync, ta = lane == 0 ? tok : kSink + 4u * lane;
    uint32_t addr[8], mask[8], val[8], ret[8];
    uint32_t seed = threadIdx.x * 2654435761u + 12345u;
    uint64_t t_start = 0, crit = 0;
    uint32_t acc = 0;
    for (uint32_t r = wave; r < rounds; r += nwaves) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            seed = seed * 1664525u + 1013904223u;
            uint32_t h = seed >> 16;
            if ((mode & 8u) && (lane & 3u) == 0) h = (r * 8 + j) & 3u;            // repeated words
            if (mode & 16u) h = 77;
            addr[j] = (h >> 1) << 2; mask[j] = 0xffffu << ((h & 1u) << 4); val[j] = (seed & 0xffffu) << ((h & 1u) << 4);
        }
        asm volatile("" : "+v"(addr[0]), "+v"(addr[1]), "+v"(addr[2]), "+v"(addr[3]), "+v"(addr[4]), "+v"(addr[5]), "+v"(addr[6]), "+v"(addr[7]),
                          "+v"(mask[0]), "+v"(mask[1]), "+v"(mask[2]), "+v"(mask[3]), "+v"(mask[4]), "+v"(mask[5]), "+v"(mask[6]), "+v"(mask[7]),
                          "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]), "+v"(val[4]), "+v"(val[5]), "+v"(val[6]), "+v"(val[7]));
        if (mode & 2u) __builtin_amdgcn_s_setprio(3);
        if (mode & 1u) {
            uint32_t a, b;
            asm volatile("ds_read_b32 %0, %2\n\tds_read_b32 %1, %2" : "=&v"(a), "=&v"(b) : "v"(tok) : "memory");
            for (;;) {
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(a) :: "memory");
                if (rfl(a) == r) break;
                asm volatile("ds_read_b32 %0, %1" : "=&v"(a) : "v"(tok) : "memory");
                asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(b) :: "memory");
                if (rfl(b) == r) break;
                asm volatile("ds_read_b32 %0, %1" : "=&v"(b) : "v"(tok) : "memory");
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            while (rfl(peek1(tok)) != r) {}
        }
        if (r == wave) t_start = __builtin_readcyclecounter();
        const uint64_t c0 = __builtin_readcyclecounter();
        const uint32_t tv = r + 1u;
        if (mode & 4u) {
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(ta), "v"(tv) : "memory");
        } else if (mode & (64u | 128u | 256u)) {
            // round 6: the exchanges under an EXEC mask of 32 / 16 / 8 lanes (every 2nd / 4th / 8th lane) — is an ordered exchange paid per lane or per instruction?
            const uint64_t em = (mode & 64u) ? 0x5555555555555555ull : (mode & 128u) ? 0x1111111111111111ull : 0x0101010101010101ull;
            asm volatile("s_mov_b64 exec, %34\n\t" XCHG8 "s_mov_b64 exec, -1\n\tds_write_b32 %32, %33\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(ret[0]), "=&v"(ret[1]), "=&v"(ret[2]), "=&v"(ret[3]), "=&v"(ret[4]), "=&v"(ret[5]), "=&v"(ret[6]), "=&v"(ret[7])
                : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),
                  "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]), "v"(mask[7]),
                  "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]), "v"(ta), "v"(tv), "s"(em) : "memory");
        } else if (mode & 32u) {
            asm volatile(XCHG8 "s_waitcnt lgkmcnt(0)\n\tds_write_b32 %32, %33" OPS(ret, addr, mask, val, ta, tv));
        } else {
            asm volatile(XCHG8 "ds_write_b32 %32, %33\n\ts_waitcnt lgkmcnt(0)" OPS(ret, addr, mask, val, ta, tv));
        }
        crit += __builtin_readcyclecounter() - c0;
        if (mode & 2u) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= ret[j];
    }
    const uint64_t t_end = __builtin_readcyclecounter();
    if (lane == 0) { out[3 * wave] = t_end - t_start; out[3 * wave + 1] = crit; out[3 * wave + 2] = acc; }
}

// Two dictionary chains by slot parity: the lanes whose slot is even exchange under token A, the odd ones under token B; the chains only
// order exchanges among their own lanes, so wave w+1 may run chain A of its round while wave w is still in chain B of its own.
__global__ __launch_bounds__(1024) void hop2_kernel(uint64_t* out, uint32_t rounds, uint32_t mode, uint32_t nwaves) {
    const uint32_t lane = threadIdx.x & 63u, wave = rfl(threadIdx.x >> 6);
    for (uint32_t i = threadIdx.x; i < kTable / 4; i += 1024) reinterpret_cast<uint32_t*>(smem)[i] = 0;
    if (threadIdx.x == 0) { *reinterpret_cast<uint32_t*>(smem + kSync) = 0; *reinterpret_cast<uint32_t*>(smem + kSync + 16) = 0; }
    __syncthreads();
    if (wave >= nwaves) return;
    uint32_t addr[8], mask[8], val[8], ret[8];
    uint64_t par[8];
    uint32_t seed = threadIdx.x * 2654435761u + 12345u;
    uint64_t t_start = 0;
    uint32_t acc = 0;
    for (uint32_t r = wave; r < rounds; r += nwaves) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            seed = seed * 1664525u + 1013904223u;
            uint32_t h = seed >> 16;
            if ((mode & 8u) && (lane & 3u) == 0) h = (r * 8 + j) & 3u;
            addr[j] = (h >> 1) << 2; mask[j] = 0xffffu << ((h & 1u) << 4); val[j] = (seed & 0xffffu) << ((h & 1u) << 4);
            par[j] = __builtin_amdgcn_ballot_w64((h & 2u) != 0);                 // chain by a slot bit that is not the half-of-dword bit
        }
        if (r == wave) t_start = __builtin_readcyclecounter();
        for (int c = 0; c < 2; ++c) {
            const uint32_t tok = kSync + 16u * c, ta = lane == 0 ? tok : kSink + 4u * lane;
            while (rfl(peek1(tok)) != r) {}
            const uint32_t tv = r + 1u;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint64_t m = c ? par[j] : ~par[j];
                asm volatile("s_mov_b64 exec, %4\n\tds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_mov_b64 exec, -1" : "+v"(ret[j]) : "v"(addr[j]), "v"(mask[j]), "v"(val[j]), "s"(m) : "memory");
            }
            asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(ta), "v"(tv) : "memory");
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= ret[j];
    }
    const uint64_t t_end = __builtin_readcyclecounter();
    if (lane == 0) { out[3 * wave] = t_end - t_start; out[3 * wave + 1] = 0; out[3 * wave + 2] = acc; }
}

int main() {
    uint64_t* d;
    hipMalloc((void**)&d, 64 * 8);
    hipFuncSetAttribute((const void*)hop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kTable + 512);
    const uint32_t rounds = 16 * 256;
    struct { uint32_t mode, waves; const char* what; } cases[] = {
        {0, 1, "lone wave, random slots"}, {8, 1, "lone wave, text-like slots"}, {16, 1, "lone wave, one slot"}, {4, 1, "lone wave, token only"},
        {0, 16, "16 waves, simple poll, random"}, {8, 16, "16 waves, simple poll, text-like"}, {1, 16, "16 waves, pipelined poll, random"},
        {2, 16, "16 waves, simple poll + setprio, random"}, {3, 16, "16 waves, pipelined poll + setprio, random"},
        {4, 16, "16 waves, token only, simple poll"}, {5, 16, "16 waves, token only, pipelined poll"}, {32, 16, "16 waves, token after answers, random"},
        {0, 4, "4 waves, simple poll, random"}, {0, 2, "2 waves, simple poll, random"},
        // round 6 (price the ordered exchange by active lanes before building two dictionary tokens)
        {64, 1, "lone wave, random, 32 of 64 lanes active"}, {128, 1, "lone wave, random, 16 of 64 lanes active"}, {256, 1, "lone wave, random, 8 of 64 lanes active"},
        {64 | 8, 1, "lone wave, text-like, 32 lanes active"},
        {64, 16, "16 waves, simple poll, random, 32 lanes active"}, {128, 16, "16 waves, simple poll, random, 16 lanes active"},
        {64 | 8, 16, "16 waves, simple poll, text-like, 32 lanes"},
    };
    for (auto& c : cases) {
        hipMemset(d, 0, 64 * 8);
        hipLaunchKernelGGL(hop_kernel, dim3(1), dim3(1024), kTable + 512, 0, d, rounds, c.mode, c.waves);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(64);
        hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
        double tot = 0, crit = 0;
        for (uint32_t w = 0; w < c.waves; ++w) { tot = h[3 * w] > tot ? h[3 * w] : tot; crit += h[3 * w + 1]; }
        printf("%-48s %7.1f cycles per round, critical section (incl. answers) %6.1f\n", c.what, tot / rounds, crit / rounds);
    }
    hipFuncSetAttribute((const void*)hop2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kTable + 512);
    for (uint32_t mode : {0u, 8u}) {
        hipMemset(d, 0, 64 * 8);
        hipLaunchKernelGGL(hop2_kernel, dim3(1), dim3(1024), kTable + 512, 0, d, rounds, mode, 16);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(64);
        hipMemcpy(h.data(), d, 64 * 8, hipMemcpyDeviceToHost);
        double tot = 0;
        for (uint32_t w = 0; w < 16; ++w) tot = h[3 * w] > tot ? h[3 * w] : tot;
        printf("%-48s %7.1f cycles per round\n", mode ? "16 waves, TWO chains by slot bit, text-like" : "16 waves, TWO chains by slot bit, random", tot / rounds);
    }
    return 0;
}
