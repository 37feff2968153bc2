// common.hpp — shared device helpers for the density gfx950 kernels.  gfx950 only: wave64, LDS 160 KiB/CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace density {

// hash: (quad * 0x9D6EF916) >> 16 — chameleon.rs:14-15,89; cheetah.rs:14-15; lion.rs:14-15
constexpr uint32_t kHashMul = 0x9D6EF916u;

// kHashMul = 2 * kHalfMul with kHalfMul odd, so P = quad*kHashMul (mod 2^32) is always even and determines
// quad mod 2^31:  quad & 0x7fffffff == ((P >> 1) * kHalfMulInv) & 0x7fffffff.
constexpr uint32_t kHalfMul = kHashMul >> 1;
constexpr uint32_t inv_mod_2_32(uint32_t a) {
    uint32_t x = a;                                  // a*a == 1 (mod 8) for odd a
    for (int i = 0; i < 5; ++i) x *= 2u - a * x;     // Newton: doubles the number of correct bits
    return x;
}
constexpr uint32_t kHalfMulInv = inv_mod_2_32(kHalfMul);
static_assert((uint32_t)(kHalfMul * kHalfMulInv) == 1u, "inverse of the odd half of the hash multiplier");
static_assert((kHashMul & 1u) == 0 && (kHalfMul & 1u) == 1u, "multiplier must be 2 x odd for the 16-bit entry packing");

// blow-up protection FSM, codec/protection_state.rs:1-47.  All fields are wave-uniform (SGPRs).
struct Guard {
    uint32_t penalty = 0, start = 1, prev = 0, counter = 0;   // counter: only bits 0..3 matter (protection_state.rs:20)
    __device__ __forceinline__ bool block_is_copy() {        // revert_to_copy, :19-27
        if ((counter & 0xfu) == 0 && start > 1) start >>= 1;
        ++counter;
        return penalty > 0;
    }
    __device__ __forceinline__ void decay() { if (--penalty == 0) start = (start + 1) & 0xffu; }   // :30-35 (u8 fields)
    __device__ __forceinline__ void update(bool incompressible) {                                   // :38-47
        if (incompressible && prev) penalty = start;
        prev = incompressible ? 1u : 0u;
    }
};

typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// unaligned global accesses (the stream format is 2-byte granular; gfx950 global memory handles unaligned dwords)
typedef uint32_t u32_u __attribute__((aligned(1)));
typedef uint16_t u16_u __attribute__((aligned(1)));
typedef uint64_t u64_u __attribute__((aligned(1)));
__device__ __forceinline__ uint32_t ld32u(const uint8_t* p) { return *reinterpret_cast<const u32_u*>(p); }
__device__ __forceinline__ uint32_t ld16u(const uint8_t* p) { return *reinterpret_cast<const u16_u*>(p); }
__device__ __forceinline__ void st32u(uint8_t* p, uint32_t v) { *reinterpret_cast<u32_u*>(p) = v; }
__device__ __forceinline__ void st16u(uint8_t* p, uint32_t v) { *reinterpret_cast<u16_u*>(p) = (uint16_t)v; }

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `m` strictly below this lane
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// readfirstlane without the builtin's signed return type (a sign-extended low word corrupts 64-bit assembly)
__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t bperm(uint32_t src_lane, uint32_t v) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}

}  // namespace density
