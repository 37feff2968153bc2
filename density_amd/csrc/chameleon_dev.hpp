// chameleon_dev.hpp — device-side pieces shared by the Chameleon kernels of chameleon.hip (one-wavefront kernels and the
// 16-wave role pipelines) and rotor.hip (wave-rotation kernels): the exact 16-bit dictionary entry, the LDS dictionary
// primitives, the zero-entry maps, the in-order decode loop and the ragged-tail encoder.  See chameleon.hip for the design notes.
#pragma once
#include "common.hpp"

namespace density {

extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

namespace {

constexpr uint32_t kTableBytes = 65536u * 2u;   // 64 Ki x u16 entries
constexpr uint32_t kZmapBytes = 65536u / 8u;    // 1 bit per slot
constexpr uint32_t kLdsBytes = kTableBytes + kZmapBytes;
constexpr uint32_t kBlock = 256;                // chameleon.rs:140
constexpr uint32_t kSig = 8;                    // chameleon.rs:146
constexpr uint32_t kIdxCopy = 0x80u, kIdxRagged = 0x7fu;   // block index entry: bit 7 raw copy; low 7 bits MAP count or 0x7f = ragged (include/density_hip.h)

__device__ __forceinline__ void lds_clear(uint32_t lane) {
    uint4* p = reinterpret_cast<uint4*>(smem);
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (uint32_t i = lane; i < kLdsBytes / 16; i += 64) p[i] = z;
    __syncthreads();
}

// inverse of the entry packing: slot index h and 16-bit entry e -> the quad (see file header)
// Stored entries are salted per slot: stored = packed ^ slot_salt(h).  The one stored value that aliases "never written"
// (0) then belongs to one pseudo-random quad per slot instead of to every quad whose packed entry is 0 (low 15 bits and top
// bit clear: round floats, small little-endian integers), so the zero-entry map is touched about once per 64 Ki quads on
// any data.  slot_salt(0) == 0 keeps slot 0 / quad 0 (the reference's zero-initialised table, chameleon.rs:41) valid from
// the start.
// The multiplier only decides WHICH quads pay the zero-entry path (one stored value per slot: about one in 64 Ki distinct quads on any data).
// Round 3's 0x9e5b happened to hit one word of the 4096-word benchmark text — 54 look-ups per 4 MiB chunk, each a late arrival for the chain;
// 0xb5ad is the first of sixteen candidates that hits none on four text samples (tools/salt_candidates.py).  Internal to the LDS table: the
// streams do not depend on it.
#ifndef DENSITY_HIP_SALT_MUL
#define DENSITY_HIP_SALT_MUL 0xb5adu
#endif
constexpr uint32_t kSaltMul = DENSITY_HIP_SALT_MUL;
__device__ __forceinline__ uint32_t slot_salt(uint32_t h) { return __umul24(h, kSaltMul) & 0xffffu; }   // 24-bit multiply: full rate (tests/datagen.py::salted_zero_quads mirrors it)
__device__ __forceinline__ uint32_t stored_entry(uint32_t q, uint32_t P) { return ((P & 0xfffeu) | (q >> 31)) ^ slot_salt(P >> 16); }
__device__ __forceinline__ uint32_t entry_to_quad(uint32_t h, uint32_t stored) {
    const uint32_t e = stored ^ slot_salt(h);
    const uint32_t Pfull = (h << 16) | (e & 0xfffeu);
    // ((Pfull >> 1) * inv) mod 2^31 == ((Pfull * inv) mod 2^32) >> 1 (Pfull is even), and bit 31 of the quad is bit 0 of the entry: one funnel
    // shift of {e, Pfull * inv} by one position does both
    return __builtin_amdgcn_alignbit(e, Pfull * kHalfMulInv, 1);
}

// the four-instruction dictionary step described in the file header; the caller masks inactive lanes with exec
__device__ __forceinline__ void dict_step(uint32_t addr, uint32_t lane, uint32_t e, uint32_t& old, uint32_t& w) {
    asm volatile(
        "ds_read_u16 %0, %2\n\t"
        "ds_write_b16 %2, %3\n\t"
        "ds_read_u16 %1, %2\n\t"
        "ds_write_b16 %2, %4\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(old), "=&v"(w)
        : "v"(addr), "v"(lane), "v"(e)
        : "memory");
}
__device__ __forceinline__ void dict_probe(uint32_t addr, uint32_t lane, uint32_t& old, uint32_t& w) {
    asm volatile(
        "ds_read_u16 %0, %2\n\t"
        "ds_write_b16 %2, %3\n\t"
        "ds_read_u16 %1, %2\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(old), "=&v"(w)
        : "v"(addr), "v"(lane)
        : "memory");
}
__device__ __forceinline__ void dict_store(uint32_t addr, uint32_t v) {
    asm volatile("ds_write_b16 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// Ordered exchange: every lane swaps its 16-bit entry into its slot and gets back what the slot held at ITS turn.  gfx950
// services the lanes of one LDS atomic in ascending lane order, so one instruction performs 64 sequential dictionary
// steps (chameleon.rs:88-100) including every same-slot dependency inside the block.  Two entries share a dword, hence
// the masked form ds_mskor_rtn_b32: mem = (mem & ~mask) | val.  Issue only; pair with lds_wait_keep / lds_wait_all.
__device__ __forceinline__ void dict_xchg_issue(uint32_t dword_addr, uint32_t mask, uint32_t val, uint32_t& ret) {
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3" : "=v"(ret) : "v"(dword_addr), "v"(mask), "v"(val) : "memory");
}
// wait until at most N LDS operations younger than the one producing `r` are outstanding (LDS returns in order)
template <int N>
__device__ __forceinline__ void lds_wait_keep(uint32_t& r) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(r) : "n"(N) : "memory"); }
__device__ __forceinline__ void lds_wait_keep_n(uint32_t& r, uint32_t n) {   // n is a compile-time constant after unrolling
    switch (n) {
        case 0: lds_wait_keep<0>(r); break;
        case 1: lds_wait_keep<1>(r); break;
        case 2: lds_wait_keep<2>(r); break;
        case 3: lds_wait_keep<3>(r); break;
        case 4: lds_wait_keep<4>(r); break;
        case 5: lds_wait_keep<5>(r); break;
        case 6: lds_wait_keep<6>(r); break;
        case 7: lds_wait_keep<7>(r); break;
        case 8: lds_wait_keep<8>(r); break;
        case 9: lds_wait_keep<9>(r); break;
        case 10: lds_wait_keep<10>(r); break;
        case 11: lds_wait_keep<11>(r); break;
        case 12: lds_wait_keep<12>(r); break;
        case 13: lds_wait_keep<13>(r); break;
        case 14: lds_wait_keep<14>(r); break;
        default: lds_wait_keep<15>(r); break;
    }
}
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t zmap_test_and_set(uint32_t zbase, uint32_t h) {
    uint32_t r;
    asm volatile("ds_or_rtn_b32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(zbase + (h >> 5) * 4u), "v"(1u << (h & 31u)) : "memory");
    return (r >> (h & 31u)) & 1u;
}
__device__ __forceinline__ uint32_t zmap_test(uint32_t zbase, uint32_t h) {
    uint32_t r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r) : "v"(zbase + (h >> 5) * 4u) : "memory");
    return (r >> (h & 31u)) & 1u;
}
__device__ __forceinline__ void zmap_set(uint32_t zbase, uint32_t h) {
    asm volatile("ds_or_b32 %0, %1" ::"v"(zbase + (h >> 5) * 4u), "v"(1u << (h & 31u)) : "memory");
}
// Where the zero-entry map lives: LDS for the one-wavefront kernels and the pipelined encoder, global memory (per chunk, 8 KiB,
// L2-resident, touched about once per 64 Ki quads) for the pipelined decoder, whose LDS is spent on rings.  Only one wavefront
// of a work-group ever touches it, in stream order.
struct ZmapLds {
    uint32_t base;
    __device__ __forceinline__ uint32_t test(uint32_t h) const { return zmap_test(base, h); }
    __device__ __forceinline__ void set(uint32_t h) const { zmap_set(base, h); }
    __device__ __forceinline__ uint32_t test_and_set(uint32_t h) const { return zmap_test_and_set(base, h); }
    __device__ __forceinline__ void clear(uint32_t h) const { asm volatile("ds_and_b32 %0, %1" ::"v"(base + (h >> 5) * 4u), "v"(~(1u << (h & 31u))) : "memory"); }
};
struct ZmapGlobal {
    uint32_t* words;
    __device__ __forceinline__ uint32_t test(uint32_t h) const {
        return (__hip_atomic_load(words + (h >> 5), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (h & 31u)) & 1u;   // bypasses the (non-coherent) L1
    }
    __device__ __forceinline__ void set(uint32_t h) const {
        const uint32_t old = atomicOr(words + (h >> 5), 1u << (h & 31u));
        asm volatile("" ::"v"(old));                          // returned value consumed: the update is in L2 before the next test
    }
    __device__ __forceinline__ uint32_t test_and_set(uint32_t h) const { return (atomicOr(words + (h >> 5), 1u << (h & 31u)) >> (h & 31u)) & 1u; }
};

// For every lane that shares its slot with other lanes of this block: the entry written by the nearest earlier lane
// of its group that is in `writers` (encode: every lane writes; decode: only PLAIN lanes do).
__device__ __forceinline__ void resolve_groups(bool active, uint32_t lane, uint32_t w, uint32_t e, uint64_t writers,
                                               bool& has_pred, uint32_t& pred_e) {
    has_pred = false;
    pred_e = 0;
    uint64_t todo = ballot64(active && w != lane);      // every multi-lane group has >= 1 lane in here
    const uint64_t lt = (1ull << lane) - 1ull;
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const uint32_t wsel = (uint32_t)__builtin_amdgcn_readlane((int)w, leader);
        const bool mine = active && (w == wsel);
        const uint64_t members = ballot64(mine);
        const uint64_t below = members & writers & lt;
        const uint32_t src = below ? (63u - (uint32_t)__builtin_clzll(below)) : lane;
        const uint32_t pe = bperm(src, e);
        if (mine && below) { has_pred = true; pred_e = pe; }
        todo &= ~members;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// decode: Codec::decode (codec/codec.rs:82-126) with Chameleon::decode_unit / decode_partial_unit
// (chameleon.rs:56-68,105-135).  The fast loop and the tail loop of the reference differ only in bounds checks; one
// vectorised stop test per record reproduces both (a record that is followed by >= 264 bytes can never trip it).
// ---------------------------------------------------------------------------------------------------------------
// In-order record loop shared by the one-wavefront kernel (whole stream) and the pipelined kernel (ragged end of the
// stream).  Runs on ONE wavefront from the state (ipos, opos, guard) until the input is exhausted; returns false where the
// reference would panic (truncated stream, output too small).
template <typename Zmap>
__device__ __forceinline__ bool decode_in_order(const uint8_t* __restrict__ src, uint64_t elen, uint8_t* __restrict__ dst, uint64_t cap,
                                                Guard& guard, uint64_t& ipos, uint64_t& opos, uint32_t tbl, Zmap zmap, uint32_t lane,
                                                bool mark_slot0 = false) {
    // mark_slot0 (last-writers passes of the segmented stream decode): a PLAIN zero quad written to slot 0 marks the slot as written
    while (ipos < elen) {
        const uint64_t rem = elen - ipos;
        const uint8_t* rec = src + ipos;
        uint8_t* o = dst + opos;
        if (guard.block_is_copy()) {                              // codec.rs:89-91,103-110
            const uint32_t take = rem > kBlock ? kBlock : (uint32_t)rem;
            if (opos + take > cap) return false;
            if (lane < (take >> 2)) st32u(o + 4u * lane, ld32u(rec + 4u * lane));
            if (lane < (take & 3u)) o[(take & ~3u) + lane] = rec[(take & ~3u) + lane];
            ipos += take;
            opos += take;
            if (rem <= kBlock) break;                             // codec.rs:107-109: no decay after the last raw block
            guard.decay();
            continue;
        }
        if (rem < kSig) return false;                             // reference: read_u64_le panics (read_buffer.rs:22)
        const uint64_t sig = (uint64_t)ld32u(rec) | ((uint64_t)ld32u(rec + 4) << 32);
        const bool hit = (sig >> lane) & 1ull;
        const uint32_t off = kSig + 4u * lane - 2u * mbcnt64(sig);
        const uint32_t rem32 = rem > 4096 ? 4096u : (uint32_t)rem;
        const int32_t left = (int32_t)rem32 - (int32_t)off;
        // chameleon.rs:121-126: PLAIN with < 4 bytes left ends the stream (copying 1..3 raw bytes); a MAP item with
        // < 2 bytes left is a truncated stream (reference panics)
        const bool stop = hit ? (left < 2) : (left < 4);
        const uint64_t stopm = ballot64(stop);
        const uint32_t kstop = stopm ? (uint32_t)__builtin_ctzll(stopm) : 64u;
        if (kstop < 64 && ((sig >> kstop) & 1ull)) return false;
        const bool active = lane < kstop;
        const uint32_t tailb = kstop < 64 ? (uint32_t)((int32_t)rem32 - __builtin_amdgcn_readlane((int)off, (int)kstop)) : 0u;
        if (opos + 4ull * kstop + tailb > cap) return false;

        uint32_t q = 0, h = 0, e = 0;
        if (active) {
            if (hit) { h = ld16u(rec + off); }
            else { q = ld32u(rec + off); const uint32_t P = q * kHashMul; h = P >> 16; e = stored_entry(q, P); }
        }
        uint32_t old = 0, w = lane;
        if (active) dict_probe(tbl + 2u * h, lane, old, w);
        const uint64_t plain = ballot64(active && !hit);
        bool has_pred;
        uint32_t pred_e;
        resolve_groups(active, lane, w, e, plain, has_pred, pred_e);
        const uint32_t eff = has_pred ? pred_e : old;             // what the slot holds when this lane's turn comes

        bool empty = false;                                       // MAP on a never-written slot yields the zero quad
        const bool hsusp = active && hit && !has_pred && old == 0 && h != 0;
        const bool psusp = active && !hit && e == 0 && (h != 0 || mark_slot0);
        if (ballot64(hsusp || psusp)) {
            if (hsusp) empty = !zmap.test(h);
            if (psusp) zmap.set(h);
        }
        if (active) {
            if (hit) q = empty ? 0u : entry_to_quad(h, eff);       // chameleon.rs:64-68: quad = chunk_map[hash]
            dict_store(tbl + 2u * h, hit ? eff : e);              // chameleon.rs:56-61: PLAIN stores, MAP leaves as is
            st32u(o + 4u * lane, q);
        }
        if (kstop < 64) {                                         // end of data inside this record
            if (lane < tailb) o[4u * kstop + lane] = rec[__builtin_amdgcn_readlane((int)off, (int)kstop) + lane];
            opos += 4ull * kstop + tailb;
            ipos = elen;
            break;
        }
        const uint32_t rec_len = kSig + kBlock - 2u * (uint32_t)__builtin_popcountll(sig);
        guard.update(rec_len >= kBlock);                          // codec.rs:98,122
        ipos += rec_len;
        opos += kBlock;
    }
    return true;
}


// zero-entry map, stream order: the lanes holding a zero entry claim their slots one at a time (rare: about one quad in 64 Ki)
template <typename Zmap>
__device__ __forceinline__ uint32_t zmap_claim_in_order(const Zmap& zmap, bool susp, uint32_t h, uint32_t lane) {
    uint32_t zbit = 1;
    uint64_t m = ballot64(susp);
    while (m) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        if (lane == l) zbit = zmap.test_and_set(h);
    }
    return zbit;
}

// The ragged last block of a stream (< 256 bytes: codec.rs:51-63 — whole quads, then 1..3 raw bytes without flag bits), encoded
// by ONE wavefront with the plain-probe dictionary step, from the FSM state and output position the full blocks left behind.
// Returns the stream length.
template <typename Zmap>
__device__ __forceinline__ uint64_t encode_ragged_block(const uint8_t* __restrict__ src, uint64_t len, uint32_t nfull, uint8_t* __restrict__ dst,
                                                        uint64_t opos, Guard& guard, uint8_t* __restrict__ idx, uint32_t tbl, const Zmap& zmap, uint32_t lane) {
    const uint64_t boff = (uint64_t)nfull * kBlock;
    const uint32_t blen = (uint32_t)(len - boff);
    if (!blen) return opos;
    const uint32_t nq = blen >> 2, tail = blen & 3u;
    const bool active = lane < nq;
    const uint32_t q = active ? ld32u(src + boff + 4u * lane) : 0u;
    uint8_t* rec = dst + opos;
    if (guard.block_is_copy()) {                                  // codec.rs:35-37
        if (active) st32u(rec + 4u * lane, q);
        if (lane < tail) rec[4u * nq + lane] = src[boff + 4u * nq + lane];
        if (idx && lane == 0) idx[nfull] = (uint8_t)(kIdxCopy | kIdxRagged);
        return opos + blen;
    }
    const uint32_t P = q * kHashMul;
    const uint32_t h = P >> 16;
    const uint32_t e = stored_entry(q, P);
    uint32_t old = 0, w = lane;
    if (active) dict_step(tbl + 2u * h, lane, e, old, w);
    bool has_pred;
    uint32_t pred_e;
    resolve_groups(active, lane, w, e, ~0ull, has_pred, pred_e);
    const bool susp = active && e == 0 && h != 0;
    const uint32_t zbit = zmap_claim_in_order(zmap, susp, h, lane);
    const bool hit = active && (has_pred ? (pred_e == e) : (old == e && (!susp || zbit)));
    const uint64_t sig = ballot64(hit);
    const uint32_t nhit = (uint32_t)__builtin_popcountll(sig);
    const uint32_t off = kSig + 4u * lane - 2u * mbcnt64(sig);
    if (lane == 0) { st32u(rec, (uint32_t)sig); st32u(rec + 4, (uint32_t)(sig >> 32)); }          // codec.rs:24-26
    if (active) {
        if (hit) st16u(rec + off, h); else st32u(rec + off, q);
    }
    const uint32_t items_end = kSig + 4u * nq - 2u * nhit;
    if (lane < tail) rec[items_end + lane] = src[boff + 4u * nq + lane];                            // codec.rs:58-61
    if (idx && lane == 0) idx[nfull] = (uint8_t)kIdxRagged;
    return opos + items_end + tail;
}
}  // namespace

}  // namespace density
