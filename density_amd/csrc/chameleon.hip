// chameleon.hip — Chameleon encode/decode kernels for gfx950 (MI355X).
//
// One 64-lane wavefront owns one chunk (= one independent reference stream) and walks it block by block:
// a Chameleon block is 64 quads with 1 flag bit each (chameleon.rs:138-146), so one block == one wavefront pass,
// the 64-bit signature == __ballot(hit) (first quad in bit 0, io/write_signature.rs:14-17) and a lane's output
// offset inside the record is a pair of mbcnt's.
//
// Dictionary.  The reference keeps 64 Ki x u32 = 256 KiB per stream (chameleon.rs:30-43); that does not fit the
// 160 KiB LDS of a CU.  Because the multiplier 0x9D6EF916 is 2 x odd, the product P = quad * M (mod 2^32) is even and
// (P, quad >> 31) determines the quad; with the slot index h = P >> 16 known, the 16-bit entry
//        e = (P & 0xfffe) | (quad >> 31)
// identifies the quad exactly.  The table is therefore 64 Ki x u16 = 128 KiB of LDS, exact, not a lossy fingerprint.
// All 2^16 entry values are legal for every slot, so "slot never written" (the reference's zero-initialised word:
// it matches only the zero quad, and only in slot 0) needs a 17th state: a never-written slot is stored as 0, and the
// only entry it aliases, e == 0 in a slot h != 0, is disambiguated by a 64 Ki-bit "this slot was written with e == 0"
// map (8 KiB) that is touched only when such a quad actually occurs (never in text: it needs two zero low bytes).
//
// Sequential semantics inside a block.  Lane i must observe the dictionary writes of lanes j < i of the same block.
// gfx950 LDS services the lanes of one ds instruction in ascending lane order (probes/lds_order.hip; re-checked by
// density_hip_selftest at start-up), so:  old = table[h];  table[h] = lane;  w = table[h];  table[h] = e;
// issued back to back gives every lane the pre-block value `old`, tells it the last lane `w` sharing its slot, and
// leaves the table in the state the sequential reference would (last writer wins).  Lanes whose slot is shared
// (w != lane somewhere in the group) take the value of the nearest earlier lane of the group instead of `old`;
// groups are enumerated with wave-level ballots.
#include <cstdio>
#include <cstdlib>

#include "chameleon_dev.hpp"
#include "kernels.hpp"

namespace density {



// ---------------------------------------------------------------------------------------------------------------
// encode: Codec::encode + encode_block (codec/codec.rs:34-80) with Chameleon::encode_quad (chameleon.rs:88-100)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void chameleon_encode_chunks(const uint8_t* __restrict__ in, uint64_t total,
                                                              uint64_t chunk_bytes, uint8_t* __restrict__ out,
                                                              uint64_t out_stride, uint64_t* __restrict__ sizes,
                                                              uint8_t* __restrict__ index) {
    const uint32_t lane = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const uint8_t* src = in + chunk * chunk_bytes;
    const uint64_t len = (total - chunk * chunk_bytes) < chunk_bytes ? (total - chunk * chunk_bytes) : chunk_bytes;
    uint8_t* dst = out + chunk * out_stride;
    const uint64_t nblk = (len + kBlock - 1) / kBlock;

    lds_clear(lane);
    const uint32_t tbl = lds_addr(smem);
    const uint32_t zmap = tbl + kTableBytes;

    Guard guard;
    uint64_t opos = 0;

    constexpr int PF = 4;   // blocks of input kept in flight per lane
    uint32_t qbuf[PF];
    auto load_quad = [&](uint64_t b) -> uint32_t {
        const uint64_t off = b * kBlock + 4u * lane;
        return (off + 4 <= len) ? ld32u(src + off) : 0u;
    };
#pragma unroll
    for (int u = 0; u < PF; ++u) qbuf[u] = load_quad(u);

    for (uint64_t b0 = 0; b0 < nblk; b0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const uint64_t b = b0 + u;
            if (b >= nblk) break;
            const uint32_t q = qbuf[u];
            qbuf[u] = load_quad(b + PF);

            const uint64_t boff = b * kBlock;
            const uint32_t blen = (len - boff) < kBlock ? (uint32_t)(len - boff) : kBlock;
            const uint32_t nq = blen >> 2, tail = blen & 3u;
            const bool active = lane < nq;
            uint8_t* rec = dst + opos;

            uint8_t* idx = index ? index + (chunk * chunk_bytes + boff) / kBlock : nullptr;    // block index entry (density_hip.h)
            if (guard.block_is_copy()) {                       // codec.rs:35-37: raw block, dictionary untouched
                if (active) st32u(rec + 4u * lane, q);
                if (lane < tail) rec[4u * nq + lane] = src[boff + 4u * nq + lane];
                if (idx && lane == 0) *idx = (uint8_t)(kIdxCopy | (blen < kBlock ? kIdxRagged : 0u));
                opos += blen;
                guard.decay();
                continue;
            }

            const uint32_t P = q * kHashMul;
            const uint32_t h = P >> 16;
            const uint32_t e = stored_entry(q, P);
            uint32_t old = 0, w = lane;
            if (active) dict_step(tbl + 2u * h, lane, e, old, w);

            bool has_pred;
            uint32_t pred_e;
            resolve_groups(active, lane, w, e, ~0ull, has_pred, pred_e);

            // e == 0 outside slot 0 aliases "never written": consult / update the zero-entry map (rare)
            const bool susp = active && e == 0 && h != 0;
            uint32_t zbit = 1;
            if (ballot64(susp)) {
                if (susp) zbit = zmap_test_and_set(zmap, h);
            }
            const bool hit = active && (has_pred ? (pred_e == e) : (old == e && (!susp || zbit)));

            const uint64_t sig = ballot64(hit);                // chameleon.rs:96: MAP flag = 1, PLAIN = 0
            const uint32_t nhit = (uint32_t)__builtin_popcountll(sig);
            const uint32_t off = kSig + 4u * lane - 2u * mbcnt64(sig);
            if (lane == 0) { st32u(rec, (uint32_t)sig); st32u(rec + 4, (uint32_t)(sig >> 32)); }   // codec.rs:24-26
            if (active) {
                if (hit) st16u(rec + off, h); else st32u(rec + off, q);
            }
            const uint32_t items_end = kSig + 4u * nq - 2u * nhit;
            if (lane < tail) rec[items_end + lane] = src[boff + 4u * nq + lane];                   // codec.rs:58-61
            const uint32_t rec_len = items_end + tail;
            if (idx && lane == 0) *idx = (uint8_t)(blen < kBlock ? kIdxRagged : nhit);
            guard.update(rec_len >= kBlock);                   // codec.rs:68
            opos += rec_len;
        }
    }
    if (lane == 0) sizes[chunk] = opos;
}

// ---------------------------------------------------------------------------------------------------------------
// Pipelined encoder: one work-group of 16 waves per chunk, every role in its own loop, one s_barrier per round of 8 blocks.
//
//   waves 0, 12  "dictionary waves": the only waves that touch the table, taking turns (even rounds on wave 0, odd rounds on
//            wave 12; the step barrier separates their accesses), so the in-order LDS pipeline gives the sequential dictionary
//            semantics.  Per block: one ordered exchange with operands held in registers, one bit-op + compare (= the
//            signature), one popcount.  The active wave runs the copy-mode FSM (protection_state.rs) per ROUND: in a round
//            without an incompressible block (< 5 hits of 64) the FSM only advances its block counter; the exchanges of a
//            round are issued speculatively "no raw-copy block in this round", and a round in which the FSM does switch to
//            copy mode is rolled back from the first copied block and redone in order.  It leaves the FSM state in LDS for
//            the other wave; in its passive step it publishes the 8 signatures and the raw-copy mask of its last round and
//            fetches the operands of its next one.
//   wave 4   "loader": global->LDS DMA (global_load_lds_dwordx4, 1 KiB = 4 blocks per instruction) of round t + kAhead into the
//            input ring, counted vmcnt.
//   5 waves  "hash" (kHashFirstTbl): two rounds ahead, quads -> {slot address, salted entry} in the operand ring; blocks holding
//            a stored entry 0 outside slot 0 are flagged for the dictionary wave's careful path (zero-entry map).
//   8 waves  "emit" (kEmitBlockTbl): two rounds behind, (signature, quads) -> record bytes: a prefix sum of the 8 record
//            lengths, a pair of mbcnt's per lane, 2-/4-byte stores through an SGPR base, the block-index bytes.
//
// Buffers: input ring of kInRing rounds, operand ring and result ring of 2 rounds; the zero-entry map lives in global memory
// (workspace).  Only whole 256-byte blocks go through the pipeline; a ragged last block (codec.rs:51-63) is finished by the
// dictionary wave with the scalar-path code of chameleon_encode_chunks.
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t kRound = 8;                               // blocks per round
constexpr uint32_t kRoundBytes = kRound * kBlock;            // 2 KiB
constexpr uint32_t kInRing = 8, kResRing = 2;              // input ring (power of two): emitted, published, finished, hashed x2, kAhead - 3 rounds in flight
constexpr uint32_t kAhead = 5;                               // the loader issues round t + kAhead during step t
constexpr uint32_t kInBase = kTableBytes;                    // the pipelined kernels keep the zero-entry map in global memory (ZmapGlobal)
constexpr uint32_t kResBase = kInBase + kInRing * kRoundBytes;
constexpr uint32_t kResBytes = 128;                          // dwords 0..15 signatures, 16 copy mask
constexpr uint32_t kOpBase = kResBase + kResRing * kResBytes;    // operand ring: per block 64 x {byte address of the slot, entry << 16*half}
constexpr uint32_t kOpRec = 512, kOpRing = 2;
constexpr uint32_t kOpRoundBytes = kRound * kOpRec;
constexpr uint32_t kZeroFlagBase = kOpBase + kOpRing * kOpRoundBytes;   // 4 dwords: bit k of word r % 4 = block k of round r holds a zero entry
constexpr uint32_t kGuardBase = kZeroFlagBase + 16;                   // FSM state handed from one dictionary wave to the other: {penalty, start, prev, counter}
constexpr uint32_t kLdsBytesPipe = kGuardBase + 16;
constexpr uint32_t kEncAddr = 0x1fffcu;                      // operand dword 0 = 2 * slot: bits 2..16 the dword address, bit 1 the half
constexpr uint32_t kPipeWaves = 16;
// Roles by wave, one nibble per wave (8 = none).  A work-group's wave w runs on SIMD w % 4 and the four waves of a SIMD share
// its issue slots, so the dictionary wave (0) shares SIMD 0 only with the loader (4) and two single-block hash waves (8, 12);
// the eight emit waves and the double-block hash waves are spread over SIMDs 1-3.
constexpr uint32_t kDictWave = 0, kDictWaveB = 12, kLoadWave = 4;   // two dictionary waves: even rounds on wave 0, odd rounds on wave 12
constexpr uint64_t kHashFirstTbl = 0x7428688088888888ull;    // first block hashed by wave w (nibble w)
constexpr uint64_t kHashCountTbl = 0x1220100200000000ull;    // number of consecutive blocks hashed by wave w
constexpr uint64_t kEmitBlockTbl = 0x8888852874186308ull;    // block emitted by wave w
static_assert(kLdsBytesPipe <= 160u * 1024u, "LDS budget");
static_assert(kRound == 8, "register arrays, asm operand lists and the result record are written for 8 blocks per round");

// one LDS-DMA instruction: lane l copies 16 bytes from its global pointer to lds_dst + 16*l (lds_dst wave-uniform)
__device__ __forceinline__ void dma_1k(const uint8_t* gsrc, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// all of this wave's LDS traffic retired, then the work-group barrier; no vmcnt: stores and DMA stay in flight
__device__ __forceinline__ void round_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Stores of the emit waves: wave-uniform base in SGPRs + 32-bit lane offset.  Written as asm so that the compiler neither
// builds 64-bit lane addresses nor tracks the stores with vmcnt (it would drain them at every loop head).
__device__ __forceinline__ void gstore32(uint8_t* base, uint32_t off, uint32_t v) {
    asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void gstore16_hi(uint8_t* base, uint32_t off, uint32_t v) {   // stores bits 16..31 of v
    asm volatile("global_store_short_d16_hi %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}
__device__ __forceinline__ void gstore8(uint8_t* base, uint32_t off, uint32_t v) {
    asm volatile("global_store_byte %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
}

// run-time lane select (in-order path only)
__device__ __forceinline__ uint32_t wlane_dyn(uint32_t vec, uint32_t value, uint32_t lane_sel, uint32_t lane) { return lane == lane_sel ? value : vec; }
__device__ __forceinline__ uint32_t rlane(uint32_t vec, uint32_t lane_sel) { return (uint32_t)__builtin_amdgcn_readlane((int)vec, (int)lane_sel); }

// optional cycle accounting (DENSITY_HIP_PROF=1): work-group 0 reports, per wave, cycles spent working and cycles spent at barriers
template <bool ON>
struct WaveClock;
template <>
struct WaveClock<false> {                                     // production: compiles to nothing
    __device__ __forceinline__ explicit WaveClock(uint64_t*) {}
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void work_done() {}
    __device__ __forceinline__ void wait_done() {}
    __device__ __forceinline__ void flush(uint32_t, uint32_t) {}
    __device__ __forceinline__ void phase_start() {}
    __device__ __forceinline__ void phase(int) {}
    __device__ __forceinline__ void flush_phases(uint32_t) {}
    __device__ __forceinline__ void set_fine(bool) {}
    __device__ __forceinline__ void stat_post(uint32_t, uint32_t, uint32_t) {}
    __device__ __forceinline__ void stat_take(uint32_t, uint32_t, uint32_t) {}
    __device__ __forceinline__ void flush_stat(uint32_t, uint32_t) {}
};
template <>
struct WaveClock<true> {
    uint64_t* out; uint64_t work = 0, wait = 0, t0 = 0;
    uint64_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp = 0;       // phase split of one wave's work (dictionary wave): out[32 + k]
    uint32_t stat = 0, last = 0, hist = 0; uint64_t summax = 0;   // straggler statistics (wave 0): LDS scratch, per-lane counts
    __device__ __forceinline__ explicit WaveClock(uint64_t* o) : out(o) {}
    __device__ __forceinline__ void start() { if (out) t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void work_done() { if (out) { const uint64_t t = __builtin_readcyclecounter(); work += t - t0; last = (uint32_t)(t - t0); t0 = t; } }
    __device__ __forceinline__ void wait_done() { if (out) { const uint64_t t = __builtin_readcyclecounter(); wait += t - t0; t0 = t; } }
    __device__ __forceinline__ void flush(uint32_t wave, uint32_t lane) { if (out && lane == 0) { out[2 * wave] = work; out[2 * wave + 1] = wait; } }
    bool fine = false;                                        // per-phase timing costs ~100 cycles a call: only on request (DENSITY_HIP_DBG bit 5)
    __device__ __forceinline__ void phase_start() { if (out && fine) tp = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void phase(int k) { if (out && fine) { const uint64_t t = __builtin_readcyclecounter(); ph[k] += t - tp; tp = t; } }
    __device__ __forceinline__ void flush_phases(uint32_t lane) { if (out && lane == 0) for (int k = 0; k < 8; ++k) out[32 + k] = ph[k]; }
    __device__ __forceinline__ void set_fine(bool f) { fine = f; }
    // which wave was the last to reach the barrier, per step: every wave posts its work time before the barrier (stat_post),
    // wave 0 reads the 16 values after it (stat_take) and counts, per lane = wave, how often that wave was the slowest
    __device__ __forceinline__ void stat_post(uint32_t stat_base, uint32_t wave, uint32_t lane) {
        if (out && lane == 0) *reinterpret_cast<uint32_t*>(smem + stat_base + 4u * wave) = last;
    }
    __device__ __forceinline__ void stat_take(uint32_t stat_base, uint32_t wave, uint32_t lane) {
        if (out && wave == 0) {
            const uint32_t v = lane < 16 ? *reinterpret_cast<const uint32_t*>(smem + stat_base + 4u * lane) : 0u;
            uint32_t m = v;
            for (int d = 1; d < 16; d <<= 1) { const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((lane ^ d) & 63u) << 2), (int)m); m = o > m ? o : m; }
            hist += (v == m && lane < 16) ? 1u : 0u;
            summax += m;
        }
    }
    __device__ __forceinline__ void flush_stat(uint32_t wave, uint32_t lane) {
        if (out && wave == 0 && lane < 16) { out[48 + lane] = hist; if (lane == 0) out[47] = summax; }
    }
};

// per-block state of the dictionary wave between issue and finish
struct Issued {
    uint32_t d0, d1, ret;         // operands staged by the hash waves (2 * slot, entry << 16*half), dictionary answer
};

// zero-entry map, stream order: the lanes holding a zero entry claim their slots one at a time (rare: about one quad in 64 Ki)
__device__ __forceinline__ uint32_t zmap_claim(const ZmapGlobal& zmap, bool susp, uint32_t h, uint32_t lane) {
    uint32_t zbit = 1;
    uint64_t m = ballot64(susp);
    while (m) {
        const uint32_t l = (uint32_t)__builtin_ctzll(m);
        m &= m - 1;
        if (lane == l) zbit = zmap.test_and_set(h);
    }
    return zbit;
}

}  // namespace

template <bool kProf>
__global__ __launch_bounds__(kPipeWaves * 64) void chameleon_encode_chunks_pipe(const uint8_t* __restrict__ in, uint64_t total,
                                                                                uint64_t chunk_bytes, uint8_t* __restrict__ out,
                                                                                uint64_t out_stride, uint64_t* __restrict__ sizes,
                                                                                uint8_t* __restrict__ index, uint32_t* __restrict__ zmap_words,
                                                                                uint32_t dbg, uint64_t* __restrict__ prof) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint64_t chunk = blockIdx.x;
    WaveClock<kProf> clk(blockIdx.x == 0 ? prof : nullptr);
    clk.set_fine((dbg & 32u) != 0);
    const uint8_t* src = in + chunk * chunk_bytes;
    const uint64_t len = (total - chunk * chunk_bytes) < chunk_bytes ? (total - chunk * chunk_bytes) : chunk_bytes;
    uint8_t* dst = out + chunk * out_stride;
    uint8_t* idx = index ? index + chunk * (chunk_bytes / kBlock) : nullptr;     // this chunk's slice of the block index
    const uint32_t nfull = (uint32_t)(len / kBlock);           // whole blocks: these go through the pipeline (the launcher bounds len)
    const uint32_t nrounds = (nfull + kRound - 1) / kRound;
    const ZmapGlobal zmap{zmap_words + chunk * (kZmapBytes / 4)};
    const bool is_dict = wave == kDictWave || wave == kDictWaveB;
    const uint32_t par = wave == kDictWave ? 0u : 1u;          // this dictionary wave owns the rounds of this parity
    if (is_dict) __builtin_amdgcn_s_setprio(3);                // the dictionary waves are the critical path: first pick of issue slots

    {   // clear table, zero-entry flags and this chunk's zero-entry map
        uint4* p = reinterpret_cast<uint4*>(smem);
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint32_t i = threadIdx.x; i < kTableBytes / 16; i += kPipeWaves * 64) p[i] = z;
        for (uint32_t i = threadIdx.x; i < kZmapBytes / 16; i += kPipeWaves * 64) reinterpret_cast<uint4*>(zmap.words)[i] = z;
        if (threadIdx.x < 4) *reinterpret_cast<uint32_t*>(smem + kZeroFlagBase + 4u * threadIdx.x) = 0;
        if (threadIdx.x == 0) *reinterpret_cast<uint4*>(smem + kGuardBase) = make_uint4(0u, 1u, 0u, 0u);   // Guard{} (common.hpp)
        // The map is used through L2 atomics by this work-group only, so all that is needed is that these stores have reached
        // L2 (the L1 is write-through): vmcnt(0), then the barrier below.  __threadfence() here would write back and invalidate
        // the whole L2 of the XCD (buffer_wbl2 / buffer_inv, once per wave), ~0.1 ms per chunk.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const uint32_t lds0 = lds_addr(smem);
    const uint32_t tbl = lds0;
    round_barrier();

    // ---------------- loader: DMA of round r into ring slot r % kInRing; lanes past the last whole block stay idle ------
    auto issue_round = [&](uint32_t r) {
        if (r >= nrounds) return;
        const uint64_t base = (uint64_t)r * kRoundBytes;
#pragma unroll
        for (uint32_t j = 0; j < kRound / 4; ++j) {
            const uint64_t off = base + j * 1024u + 16u * lane;
            if (off + 16 <= (uint64_t)nfull * kBlock) dma_1k(src + off, lds0 + kInBase + (r & (kInRing - 1u)) * kRoundBytes + j * 1024u);
        }
    };
    // Loader invariant: before the barrier that ends step t, rounds <= t + 3 have landed (the hash waves read round t + 3 in
    // step t + 1) and rounds up to t + kAhead have been issued.  vmcnt retires in order and every round before the last one is
    // exactly kRound / 4 instructions, so the wait is exact away from the chunk's end.
    constexpr uint32_t kInFlight = (kAhead - 3) * (kRound / 4);
    if (wave == kLoadWave) {
#pragma unroll
        for (uint32_t r = 0; r < kAhead; ++r) issue_round(r);
        if (kAhead < nrounds && !(dbg & 1u)) wait_vm<kInFlight>(); else wait_vm<0>();      // rounds 0..2 landed
    }
    round_barrier();

    // ---------------- hash waves: round r into the operand ring (two rounds ahead of the dictionary wave, which fetches the
    // operands of round t+1 while the exchanges of round t run) --------
    const uint32_t hb0 = (uint32_t)(kHashFirstTbl >> (4u * wave)) & 15u, hbn = (uint32_t)(kHashCountTbl >> (4u * wave)) & 15u;
    const uint32_t eb = (uint32_t)(kEmitBlockTbl >> (4u * wave)) & 15u;
    auto hash_round = [&](uint32_t r) {
        if (r >= nrounds) return;
        const uint32_t qbase = kInBase + (r & (kInRing - 1u)) * kRoundBytes + 4u * lane;
        const uint32_t obase = kOpBase + (r & (kOpRing - 1u)) * kOpRoundBytes + 8u * lane;
        const uint32_t left = nfull - r * kRound;
        const uint32_t nb = left < kRound ? left : kRound;
#pragma unroll
        for (uint32_t i = 0; i < 2; ++i) {
            const uint32_t k = hb0 + i;
            if (i < hbn && k < nb) {
                const uint32_t q = *reinterpret_cast<const uint32_t*>(smem + qbase + kBlock * k);
                const uint32_t P = q * kHashMul;
                const uint32_t h = P >> 16;
                const uint32_t e = stored_entry(q, P);
                *reinterpret_cast<uint2*>(smem + obase + k * kOpRec) = make_uint2(h << 1, e << ((h & 1u) << 4));
                // a stored entry of 0 outside slot 0 aliases "never written": tell the dictionary wave to take the careful path
                if (ballot64(e == 0)) {
                    if (ballot64(e == 0 && h != 0) && lane == 0) atomicOr(reinterpret_cast<uint32_t*>(smem + kZeroFlagBase + 4u * (r & 3u)), 1u << k);
                }
            }
        }
    };
    if (hbn) { hash_round(0); hash_round(1); }
    round_barrier();

    uint64_t opos_run = 0;                                    // emit waves: output offset of the round being emitted (each tracks it)
    Guard guard;                                              // dictionary wave
    // the dictionary wave takes the operands of round 0 before the hash waves reuse that ring slot for round 2
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 nx0[kRound / 2];
    uint32_t zero0 = 0;
    if (wave == kDictWave) {
        const uint32_t a = lds0 + kOpBase + 8u * lane, z = lds0 + kZeroFlagBase;
        asm volatile(
            "ds_read2st64_b64 %0, %5 offset1:1\n\t"
            "ds_read2st64_b64 %1, %5 offset0:2 offset1:3\n\t"
            "ds_read2st64_b64 %2, %5 offset0:4 offset1:5\n\t"
            "ds_read2st64_b64 %3, %5 offset0:6 offset1:7\n\t"
            "ds_read_b32 %4, %6\n\t"
            "s_waitcnt lgkmcnt(0)"
            : "=&v"(nx0[0]), "=&v"(nx0[1]), "=&v"(nx0[2]), "=&v"(nx0[3]), "=&v"(zero0)
            : "v"(a), "v"(z)
            : "memory");
    }
    round_barrier();

    if (wave == kLoadWave) {
        // ---------------- loader ----------------
        for (uint32_t t = 0; t <= nrounds + 1; ++t) {
            clk.start();
            issue_round(t + kAhead);
            if (t + kAhead + 1 < nrounds && !(dbg & 1u)) wait_vm<kInFlight>(); else wait_vm<0>();
            clk.work_done();
            clk.stat_post(kLdsBytesPipe, wave, lane);
            round_barrier();
            clk.wait_done();
            clk.stat_take(kLdsBytesPipe, wave, lane);
        }
    } else if (is_dict) {
        // ---------------- dictionary waves: wave 0 runs the even rounds, wave 12 the odd ones.  While one of them works
        // through the exchanges of round t, the other publishes its results of round t-1 and fetches its operands of round t+1;
        // the FSM state travels through LDS (kGuardBase).  The barrier between steps orders their accesses to the table. ----
        // (the compiler lays this branch out behind the other roles' loops and carries their pending LDS accesses into it: clear
        // its scoreboard with a wait it can see, or it guards registers here with lgkmcnt waits that drain the exchanges)
        __builtin_amdgcn_s_waitcnt(0xc07f);
        Issued blk[kRound];
        u32x4 nx[kRound / 2];                                 // operands of the next round: {d0, d1} of blocks 2i, 2i+1
        uint32_t zero_blocks = rfl(zero0), zero_nxt = 0;
#pragma unroll
        for (uint32_t i = 0; i < kRound / 2; ++i) {
            blk[2 * i].d0 = nx0[i].x; blk[2 * i].d1 = nx0[i].y; blk[2 * i + 1].d0 = nx0[i].z; blk[2 * i + 1].d1 = nx0[i].w;
            blk[2 * i].ret = 0; blk[2 * i + 1].ret = 0;
        }
        // Operands (and zero-entry flags) of round r: five LDS reads, asynchronous — pair with ops_take().  The compiler believes an
        // asm's outputs are valid when the asm ends, so it is free to copy them elsewhere before the wait (it did: copies of
        // registers still in flight).  Both ends are therefore pinned to fixed registers the allocator never has a reason to move,
        // and the copies into the working registers happen inside the same asm as the wait.
        auto load_ops = [&](uint32_t r) {
            const uint32_t a = lds0 + kOpBase + (r & (kOpRing - 1u)) * kOpRoundBytes + 8u * lane;
            const uint32_t z = lds0 + kZeroFlagBase + 4u * (r & 3u);
            asm volatile(
                "ds_read2st64_b64 %0, %5 offset1:1\n\t"
                "ds_read2st64_b64 %1, %5 offset0:2 offset1:3\n\t"
                "ds_read2st64_b64 %2, %5 offset0:4 offset1:5\n\t"
                "ds_read2st64_b64 %3, %5 offset0:6 offset1:7\n\t"
                "ds_read_b32 %4, %6"
                : "={v[100:103]}"(nx[0]), "={v[104:107]}"(nx[1]), "={v[108:111]}"(nx[2]), "={v[112:115]}"(nx[3]), "={v116}"(zero_nxt)
                : "v"(a), "v"(z)
                : "memory");
        };
        auto ops_take = [&]() {                               // all LDS traffic of this wave retired; the prefetched operands become current
            uint32_t zf;
            asm volatile(
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_mov_b32 %0, v100\n\tv_mov_b32 %1, v101\n\tv_mov_b32 %2, v102\n\tv_mov_b32 %3, v103\n\t"
                "v_mov_b32 %4, v104\n\tv_mov_b32 %5, v105\n\tv_mov_b32 %6, v106\n\tv_mov_b32 %7, v107\n\t"
                "v_mov_b32 %8, v108\n\tv_mov_b32 %9, v109\n\tv_mov_b32 %10, v110\n\tv_mov_b32 %11, v111\n\t"
                "v_mov_b32 %12, v112\n\tv_mov_b32 %13, v113\n\tv_mov_b32 %14, v114\n\tv_mov_b32 %15, v115\n\t"
                "v_mov_b32 %16, v116"
                : "=&v"(blk[0].d0), "=&v"(blk[0].d1), "=&v"(blk[1].d0), "=&v"(blk[1].d1), "=&v"(blk[2].d0), "=&v"(blk[2].d1), "=&v"(blk[3].d0), "=&v"(blk[3].d1),
                  "=&v"(blk[4].d0), "=&v"(blk[4].d1), "=&v"(blk[5].d0), "=&v"(blk[5].d1), "=&v"(blk[6].d0), "=&v"(blk[6].d1), "=&v"(blk[7].d0), "=&v"(blk[7].d1), "=&v"(zf)
                : "{v[100:103]}"(nx[0]), "{v[104:107]}"(nx[1]), "{v[108:111]}"(nx[2]), "{v[112:115]}"(nx[3]), "{v116}"(zero_nxt)
                : "memory");
#pragma unroll
            for (uint32_t j = 0; j < kRound; ++j) blk[j].ret = 0;
            zero_blocks = rfl(zf);
        };
        uint32_t copy_mask = 0;                               // results of this wave's last round, published one step later
        uint64_t sig[kRound];
#pragma unroll
        for (uint32_t j = 0; j < kRound; ++j) sig[j] = 0;
        bool unpublished = false;
        // results for the emit waves: 8 signatures + the copy mask of round r, written by lane 0 (the values are wave-uniform)
        auto publish = [&](uint32_t r) {
            if (lane == 0) {
                const u32x4 s01 = {(uint32_t)sig[0], (uint32_t)(sig[0] >> 32), (uint32_t)sig[1], (uint32_t)(sig[1] >> 32)};
                const u32x4 s23 = {(uint32_t)sig[2], (uint32_t)(sig[2] >> 32), (uint32_t)sig[3], (uint32_t)(sig[3] >> 32)};
                const u32x4 s45 = {(uint32_t)sig[4], (uint32_t)(sig[4] >> 32), (uint32_t)sig[5], (uint32_t)(sig[5] >> 32)};
                const u32x4 s67 = {(uint32_t)sig[6], (uint32_t)(sig[6] >> 32), (uint32_t)sig[7], (uint32_t)(sig[7] >> 32)};
                asm volatile(
                    "ds_write_b128 %0, %1\n\t"
                    "ds_write_b128 %0, %2 offset:16\n\t"
                    "ds_write_b128 %0, %3 offset:32\n\t"
                    "ds_write_b128 %0, %4 offset:48\n\t"
                    "ds_write_b32 %0, %5 offset:64"
                    ::"v"(lds0 + kResBase + (r & (kResRing - 1u)) * kResBytes), "v"(s01), "v"(s23), "v"(s45), "v"(s67), "v"(copy_mask) : "memory");
            }
        };
        for (uint32_t t = 0; t <= nrounds + 1; ++t) {
            clk.start();
            if ((t & 1u) != par) {
                // passive step: hand round t-1 to the emit waves, take the operands of round t+1 (hashed during step t-1)
                // (the reads first: they are back by the time the results are written out)
                if (t + 1 < nrounds) load_ops(t + 1);
                if (unpublished) { publish(t - 1); unpublished = false; }
                if (t + 1 < nrounds) ops_take();
            } else if (t < nrounds) {
                clk.phase_start();
                const uint32_t left = nfull - t * kRound;
                const uint32_t nb = left < kRound ? left : kRound;
                copy_mask = 0;
                {                                                 // the FSM as the other wave left it after round t-1 (Guard{} before round 0)
                    const uint4 g = *reinterpret_cast<const uint4*>(smem + kGuardBase);
                    guard.penalty = rfl(g.x); guard.start = rfl(g.y); guard.prev = rfl(g.z); guard.counter = rfl(g.w);
                }

                auto issue = [&](Issued& b) {
                    const uint32_t sh = (b.d0 << 3) & 31u;       // (d0 & 2) << 3: the half of the dword the slot lives in
                    dict_xchg_issue(tbl + (b.d0 & kEncAddr), 0xffffu << sh, b.d1, b.ret);
                };
                // signature of a block from the dictionary answers, including the zero-entry disambiguation (rare path)
                auto signature = [&](const Issued& b) -> uint64_t {
                    const uint32_t sh = (b.d0 & 2u) << 3;
                    const uint32_t e = (b.d1 >> sh) & 0xffffu, h = b.d0 >> 1;
                    const uint32_t old = (b.ret >> sh) & 0xffffu;
                    const bool susp = e == 0 && h != 0;
                    const uint32_t zbit = zmap_claim(zmap, susp, h, lane);
                    return ballot64(old == e && (!susp || zbit));
                };
                // undo a speculatively applied block: the lowest lane of a slot holds the pre-block entry, so the lanes
                // write their answers back in descending order
                auto rollback = [&](const Issued& b) {
                    const uint32_t sh = (b.d0 & 2u) << 3;
                    const uint32_t a16 = tbl + b.d0;
                    const uint32_t prev = (b.ret >> sh) & 0xffffu;
#pragma nounroll
                    for (int l = 63; l >= 0; --l) {
                        if (lane == (uint32_t)l) dict_store(a16, prev);
                    }
                };

                clk.phase(1);

                // what the other wave needs to know: the FSM state after this round
                auto leave_guard = [&]() {
                    if (lane == 0) *reinterpret_cast<uint4*>(smem + kGuardBase) = make_uint4(guard.penalty, guard.start, guard.prev, guard.counter);
                };
                // eight answers -> eight signatures and the smallest MAP count (plain rounds: a hit is simply "answer == entry")
                uint32_t min_hits = 64;
                auto take_signatures = [&](uint32_t younger = 0) {               // `younger`: LDS operations issued after the exchanges
#pragma unroll
                    for (uint32_t j = 0; j < kRound; ++j) {
                        lds_wait_keep_n(blk[j].ret, kRound - 1 - j + younger);    // later exchanges stay in flight
                        const uint32_t sh = (blk[j].d0 << 3) & 31u;
                        sig[j] = ballot64(((blk[j].ret ^ blk[j].d1) & (0xffffu << sh)) == 0);
                        const uint32_t nh = (uint32_t)__builtin_popcountll(sig[j]);
                        min_hits = nh < min_hits ? nh : min_hits;
                    }
                };
                // Everything that is not the common round.  `issued_all`: the eight exchanges of the round are done (speculatively: "no
                // raw-copy block in this round") and, in a plain round, the signatures are taken.
                auto slow_round = [&](bool issued_all, bool plain_round) {
                    uint32_t k = 0;
                    bool pending_copy = false;                    // guard already advanced for block k and said "copy"
                    if (issued_all) {
                        // walk the FSM block by block; stop at the first block it turns into a raw copy.  With zero-entry quads
                        // in the round the signature itself updates the zero-entry map, so it is taken only for blocks the FSM
                        // has admitted.
#pragma unroll
                        for (uint32_t j = 0; j < kRound; ++j) {
                            if (k == j) {
                                if (guard.block_is_copy()) {
                                    pending_copy = true;
                                } else {
                                    if (!plain_round) sig[j] = signature(blk[j]);
                                    guard.update((uint32_t)__builtin_popcountll(sig[j]) <= 4);
                                    k = j + 1;
                                }
                            }
                        }
                        if (pending_copy) {
                            lds_wait_all();
#pragma unroll
                            for (int j = (int)kRound - 1; j >= 0; --j) {
                                if ((uint32_t)j >= k) { rollback(blk[j]); sig[j] = 0; }
                            }
                            lds_wait_all();
                        }
                    }
                    if (k < nb) {                                 // in-order path: copy runs, the blocks after a mis-speculation, short rounds
#pragma unroll
                        for (uint32_t j = 0; j < kRound; ++j) {
                            if (j >= k && j < nb) {
                                const bool cp = pending_copy ? true : guard.block_is_copy();
                                pending_copy = false;
                                if (cp) {                         // codec.rs:35-37
                                    copy_mask |= 1u << j;
                                    guard.decay();
                                } else {
                                    issue(blk[j]);
                                    lds_wait_all();
                                    sig[j] = signature(blk[j]);
                                    guard.update((uint32_t)__builtin_popcountll(sig[j]) <= 4);   // codec.rs:68
                                }
                            }
                        }
                    }
                };
                if (__builtin_expect((guard.penalty | guard.prev | zero_blocks) == 0 && left >= kRound, 1)) {
                    // The common round, one straight block: a whole round, the FSM calm, no quad that packs to entry 0.  Computing
                    // the signatures has no side effect, so all eight are taken before the FSM is consulted.
#pragma unroll
                    for (uint32_t j = 0; j < kRound; ++j) issue(blk[j]);
                    clk.phase(2);
                    // What the FSM will look like if no record of the round is incompressible (codec.rs:68: 8 + 256 - 2*hits >= 256):
                    // it only counts blocks (protection_state.rs:19-27), and one of 8 consecutive counters is a multiple of 16 iff
                    // c == 0 or c > 8.  That state goes to LDS right away, behind the exchanges, so the end of the step does not
                    // wait for the write; any other outcome overwrites it.
                    const Guard before = guard;
                    const uint32_t c = guard.counter & 15u;
                    guard.start >>= (uint32_t)((c - 1u) >= 8u) & (uint32_t)(guard.start > 1u);
                    guard.counter += kRound;
                    if (lane == 0) {
                        const u32x4 gv = {guard.penalty, guard.start, guard.prev, guard.counter};
                        asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + kGuardBase), "v"(gv) : "memory");
                    }
                    take_signatures(1);
                    clk.phase(3);
                    if (__builtin_expect(min_hits <= 4, 0)) {
                        guard = before;
                        slow_round(true, true);                   // the FSM has to look at the blocks one by one
                        leave_guard();
                    }
                } else {
#pragma unroll
                    for (uint32_t j = 0; j < kRound; ++j) sig[j] = 0;     // raw-copy blocks and blocks past a short round's end keep 0
                    const bool spec = nb == kRound && guard.penalty == 0;
                    const bool plain_round = zero_blocks == 0;
                    if (spec) {
#pragma unroll
                        for (uint32_t j = 0; j < kRound; ++j) issue(blk[j]);
                        if (plain_round) take_signatures(); else lds_wait_all();
                    }
                    slow_round(spec, plain_round);
                    leave_guard();
                }
                clk.phase(4);
                unpublished = true;                               // the results go out in the next (passive) step
                clk.phase(5);
            }
            clk.work_done();
            clk.stat_post(kLdsBytesPipe, wave, lane);
            round_barrier();
            clk.wait_done();
            clk.stat_take(kLdsBytesPipe, wave, lane);
        }
    } else {
        // ---------------- emit waves: round t-1; hash waves: round t+1 ----------------
        const uint32_t c_off = kSig + 4u * lane;              // item offset of this lane in a record without MAP flags
        const uint32_t sl = lane & 7u;
        for (uint32_t t = 0; t <= nrounds + 1; ++t) {
            clk.start();
            // round t-1's zero-entry flags were read by the dictionary wave during the previous step; the word is next used for round t+3
            if (hb0 == 0 && t >= 1 && lane == 0) *reinterpret_cast<uint32_t*>(smem + kZeroFlagBase + 4u * ((t - 1) & 3u)) = 0;
            if (eb < kRound && t >= 2 && !(dbg & 2u)) {
                const uint32_t r = t - 2;
                const uint32_t rbase = kResBase + (r & (kResRing - 1u)) * kResBytes;
                const uint32_t left = nfull - r * kRound;
                // lanes 0..7 (and their images): signature and record length of block `lane & 7`; prefix over the round = record offsets
                const uint2 sg = *reinterpret_cast<const uint2*>(smem + rbase + 8u * sl);
                const uint32_t cmask = rfl(*reinterpret_cast<const uint32_t*>(smem + rbase + 64));
                const uint32_t q = *reinterpret_cast<const uint32_t*>(smem + kInBase + (r & (kInRing - 1u)) * kRoundBytes + kBlock * eb + 4u * lane);
                const uint32_t myhits = (uint32_t)(__builtin_popcount(sg.x) + __builtin_popcount(sg.y));
                uint32_t mylen = kSig + kBlock - 2u * myhits;
                if (cmask | (uint32_t)(left < kRound)) {         // raw-copy blocks in the round, or the chunk's last (short) round
                    mylen = ((cmask >> sl) & 1u) ? kBlock : mylen;
                    mylen = sl < left ? mylen : 0u;
                }
                uint32_t incl = mylen;
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
                const uint32_t before = rlane(incl - mylen, eb), round_len = rlane(incl, 7);
                const uint32_t slo = rlane(sg.x, eb), shi = rlane(sg.y, eb);
                uint8_t* recp = dst + (opos_run + before);
                if (eb == 0 && idx && lane < kRound && lane < left)
                    gstore8(idx + (uint64_t)r * kRound, lane, ((cmask >> sl) & 1u) ? kIdxCopy : myhits);
                if (eb < left) {
                    if ((cmask >> eb) & 1u) {
                        gstore32(recp, 4u * lane, q);
                    } else {
                        const uint64_t sgk = (uint64_t)slo | ((uint64_t)shi << 32);
                        const uint32_t off = c_off - 2u * mbcnt64(sgk);
                        if (lane < 2) gstore32(recp, 4u * lane, lane ? shi : slo);
                        if ((sgk >> lane) & 1ull) gstore16_hi(recp, off, q * kHashMul); else gstore32(recp, off, q);
                    }
                }
                opos_run += round_len;
            }
            if (hbn) hash_round(t + 2);
            clk.work_done();
            clk.stat_post(kLdsBytesPipe, wave, lane);
            round_barrier();
            clk.wait_done();
            clk.stat_take(kLdsBytesPipe, wave, lane);
        }
    }
    clk.flush(wave, lane);
    clk.flush_stat(wave, lane);
    if (wave == kDictWave) clk.flush_phases(lane);

    // hand the stream length so far to the dictionary wave, which finishes a ragged last block with the scalar-path code
    if (eb == 0 && lane == 0) *reinterpret_cast<uint64_t*>(smem + kResBase) = opos_run;
    round_barrier();
    if (wave == kDictWave) {
        {
            const uint4 g = *reinterpret_cast<const uint4*>(smem + kGuardBase);       // as the last round's wave left it
            guard.penalty = rfl(g.x); guard.start = rfl(g.y); guard.prev = rfl(g.z); guard.counter = rfl(g.w);
        }
        uint64_t opos = *reinterpret_cast<const uint64_t*>(smem + kResBase);
        const uint64_t boff = (uint64_t)nfull * kBlock;
        const uint32_t blen = (uint32_t)(len - boff);
        if (blen) {
            const uint32_t nq = blen >> 2, tail = blen & 3u;
            const bool active = lane < nq;
            const uint32_t q = active ? ld32u(src + boff + 4u * lane) : 0u;
            uint8_t* rec = dst + opos;
            if (guard.block_is_copy()) {
                if (active) st32u(rec + 4u * lane, q);
                if (lane < tail) rec[4u * nq + lane] = src[boff + 4u * nq + lane];
                if (idx && lane == 0) idx[nfull] = (uint8_t)(kIdxCopy | kIdxRagged);
                opos += blen;
            } else {
                const uint32_t P = q * kHashMul;
                const uint32_t h = P >> 16;
                const uint32_t e = stored_entry(q, P);
                uint32_t old = 0, w = lane;
                if (active) dict_step(tbl + 2u * h, lane, e, old, w);
                bool has_pred;
                uint32_t pred_e;
                resolve_groups(active, lane, w, e, ~0ull, has_pred, pred_e);
                const bool susp = active && e == 0 && h != 0;
                const uint32_t zbit = zmap_claim(zmap, susp, h, lane);
                const bool hit = active && (has_pred ? (pred_e == e) : (old == e && (!susp || zbit)));
                const uint64_t sig = ballot64(hit);
                const uint32_t nhit = (uint32_t)__builtin_popcountll(sig);
                const uint32_t off = kSig + 4u * lane - 2u * mbcnt64(sig);
                if (lane == 0) { st32u(rec, (uint32_t)sig); st32u(rec + 4, (uint32_t)(sig >> 32)); }
                if (active) {
                    if (hit) st16u(rec + off, h); else st32u(rec + off, q);
                }
                const uint32_t items_end = kSig + 4u * nq - 2u * nhit;
                if (lane < tail) rec[items_end + lane] = src[boff + 4u * nq + lane];
                if (idx && lane == 0) idx[nfull] = (uint8_t)kIdxRagged;
                opos += items_end + tail;
            }
        }
        if (lane == 0) sizes[chunk] = opos;
    }
}

__global__ __launch_bounds__(64) void chameleon_decode_chunks(const uint8_t* __restrict__ in,
                                                              const uint64_t* __restrict__ offsets,
                                                              const uint64_t* __restrict__ sizes,
                                                              uint8_t* __restrict__ out, uint64_t out_stride,
                                                              uint64_t out_total, uint32_t exact,
                                                              uint64_t* __restrict__ produced,
                                                              uint32_t* __restrict__ err) {
    const uint32_t lane = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const uint8_t* src = in + offsets[chunk];
    const uint64_t elen = sizes[chunk];
    uint8_t* dst = out + chunk * out_stride;
    // bytes this chunk may produce: its slice of the output
    const uint64_t room_all = out_total - chunk * out_stride;
    const uint64_t cap = room_all < out_stride ? room_all : out_stride;

    lds_clear(lane);
    const uint32_t tbl = lds_addr(smem);
    const uint32_t zmap = tbl + kTableBytes;

    Guard guard;
    uint64_t ipos = 0, opos = 0;
    bool bad = !decode_in_order(src, elen, dst, cap, guard, ipos, opos, tbl, ZmapLds{zmap}, lane);
    if (exact && !bad && opos != cap) bad = true;
    if (lane == 0) {
        produced[chunk] = opos;
        if (bad) atomicOr(err, 1u);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Pipelined decoder: one work-group (16 waves) per chunk, five stages one round (<= 8 records) apart, separated by one
// s_barrier per step.  At step s:
//
//   wave 1   "feeder"   round s+2: publishes each record's position — from the container's block index by a DPP prefix sum,
//                       or, without an index, by walking the record chain (signature -> popcount -> next record; the only
//                       inherently serial part of the format, codec.rs:88-100) in the LDS byte ring and running the
//                       copy-mode FSM.  It also issues the global->LDS DMA that keeps the ring filled.
//   waves 3-10 "fetch"  round s+1, one record each: pull every lane's item out of the ring (2-byte granular), hash PLAIN
//                       quads, stage the operands of the dictionary step {slot address, half, write flag | entry}.
//   wave 0   "dictionary wave"   round s: nothing but the ordered LDS exchange per record (PLAIN lanes write their entry,
//                       MAP lanes only read: mask 0) — exactly the sequential chameleon.rs:56-68 semantics — and one
//                       store of {entry now in the slot} per lane.  It is the critical path, so everything else is elsewhere.
//   wave 2   "finisher" round s-1: the zero-entry disambiguation (MAP of a never-written slot yields quad 0), in stream
//                       order; touches the zero-entry map about once per 64 Ki quads.
//   waves 11-15 "emit"  round s-2: entry -> quad (inverse of the hash product), coalesced 256-byte stores.
//
// The pipeline handles whole coded records and whole raw blocks that are followed by more data; everything the reference
// handles with per-unit checks (the ragged end: codec.rs:102-123) is left to decode_in_order on wave 0 once the
// pipeline has drained, starting from the feeder's final (position, FSM) state.
// ---------------------------------------------------------------------------------------------------------------
namespace {

constexpr uint32_t kDecWaves = 16;
constexpr uint32_t kRingBytes = 8192;                        // compressed-byte ring (power of two)
constexpr uint32_t kRingTiles = kRingBytes / 1024;
constexpr uint32_t kDescBytes = 128, kDescRing = 8;          // dwords 16..23 record positions, 24 copy mask, 25 count, 26 flags, 27 first ordinal, 28..29 the records' index entries (indexed feeder)
constexpr uint32_t kStageRec = 512, kStageRing = 4;          // per record: 64 x {d0, d1}; the dictionary wave turns d1 into the slot's entry
constexpr uint32_t kDRingBase = kTableBytes;                 // no zero-entry map in LDS here (ZmapGlobal)
constexpr uint32_t kDDescBase = kDRingBase + kRingBytes;
constexpr uint32_t kDStageBase = kDDescBase + kDescRing * kDescBytes;
constexpr uint32_t kDHandBase = kDStageBase + kStageRing * kRound * kStageRec;   // feeder -> wave 0 hand-over (32 B)
constexpr uint32_t kDIdxBase = kDHandBase + 32;              // block-index staging: 512 entries (two LDS-DMA pieces of 256)
constexpr uint32_t kLdsBytesDec = kDIdxBase + 512;
static_assert(kLdsBytesDec <= 160u * 1024u, "LDS budget");
constexpr uint32_t kFlagLast = 1u;
constexpr uint32_t kD0Write = 2u, kD0Half = 1u, kD0Empty = 0x80000000u, kD0Addr = 0x1fffcu;
constexpr uint32_t kFetchWave0 = 3, kEmitWave0 = 11, kNumEmit = kDecWaves - kEmitWave0;

}  // namespace

template <bool kProf>
__global__ __launch_bounds__(kDecWaves * 64) void chameleon_decode_chunks_pipe(const uint8_t* __restrict__ in,
                                                                               const uint64_t* __restrict__ offsets,
                                                                               const uint64_t* __restrict__ sizes,
                                                                               uint8_t* __restrict__ out, uint64_t out_stride,
                                                                               uint64_t out_total, uint32_t exact,
                                                                               const uint8_t* __restrict__ index,
                                                                               uint32_t* __restrict__ zmap_words,
                                                                               uint64_t* __restrict__ produced,
                                                                               uint32_t* __restrict__ err, uint32_t dbg, uint64_t* __restrict__ prof) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint64_t chunk = blockIdx.x;
    WaveClock<kProf> clk(blockIdx.x == 0 ? prof : nullptr);
    const uint8_t* src = in + offsets[chunk];
    const uint8_t* idx = index ? index + chunk * (out_stride / kBlock) : nullptr;   // this chunk's slice of the block index
    const uint64_t elen64 = sizes[chunk];
    uint8_t* dst = out + chunk * out_stride;
    const uint64_t room_all = out_total - chunk * out_stride;
    const uint64_t cap = room_all < out_stride ? room_all : out_stride;
    // the pipeline addresses the stream with 32-bit offsets; longer streams (only possible through the single-stream entry
    // points) are cut off here and finished by the in-order loop
    const uint32_t elen = elen64 > 0xfff00000ull ? 0xfff00000u : (uint32_t)elen64;
    const ZmapGlobal zmap{zmap_words + chunk * (kZmapBytes / 4)};
    if (wave == 0) __builtin_amdgcn_s_setprio(3);              // the dictionary wave is the critical path: first pick of issue slots

    {   // clear table, descriptor ring and this chunk's zero-entry map
        uint4* p = reinterpret_cast<uint4*>(smem);
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint32_t i = threadIdx.x; i < kTableBytes / 16; i += kDecWaves * 64) p[i] = z;
        for (uint32_t i = threadIdx.x; i < kDescRing * kDescBytes / 16; i += kDecWaves * 64) reinterpret_cast<uint4*>(smem + kDDescBase)[i] = z;
        for (uint32_t i = threadIdx.x; i < kZmapBytes / 16; i += kDecWaves * 64) reinterpret_cast<uint4*>(zmap.words)[i] = z;
        // the map is read back through L2 (ZmapGlobal::test) by this work-group only: the stores must have reached L2 (vmcnt(0),
        // then the barrier below); a device-scope fence would write back and invalidate the XCD's whole L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const uint32_t lds0 = lds_addr(smem);
    const uint32_t tbl = lds0;
    round_barrier();

    // parser state (wave 1)
    Guard guard;
    uint32_t ipos = 0, recs = 0;              // parse position; records published so far
    uint32_t tiles = 0;                       // DMA tiles issued
    uint32_t keep = 0;                        // start of the round parsed by the previous parse_round call
    uint32_t idle = 0;                        // consecutive rounds without progress (watchdog)
    bool parse_done = false;
    uint32_t last_round = 0xffffffffu;        // every wave learns it from the descriptor flags

    const uint32_t ntiles = (elen + 1023u) / 1024u;
    auto issue_tiles = [&](uint32_t limit_tile) {             // tiles < min(limit_tile, ntiles)
        while (tiles < ntiles && tiles < limit_tile) {
            const uint32_t off = tiles * 1024u + 16u * lane;
            if (off < elen) dma_1k(src + off, lds0 + kDRingBase + (tiles % kRingTiles) * 1024u);
            ++tiles;
        }
    };
    auto ring16 = [&](uint32_t pos) -> uint32_t {            // little-endian u16 at stream position pos (even)
        return *reinterpret_cast<const uint16_t*>(smem + kDRingBase + (pos & (kRingBytes - 1u)));
    };

    // ---- stage bodies ----
    // number of MAP flags (set bits) in the 8-byte signature at stream position pos: lanes 0..3 fetch one u16 each
    auto sig_hits = [&](uint32_t pos) -> uint32_t {
        const uint32_t part = lane < 4 ? ring16(pos + 2u * lane) : 0u;
        uint32_t c = (uint32_t)__builtin_popcount(part);
        c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x111, 0xf, 0xf, true);   // row_shr:1
        c += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)c, 0x112, 0xf, 0xf, true);   // row_shr:2
        return rlane(c, 3);
    };

    auto parse_round = [&](uint32_t r) {                      // wave 1
        // the record walker reads the stream here, so everything issued so far must have landed; the indexed feeder reads
        // nothing and waits at the end of the call, only for the tiles the next fetch needs
        wait_vm<0>();
        const uint32_t landed = tiles * 1024u < elen ? tiles * 1024u : elen;
        const uint32_t dbase = kDDescBase + (r % kDescRing) * kDescBytes;
        const uint32_t round_start = ipos, recs_before = recs;
        uint32_t rec = 0, copy_mask = 0, n = 0;
        constexpr uint32_t kMaxRound = kRound * (kSig + kBlock);
        if (!parse_done && guard.penalty == 0 && !guard.prev && ipos + kMaxRound <= landed && elen - ipos > kMaxRound &&
            ((uint64_t)recs + kRound) * kBlock <= cap) {
            // Fast round: eight coded records are certainly staged, complete and followed by more data, so the per-record work
            // is just the chain  signature -> popcount -> next position.  The FSM is advanced once per round unless a record
            // turns out incompressible (< 5 MAP flags: 8 + 256 - 2*hits >= 256, codec.rs:98), in which case the records after
            // it are dropped again and the careful loop below takes over.
            uint32_t pos = ipos, inc_at = kRound;
#pragma unroll
            for (uint32_t j = 0; j < kRound; ++j) {
                const uint32_t hits = sig_hits(pos);
                rec = lane == 16 + j ? pos : rec;
                if (hits <= 4 && inc_at == kRound) inc_at = j;
                pos += kSig + kBlock - 2u * hits;
            }
            if (inc_at == kRound) {
                const uint32_t to16 = (16u - (guard.counter & 15u)) & 15u;      // protection_state.rs:19-27, once per round
                if (to16 < kRound && guard.start > 1) guard.start >>= 1;
                guard.counter += kRound;
                ipos = pos;
                n = kRound;
                recs += kRound;
            }
            // else: nothing committed; reparse the round record by record with the full FSM
        }
        while (n < kRound && !parse_done) {
            const uint32_t rem = elen - ipos;
            if ((uint64_t)recs * kBlock + kBlock > cap) { parse_done = true; break; }   // the in-order loop reports the overflow
            if (guard.penalty > 0) {                          // raw block; the last one of a stream is left to the in-order loop
                if (rem <= kBlock) { parse_done = true; break; }                        // codec.rs:104-109
                if (ipos + kBlock > landed) break;            // not staged yet: short round
                (void)guard.block_is_copy();
                guard.decay();
                copy_mask |= 1u << n;
                rec = wlane_dyn(rec, ipos, 16 + n, lane);
                ipos += kBlock;
            } else {
                if (rem < kSig) { parse_done = true; break; }
                if (ipos + kSig > landed) break;
                const uint32_t len = kSig + kBlock - 2u * sig_hits(ipos);
                if (rem < len) { parse_done = true; break; }  // ragged last record: in-order loop
                if (ipos + len > landed) break;
                (void)guard.block_is_copy();
                guard.update(len >= kBlock);                  // codec.rs:98
                rec = wlane_dyn(rec, ipos, 16 + n, lane);
                ipos += len;
            }
            ++n;
            ++recs;
        }
        // watchdog: the ring always has room for the next round, so an empty round means the DMA has not
        // landed yet; after a few of them give the rest of the stream to the in-order loop rather than spin
        idle = (n == 0 && !parse_done) ? idle + 1 : 0;
        if (idle >= 8) parse_done = true;
        if (parse_done && last_round == 0xffffffffu) last_round = r;
        rec = wlane_dyn(rec, copy_mask, 24, lane);
        rec = wlane_dyn(rec, n, 25, lane);
        rec = wlane_dyn(rec, (last_round == r) ? kFlagLast : 0u, 26, lane);
        rec = wlane_dyn(rec, recs_before, 27, lane);
        if (lane >= 16 && lane < 28) *reinterpret_cast<uint32_t*>(smem + dbase + 4u * lane) = rec;
        // Refill.  While these tiles land, the fetch waves read the round parsed by the PREVIOUS call (it starts at `keep`),
        // so tile i may only replace tile i-8 if that one ends at or before `keep`.
        issue_tiles(keep / 1024u + kRingTiles);
        keep = round_start;
    };

    // ---- indexed feeder (wave 1): the block index says where every record starts and which blocks are raw copies, so there is
    // no record chain to walk, no FSM to run and no stream byte to read (include/density_hip.h).  The index itself is staged in
    // LDS by DMA, 256 entries at a time; positions are computed for a window of 64 records at once (one per lane, wave-wide
    // prefix sum of the record lengths); a round then only publishes its eight.  Kept to a minimum of instructions: a lone
    // wavefront retires roughly one instruction per 8 cycles.
    const uint32_t nblocks_out = (uint32_t)((cap + kBlock - 1) / kBlock);
    uint32_t win_pos = 0, win_end = 0, win_ent = 0;           // per lane: position / end / index entry of record idx_base + lane
    uint32_t win_base = 0, win_used = 64, win_stop = 0;       // first record of the window, records consumed, first lane that stops the pipeline
    uint64_t win_copy = 0;
    uint32_t idx_staged = 0;                                  // index entries [0, idx_staged) have been requested
    auto stage_index = [&]() {                                // one DMA: entries idx_staged .. +255 -> LDS (4 per lane)
        const uint32_t first = idx_staged + 4u * lane;
        if (first < nblocks_out) {
            uint32_t keepm0;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keepm0) : "v"(idx + first), "s"(rfl(lds0 + kDIdxBase + (idx_staged & 511u))) : "memory");
        }
        idx_staged += 256;
    };
    auto feed_round = [&](uint32_t r) {
        const uint32_t dbase = kDDescBase + (r % kDescRing) * kDescBytes;
        const uint32_t round_start = ipos, recs_before = recs;
        uint32_t n = 0, copy_mask = 0;
        if (!parse_done) {
            if (win_used == 64) {                                 // next window of 64 records
                win_base = recs;
                if (win_base + 128 > idx_staged && idx_staged < nblocks_out) stage_index();   // two windows ahead; landed long before use
                const uint32_t rec_no = win_base + lane;
                const uint32_t ent = rec_no < nblocks_out ? (uint32_t)*reinterpret_cast<const uint8_t*>(smem + kDIdxBase + (rec_no & 511u)) : kIdxRagged;
                const uint32_t mylen = (ent & kIdxCopy) ? kBlock : (kSig + kBlock - 2u * (ent & 0x7fu));
                uint32_t incl = mylen;                            // inclusive scan: within rows of 16 by DPP, then the three row totals
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
                const uint32_t t0 = rlane(incl, 15), t1 = rlane(incl, 31), t2 = rlane(incl, 47);
                incl += (lane >= 16 ? t0 : 0u) + (lane >= 32 ? t1 : 0u) + (lane >= 48 ? t2 : 0u);
                win_end = ipos + incl;
                win_pos = win_end - mylen;
                win_ent = ent;
                // ragged last block, last block of the chunk, or an index that disagrees with the stream length: the in-order loop
                // finishes from there, told whether its (single) block is a raw copy
                const bool stop = (ent & 0x7fu) == kIdxRagged || win_pos >= elen || elen - win_pos <= mylen || ((uint64_t)rec_no + 1) * kBlock > cap;   // (a partial last output block is the in-order loop's)
                const uint64_t stopm = ballot64(stop);
                win_stop = stopm ? (uint32_t)__builtin_ctzll(stopm) : 64u;
                win_copy = ballot64((ent & kIdxCopy) != 0);
                win_used = 0;
            }
            const uint32_t left = win_stop - win_used;           // win_stop >= win_used while !parse_done
            n = left < kRound ? left : kRound;
            copy_mask = (uint32_t)(win_copy >> win_used) & ((1u << n) - 1u);
            if (lane >= win_used && lane < win_used + n) {
                *reinterpret_cast<uint32_t*>(smem + dbase + 64 + 4u * (lane - win_used)) = win_pos;
                smem[dbase + 112 + (lane - win_used)] = (uint8_t)win_ent;                        // what the index says of the record: held against its signature by the wave that reads it
            }
            if (n) ipos = rlane(win_end, win_used + n - 1);
            recs += n;
            if (n < kRound) {
                parse_done = true;
                guard.penalty = ((win_copy >> (win_used + n)) & 1ull) ? 1u : 0u; guard.start = 1; guard.prev = 0; guard.counter = 1;
                if (last_round == 0xffffffffu) last_round = r;
            }
            win_used += n;
        }
        if (lane == 0) *reinterpret_cast<uint4*>(smem + dbase + 96) = make_uint4(copy_mask, n, (last_round == r) ? kFlagLast : 0u, recs_before);
        // Refill.  While these tiles land, the fetch waves read the round published by the PREVIOUS call (it starts at `keep`),
        // so tile i may only replace tile i-8 if that one ends at or before `keep`.
        issue_tiles(keep / 1024u + ((dbg & 512u) ? kRingTiles - 2u : kRingTiles));
        keep = round_start;
        // The fetch waves read this round's bytes [round_start, ipos) during the next step.  DMA retires in order, so exactly
        // the tiles issued beyond that range may stay in flight (an index piece in the queue only makes the wait stricter).
        const uint32_t need = (ipos + 1023u) / 1024u;            // tiles [0, need) must have landed
        const uint32_t in_flight_ok = tiles > need ? tiles - need : 0u;
        if (in_flight_ok >= 4) wait_vm<4>(); else wait_vm<0>();   // (steady state keeps 4-5 tiles in flight)
    };

    auto fetch_round = [&](uint32_t r) {                      // waves 3..10, record w of the round
        const uint32_t w = wave - kFetchWave0;
        const uint32_t dbase = kDDescBase + (r % kDescRing) * kDescBytes;
        const uint32_t sbase = kDStageBase + (r % kStageRing) * kRound * kStageRec;
        // descriptor and this wave's record position in one LDS round trip
        const uint4 dt = *reinterpret_cast<const uint4*>(smem + dbase + 96);       // {copy mask, count, flags, first ordinal}
        const uint32_t posv = *reinterpret_cast<const uint32_t*>(smem + dbase + 64 + 4u * w);
        const uint32_t n = rfl(dt.y), copy_mask = rfl(dt.x), flags = rfl(dt.z);
        const uint32_t pos = rfl(posv);
        // a coded record: signature, then one 2- or 4-byte item per lane
        auto stage_coded = [&]() {
            const uint32_t part = lane < 4 ? ring16(pos + 2u * lane) : 0u;       // the record's signature (codec.rs:28-31)
            const uint64_t sig = (uint64_t)(rlane(part, 0) | (rlane(part, 1) << 16)) | ((uint64_t)(rlane(part, 2) | (rlane(part, 3) << 16)) << 32);
            const bool hit = (sig >> lane) & 1ull;
            // the index must agree with the stream it describes (include/density_hip.h): a coded record's entry is its signature's MAP count
            if (idx && (uint32_t)__builtin_popcountll(sig) != (uint32_t)(smem[dbase + 112 + w] & 0x7fu)) { if (lane == 0) atomicOr(err, 8u); }
            const uint32_t a = pos + kSig + 4u * lane - 2u * mbcnt64(sig);
            // both halves of a possible quad are read whether or not the lane holds a MAP item (2 bytes): no divergence, one wait
            const uint32_t lo = ring16(a), hi = ring16(a + 2);
            // MAP: the item is the slot index (chameleon.rs:64-68).  PLAIN: hash the quad, stage its entry (chameleon.rs:56-61).
            const uint32_t q = lo | (hi << 16);
            const uint32_t P = q * kHashMul;
            const uint32_t h = hit ? lo : (P >> 16);
            const uint32_t d0 = ((h >> 1) << 2) | (h & 1u) | (hit ? 0u : kD0Write);
            const uint32_t d1 = hit ? 0u : (stored_entry(q, P) << ((h & 1u) << 4));
            *reinterpret_cast<uint2*>(smem + sbase + w * kStageRec + 8u * lane) = make_uint2(d0, d1);
        };
        if (__builtin_expect((copy_mask | flags) == 0 && n == kRound, 1)) { stage_coded(); return; }   // the common round: eight coded records
        if (flags & kFlagLast) last_round = r;                  // every wave must learn where to stop
        if (w >= n) return;
        if ((copy_mask >> w) & 1u) {
            const uint32_t a = pos + 4u * lane;
            *reinterpret_cast<uint2*>(smem + sbase + w * kStageRec + 8u * lane) = make_uint2(0u, ring16(a) | (ring16(a + 2) << 16));
        } else {
            stage_coded();
        }
    };

    auto dict_round = [&](uint32_t r) {                       // wave 0
        const uint32_t dbase = kDDescBase + (r % kDescRing) * kDescBytes;
        const uint32_t sbase = kDStageBase + (r % kStageRing) * kRound * kStageRec;
        // descriptor and all eight staged records in one LDS round trip (records beyond the round's count are read but unused)
        const uint4 dt = *reinterpret_cast<const uint4*>(smem + dbase + 96);       // {copy mask, count, flags, first ordinal}
        uint32_t d0[kRound], d1[kRound], ret[kRound];
#pragma unroll
        for (uint32_t j = 0; j < kRound; ++j) {
            const uint2 v = *reinterpret_cast<const uint2*>(smem + sbase + j * kStageRec + 8u * lane);
            d0[j] = v.x; d1[j] = v.y; ret[j] = 0;
        }
        const uint32_t n = rfl(dt.y), copy_mask = rfl(dt.x), flags = rfl(dt.z);
        const bool common = (copy_mask | flags) == 0 && n == kRound;   // eight coded records, not the last round
        auto operands_landed = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d0[0]), "+v"(d0[1]), "+v"(d0[2]), "+v"(d0[3]), "+v"(d0[4]), "+v"(d0[5]), "+v"(d0[6]), "+v"(d0[7]),
                                                  "+v"(d1[0]), "+v"(d1[1]), "+v"(d1[2]), "+v"(d1[3]), "+v"(d1[4]), "+v"(d1[5]), "+v"(d1[6]), "+v"(d1[7]) :: "memory");
        };
        uint32_t mask[kRound];
        auto issue = [&](uint32_t j) {
            const uint32_t sh = (d0[j] & kD0Half) << 4;
            mask[j] = (uint32_t)(((int32_t)(d0[j] << 30) >> 31) & 0xffff) << sh;   // write flag -> 0xffff or 0
            dict_xchg_issue(tbl + (d0[j] & kD0Addr), mask[j], d1[j], ret[j]);
        };
        // the dword as it stands after this lane's turn — (old & ~mask) | entry: its own entry for PLAIN lanes, what it read for MAP
        // lanes (mask and entry are 0 there) — goes back in place of the operand; the emit waves pick the lane's half
        const uint32_t wbase = lds0 + sbase + 8u * lane;
        auto finish = [&](uint32_t j) {
            const uint32_t m = (ret[j] & ~mask[j]) | d1[j];
            switch (j) {                                          // (offsets are instruction immediates)
                case 0: asm volatile("ds_write_b32 %0, %1 offset:4" ::"v"(wbase), "v"(m) : "memory"); break;
                case 1: asm volatile("ds_write_b32 %0, %1 offset:516" ::"v"(wbase), "v"(m) : "memory"); break;
                case 2: asm volatile("ds_write_b32 %0, %1 offset:1028" ::"v"(wbase), "v"(m) : "memory"); break;
                case 3: asm volatile("ds_write_b32 %0, %1 offset:1540" ::"v"(wbase), "v"(m) : "memory"); break;
                case 4: asm volatile("ds_write_b32 %0, %1 offset:2052" ::"v"(wbase), "v"(m) : "memory"); break;
                case 5: asm volatile("ds_write_b32 %0, %1 offset:2564" ::"v"(wbase), "v"(m) : "memory"); break;
                case 6: asm volatile("ds_write_b32 %0, %1 offset:3076" ::"v"(wbase), "v"(m) : "memory"); break;
                default: asm volatile("ds_write_b32 %0, %1 offset:3588" ::"v"(wbase), "v"(m) : "memory"); break;
            }
        };
        if (__builtin_expect(common, 1)) {                        // the common round: eight coded records, straight-line code
            operands_landed();
#pragma unroll
            for (uint32_t j = 0; j < kRound; ++j) issue(j);
#pragma unroll
            for (uint32_t j = 0; j < kRound; ++j) {
                // outstanding: the exchanges after j and the j results already written back
                asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(ret[j]) :: "memory");
                finish(j);
            }
        } else {
            if (flags & kFlagLast) last_round = r;
            if (n == 0) return;
            operands_landed();
            const uint32_t live = ((1u << n) - 1u) & ~copy_mask;  // records that go through the table
#pragma unroll
            for (uint32_t j = 0; j < kRound; ++j) {
                if ((live >> j) & 1u) issue(j);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ret[0]), "+v"(ret[1]), "+v"(ret[2]), "+v"(ret[3]), "+v"(ret[4]), "+v"(ret[5]), "+v"(ret[6]), "+v"(ret[7]) :: "memory");
#pragma unroll
            for (uint32_t j = 0; j < kRound; ++j) {
                if ((live >> j) & 1u) finish(j);
            }
        }
    };

    // zero-entry disambiguation, in stream order (see the file header; entries are salted, so this fires about once per 64 Ki quads)
    auto finish_round = [&](uint32_t r) {                     // wave 2
        const uint32_t dbase = kDDescBase + (r % kDescRing) * kDescBytes;
        const uint32_t sbase = kDStageBase + (r % kStageRing) * kRound * kStageRec;
        // descriptor and all eight records in one LDS round trip; a per-lane OR of "stored entry 0 outside slot 0"; the ballots and
        // the serial part run only when a zero entry shows up (records beyond the round's count hold stale data: filtered there)
        const uint4 dt = *reinterpret_cast<const uint4*>(smem + dbase + 96);       // {copy mask, count, flags, first ordinal}
        uint2 v[kRound];
#pragma unroll
        for (uint32_t j = 0; j < kRound; ++j) v[j] = *reinterpret_cast<const uint2*>(smem + sbase + j * kStageRec + 8u * lane);
        const uint32_t n = rfl(dt.y), copy_mask = rfl(dt.x);
        if (rfl(dt.z) & kFlagLast) last_round = r;
        if (n == 0) return;
        uint32_t zacc = 0;
#pragma unroll
        for (uint32_t j = 0; j < kRound; ++j) {
            const uint32_t sh = (v[j].x & kD0Half) << 4;
            const uint32_t e16 = (v[j].y >> sh) & 0xffffu;
            zacc |= e16 == 0 ? (v[j].x & (kD0Addr | kD0Half)) : 0u;
        }
        if (!ballot64(zacc != 0)) return;
        const uint32_t live = ((1u << n) - 1u) & ~copy_mask;
        uint64_t zeros[kRound];
        uint32_t any = 0;
#pragma unroll
        for (uint32_t j = 0; j < kRound; ++j) {
            const uint32_t sh = (v[j].x & kD0Half) << 4;
            const bool zero = ((v[j].y >> sh) & 0xffffu) == 0 && (v[j].x & (kD0Addr | kD0Half)) != 0;   // stored entry 0 outside slot 0
            zeros[j] = ((live >> j) & 1u) ? ballot64(zero) : 0ull;
            any |= (zeros[j] != 0) ? 1u : 0u;
        }
        if (!any) return;
#pragma unroll
        for (uint32_t j = 0; j < kRound; ++j) {
            uint64_t todo = zeros[j];
            while (todo) {                                        // ascending record, ascending lane == stream order
                const uint32_t l = (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t x = rlane(v[j].x, l);
                const uint32_t h = ((x & kD0Addr) >> 1) | (x & kD0Half);
                if (x & kD0Write) { if (lane == l) zmap.set(h); }  // PLAIN wrote a genuine zero entry
                else if (!zmap.test(h) && lane == l)              // MAP read a never-written slot: quad 0 (chameleon.rs:64-68 on a zero word)
                    *reinterpret_cast<uint32_t*>(smem + sbase + j * kStageRec + 8u * lane) = v[j].x | kD0Empty;
            }
        }
    };

    auto emit_round = [&](uint32_t r) {                       // waves 11..15: records w and w + 5 of the round
        const uint32_t w = wave - kEmitWave0;
        const uint32_t dbase = kDDescBase + (r % kDescRing) * kDescBytes;
        const uint32_t sbase = kDStageBase + (r % kStageRing) * kRound * kStageRec + 8u * lane;
        // descriptor and both staged records in one LDS round trip (a record beyond the round's count is read but unused)
        const uint4 dt = *reinterpret_cast<const uint4*>(smem + dbase + 96);       // {copy mask, count, flags, first ordinal}
        const uint2 va = *reinterpret_cast<const uint2*>(smem + sbase + w * kStageRec);
        const uint2 vb = *reinterpret_cast<const uint2*>(smem + sbase + ((w + kNumEmit) & (kRound - 1u)) * kStageRec);
        const uint32_t n = rfl(dt.y), copy_mask = rfl(dt.x), flags = rfl(dt.z), first = rfl(dt.w);
        auto quad_coded = [&](const uint2& v) -> uint32_t {
            const uint32_t sh = (v.x & kD0Half) << 4;
            const uint32_t h = ((v.x & kD0Addr) >> 1) | (v.x & kD0Half);
            return (v.x & kD0Empty) ? 0u : entry_to_quad(h, (v.y >> sh) & 0xffffu);
        };
        uint8_t* base = dst + (uint64_t)first * kBlock;           // wave-uniform
        if (__builtin_expect((copy_mask | flags) == 0 && n == kRound, 1)) {          // the common round: eight coded records
            gstore32(base + w * kBlock, 4u * lane, quad_coded(va));
            if (w + kNumEmit < kRound) gstore32(base + (w + kNumEmit) * kBlock, 4u * lane, quad_coded(vb));
            return;
        }
        if (flags & kFlagLast) last_round = r;
        if (w < n) gstore32(base + w * kBlock, 4u * lane, ((copy_mask >> w) & 1u) ? va.y : quad_coded(va));
        if (w + kNumEmit < n) gstore32(base + (w + kNumEmit) * kBlock, 4u * lane, ((copy_mask >> (w + kNumEmit)) & 1u) ? vb.y : quad_coded(vb));
    };

    // ---- prologue: fill the pipeline ----
    const bool is_fetch = wave >= kFetchWave0 && wave < kEmitWave0;
    if (wave == 1) {
        if (idx) { stage_index(); issue_tiles(kRingTiles); wait_vm<0>(); feed_round(0); feed_round(1); }
        else { issue_tiles(kRingTiles); parse_round(0); parse_round(1); }
    }
    round_barrier();
    if (is_fetch) fetch_round(0);
    round_barrier();

    // ---- steady state: step s = feed s+2 | fetch s+1 | dictionary s | finish s-1 | emit s-2 ----
    // (dbg bits 16/32/64/256 idle the emit/fetch/dictionary/finisher stage for timing experiments; a stage that is idled still
    // has to learn the last round from the descriptor flags)
    auto only_flags = [&](uint32_t r) {
        if (rfl(*reinterpret_cast<const uint32_t*>(smem + kDDescBase + (r % kDescRing) * kDescBytes + 104)) & kFlagLast) last_round = r;
    };
    // one loop per role (the role never changes): the compiler keeps only that role's state live in each loop
    auto step_end = [&](uint32_t s) -> bool {
        clk.work_done();
        clk.stat_post(kLdsBytesDec, wave, lane);
        round_barrier();
        clk.wait_done();
        clk.stat_take(kLdsBytesDec, wave, lane);
        return last_round != 0xffffffffu && s >= last_round + 2;
    };
#define DENSITY_ROLE_LOOP(BODY) for (uint32_t s = 0;; ++s) { clk.start(); BODY; if (step_end(s)) break; }
    if (wave == 1) {
        if (idx) DENSITY_ROLE_LOOP(feed_round(s + 2)) else DENSITY_ROLE_LOOP(parse_round(s + 2))
    } else if (wave == 0) {
        if (!(dbg & 64u)) DENSITY_ROLE_LOOP(dict_round(s)) else DENSITY_ROLE_LOOP(only_flags(s))
    } else if (wave == 2) {
        if (!(dbg & 256u)) DENSITY_ROLE_LOOP(if (s >= 1) finish_round(s - 1)) else DENSITY_ROLE_LOOP(if (s >= 1) only_flags(s - 1))
    } else if (is_fetch) {
        if (!(dbg & 32u)) DENSITY_ROLE_LOOP(fetch_round(s + 1)) else DENSITY_ROLE_LOOP(only_flags(s + 1))
    } else {
        if (!(dbg & 16u)) DENSITY_ROLE_LOOP(if (s >= 2) emit_round(s - 2)) else DENSITY_ROLE_LOOP(if (s >= 2) only_flags(s - 2))
    }
#undef DENSITY_ROLE_LOOP
    clk.flush(wave, lane);
    clk.flush_stat(wave, lane);

    // hand the feeder's final state to wave 0, which finishes the ragged end of the stream in order
    if (wave == 1 && lane == 0) {
        uint32_t* hand = reinterpret_cast<uint32_t*>(smem + kDHandBase);
        hand[0] = ipos; hand[1] = recs; hand[2] = guard.penalty; hand[3] = guard.start; hand[4] = guard.prev; hand[5] = guard.counter;
    }
    round_barrier();
    if (wave == 0) {
        const uint32_t* hand = reinterpret_cast<const uint32_t*>(smem + kDHandBase);
        Guard g;
        uint64_t ip = rfl(hand[0]), op = (uint64_t)rfl(hand[1]) * kBlock;
        g.penalty = rfl(hand[2]); g.start = rfl(hand[3]); g.prev = rfl(hand[4]); g.counter = rfl(hand[5]);
        bool bad = !decode_in_order(src, elen64, dst, cap, g, ip, op, tbl, zmap, lane);
        if (exact && !bad && op != cap) bad = true;
        if (lane == 0) {
            produced[chunk] = op;
            if (bad) atomicOr(err, 1u);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------------------
bool g_force_simple = false;     // test hook (density_hip_set_kernel_variant): run the one-wave kernels
bool g_force_pipeline = false;   // test hook: the 16-wave role pipelines of this file instead of rotor.hip's wave-rotation kernels
bool g_exchange_unsafe = false, g_rotor_unsafe = false;   // set by the start-up self-test (api.hip): which kernel families this device may run

namespace {
// DENSITY_HIP_PROF=1: per-wave cycle accounting of work-group 0, printed to stderr after every pipelined launch (synchronises)
uint64_t* prof_buffer() {
    static uint64_t* buf = nullptr;
    if (!debug_env("DENSITY_HIP_PROF")) return nullptr;
    if (!buf && hipMalloc((void**)&buf, 72 * sizeof(uint64_t)) != hipSuccess) buf = nullptr;
    if (buf) (void)hipMemset(buf, 0, 72 * sizeof(uint64_t));
    return buf;
}
void prof_report(const char* what, uint64_t* buf, hipStream_t stream) {
    if (!buf) return;
    uint64_t h[72];
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
    fprintf(stderr, "[density_hip prof] %s work-group 0: ", what);
    for (int w = 0; w < 16; ++w) if (h[2 * w] | h[2 * w + 1]) fprintf(stderr, "w%d %lluk/%lluk | ", w, (unsigned long long)h[2 * w] / 1000, (unsigned long long)h[2 * w + 1] / 1000);
    fprintf(stderr, "\n[density_hip prof] %s wave-0 phases:", what);
    for (int k = 0; k < 8; ++k) fprintf(stderr, " p%d %llu", k, (unsigned long long)h[32 + k]);
    fprintf(stderr, "\n[density_hip prof] %s slowest wave per step (count):", what);
    for (int w = 0; w < 16; ++w) fprintf(stderr, " w%d %llu", w, (unsigned long long)h[48 + w]);
    fprintf(stderr, " | sum of per-step maxima %lluk\n", (unsigned long long)h[47] / 1000);
}
}  // namespace
hipError_t launch_chameleon_encode(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks,
                                   uint8_t* d_out, uint64_t out_stride, uint64_t* d_sizes, uint8_t* d_index, uint32_t* d_zmap, uint32_t* d_err, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)chameleon_encode_chunks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) return e;
    if (n_chunks == 0) return hipSuccess;
    if (!g_force_simple && !g_force_pipeline && !g_rotor_unsafe && rotor_encode_eligible(d_in, total, chunk_bytes, n_chunks))
        return launch_rotor_encode(d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_index, d_err, stream);   // default path: rotor.hip
    // the pipelined kernel stages its input with 16-byte LDS-DMA pieces: needs 16-byte aligned chunk bases
    // (and counts blocks in 32 bits, keeps its zero-entry maps in the workspace)
    const bool aligned = ((uintptr_t)d_in % 16 == 0) && (n_chunks == 1 || chunk_bytes % 16 == 0);
    const bool fits = (n_chunks == 1 ? total : chunk_bytes) < (1ull << 40) && d_zmap && n_chunks <= kMaxPipelinedChunks;
    if (aligned && fits && !g_force_simple && !g_exchange_unsafe) {
        uint64_t* prof = prof_buffer();
        auto kernel = prof ? chameleon_encode_chunks_pipe<true> : chameleon_encode_chunks_pipe<false>;
        e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesPipe + 64);
        if (e != hipSuccess) return e;
#ifdef DENSITY_HIP_DEBUG
        const uint32_t dbg = debug_env("DENSITY_HIP_DBG") ? (uint32_t)atoi(debug_env("DENSITY_HIP_DBG")) : 0u;   // stage-idling switches: debug builds only
#else
        const uint32_t dbg = 0u;
#endif
        hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(kPipeWaves * 64), kLdsBytesPipe + 64, stream, d_in, total, chunk_bytes, d_out, out_stride, d_sizes, d_index, d_zmap, dbg, prof);
        prof_report("encode", prof, stream);
    } else {
        hipLaunchKernelGGL(chameleon_encode_chunks, dim3(n_chunks), dim3(64), kLdsBytes, stream, d_in, total, chunk_bytes, d_out, out_stride, d_sizes, d_index);
    }
    return hipGetLastError();
}

hipError_t launch_chameleon_decode(const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes,
                                   uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride, uint64_t out_total,
                                   bool exact, const uint8_t* d_index, uint32_t* d_zmap, uint64_t* d_produced, uint32_t* d_err, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)chameleon_decode_chunks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
    if (e != hipSuccess) return e;
    if (n_chunks == 0) return hipSuccess;
    if (!g_force_simple && !g_force_pipeline && !g_rotor_unsafe && rotor_decode_eligible(d_out, n_chunks, out_stride, out_total, d_index, d_zmap))
        return launch_rotor_decode(d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact, d_index, d_zmap, d_produced, d_err, stream);   // default: rotor.hip
    // the pipelined kernel stages the stream with 16-byte LDS-DMA pieces (container payloads are 16-byte aligned) and stores
    // quads with aligned dwords
    const bool aligned = ((uintptr_t)d_in % 16 == 0) && ((uintptr_t)d_out % 4 == 0) && (n_chunks == 1 || out_stride % 4 == 0);
    if (aligned && !g_force_simple && !g_exchange_unsafe && d_zmap && n_chunks <= kMaxPipelinedChunks) {
        uint64_t* prof = prof_buffer();
        auto kernel = prof ? chameleon_decode_chunks_pipe<true> : chameleon_decode_chunks_pipe<false>;
        e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytesDec + 64);
        if (e != hipSuccess) return e;
#ifdef DENSITY_HIP_DEBUG
        const uint32_t dbg = debug_env("DENSITY_HIP_DBG") ? (uint32_t)atoi(debug_env("DENSITY_HIP_DBG")) : 0u;   // stage-idling switches: debug builds only
#else
        const uint32_t dbg = 0u;
#endif
        hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(kDecWaves * 64), kLdsBytesDec + 64, stream, d_in, d_offsets, d_sizes, d_out, out_stride, out_total, exact ? 1u : 0u,
                           // the feeder stages the index with 4-byte DMA pieces: per-chunk slices must start 4-byte aligned
                           (d_index && (out_stride / 256) % 4 == 0 && (uintptr_t)d_index % 4 == 0) ? d_index : nullptr, d_zmap, d_produced, d_err, dbg, prof);
        prof_report("decode", prof, stream);
    } else {
        hipLaunchKernelGGL(chameleon_decode_chunks, dim3(n_chunks), dim3(64), kLdsBytes, stream, d_in, d_offsets, d_sizes, d_out, out_stride, out_total, exact ? 1u : 0u, d_produced, d_err);
    }
    return hipGetLastError();
}

}  // namespace density
