// rotor.hip — Chameleon wave-rotation kernels for gfx950 (MI355X): the default encode / index-fed decode path.
//
// One work-group of W wavefronts owns one chunk (= one independent reference stream, chameleon.rs:45-53) and its dictionary
// (64 Ki exact 16-bit entries = 128 KiB of LDS, chameleon_dev.hpp).  The chunk is cut into ROUNDS of R blocks; the kernels are
// templates <R, W> and what ships is R = 16 on W = 8 waves for the encoder (4 KiB rounds, 256 registers per wave: the quads of a
// round stay in registers from the hash to the emit) and R = 12 on W = 12 for the decoder (the numbers 8 and 16 in the text
// below are the original geometry, rounds of 8 on 16 waves, still selectable: DENSITY_HIP_TUNE).  Wave w
// takes the rounds r = w, w + W, w + 2W, ... and does EVERYTHING for its round itself, in registers: global loads, hashing,
// the dictionary step, signatures, the copy-mode FSM, record offsets, stores.  There are no staging rings and no per-round
// work-group barrier.  What IS sequential in the reference — the dictionary (every quad sees the table its predecessors left,
// chameleon.rs:88-100) and the running output position / ProtectionState (codec.rs:34-70) — is passed from round to round by
// two token chains through LDS:
//
//   D chain  "dictionary token".  The holder issues its 8 ordered exchanges (ds_mskor_rtn_b32: one instruction = the 64
//            sequential dictionary steps of a block, LDS lane order; chameleon_dev.hpp) back to back from prepared registers
//            and writes the token for the next round BEHIND them in its own LDS instruction stream.  A wave's LDS
//            instructions execute in issue order, so whoever sees the new token also sees the table after those exchanges.
//            This chain is the critical path of the kernel: ~8 x 23 cycles of exchanges + one LDS write->read hand-off per round.
//   O chain  "commit token" + payload {output position, FSM state}.  After its exchanges a wave turns the 8 answers into 8
//            signatures (hit == answer equals own entry; __ballot == the signature word, io/write_signature.rs:14-17), waits for
//            the commit token, runs the FSM over its 8 blocks in closed form, passes position + state on, and only then stores.
//
// Copy mode (protection_state.rs) is a feedback from the signatures to "which blocks touch the dictionary at all", so the
// exchanges of a round are speculative: "no raw-copy block in this round".  The commit step knows the truth.  When it finds
// a block that had to be a raw copy it raises an abort: all 16 waves meet at a barrier, the rounds that exchanged after the
// last committed one roll their blocks back in reverse order (the lowest lane of a slot holds the pre-block entry, so the
// answers are written back lane-reversed with one ds_write_b16 per block), and the chain restarts at the failed round in SLOW
// mode: the token holder first waits for its commit payload and then walks its blocks one by one with the full FSM
// (raw-copy blocks skip the dictionary).  Slow mode ends after a round that leaves the FSM calm.  Every chunk starts in slow
// mode (its first blocks are incompressible by construction: empty dictionary).  Rounds that contain a zero entry outside
// slot 0 (zero-entry map, about one quad in 64 Ki) and the chunk's last partial round are also walked in order.
//
// Decode (index-fed: the container's block index gives record lengths and raw-copy blocks, include/density_hip.h) uses the
// same D chain for the dictionary (MAP lanes exchange with mask 0 = read, PLAIN lanes write: chameleon.rs:56-68); record
// positions of the whole chunk come from one prefix sum over the index at kernel start; a Z chain orders the rare
// zero-entry-map accesses.  Streams without an index, chunks above 4 MiB and unaligned buffers run on chameleon.hip's kernels.
#include <cstdio>
#include <cstdlib>

#include "chameleon_dev.hpp"
#include "kernels.hpp"

namespace density {

namespace {

constexpr uint32_t kRotWaves = 16, kRotThreads = kRotWaves * 64;
constexpr uint32_t kR = 8;                                   // blocks per round
constexpr uint32_t kNone = 0xffffffffu;
// sync block (bytes from its base): D line {D, A}; O line {O, A', P0, P1}; a 256-byte sink for the idle lanes of a token write;
// 16 words "rounds whose zero-entry-map phase this wave has finished" (decoder); 16 words "the round of this wave that is about to mark the map" (decoder)
constexpr uint32_t kSyD = 0, kSyO = 16, kSyZ = 32, kSyEnd = 48, kSySink = 64, kSyZdone = 64 + 256, kSyWsum = 64 + 256 + 64, kSyZset = 64 + 256 + 64 + 64, kSyBytes = 64 + 256 + 64 + 64 + 64;
// (encoder: the words of the decoder's zero-entry chain hold the memo of FSM predictions instead — 8 entries of {state, raw-copy blocks, end state, -})
constexpr uint32_t kSyMemo = kSyZdone, kMemoEntries = 8;
constexpr uint32_t kSyPage = kSyZ;                                             // (PAGED encoder: 16 bytes of page state, the commit token's holder's)
static_assert(kSyMemo + 16u * kMemoEntries <= kSyBytes, "memo inside the sync block");
// encoder LDS: table | zero-entry map | sync
constexpr uint32_t kEncZmap = kTableBytes, kEncSync = kTableBytes + kZmapBytes, kEncStage = kEncSync + kSyBytes;
// (encoder staging: two arrays of up to 16 blocks x 64 lanes for the rolled loops of the rare paths — rollback, in-order rounds, zero-entry
// quads at commit — which exclude one another in time, so the whole work-group shares one copy)
constexpr uint32_t kEncStageBytes = 2u * 16u * 256u, kEncLds = kEncStage + kEncStageBytes;
// decoder LDS: table | block-index copy | round positions | zero-entry map (rounds of 12 and more; else in global memory: ZmapGlobal) | sync
constexpr uint32_t kRotMaxBlocks = 16384;                    // blocks per chunk the decoder keeps an index copy for (4 MiB chunks)
// (per round length R: rounds of 12 and more leave room for the zero-entry map in LDS — a look-up in global memory is a memory round trip
// of microseconds, and one round in 25 has one on repetitive text; rounds of 8 keep it in global memory)
constexpr uint32_t kDecIdx = kTableBytes, kDecPos = kDecIdx + kRotMaxBlocks;
constexpr uint32_t dec_pos_bytes(uint32_t R) { return ((kRotMaxBlocks / R + 1u) * 4u + 15u) & ~15u; }
constexpr bool dec_zmap_in_lds(uint32_t R) { return kDecPos + dec_pos_bytes(R) + kZmapBytes + kSyBytes <= 160u * 1024u; }
constexpr uint32_t dec_zmap_at(uint32_t R) { return kDecPos + dec_pos_bytes(R); }
constexpr uint32_t dec_sync_at(uint32_t R) { return dec_zmap_at(R) + (dec_zmap_in_lds(R) ? kZmapBytes : 0u); }
constexpr uint32_t dec_lds_bytes(uint32_t R) { return dec_sync_at(R) + kSyBytes; }
// PAGED decoder: behind the sync block, per page of the chunk its first block and what turns a stream position into an offset from page 0
constexpr uint32_t kDecMaxPages = kPagedMaxPages;
static_assert(kPagedMaxChunk == (uint64_t)kRotMaxBlocks * 256u, "the paged form ends where the index-fed decoder does");
constexpr uint32_t dec_pages_at(uint32_t R) { return dec_lds_bytes(R); }
constexpr uint32_t dec_lds_bytes_paged(uint32_t R) { return dec_lds_bytes(R) + 8u * kDecMaxPages; }
static_assert(dec_lds_bytes_paged(12) <= 160u * 1024u, "LDS budget of the paged decoder");
// SPLIT encoder (round 5): eight CHAIN waves (hash, exchange, signatures, commit) and eight EMIT waves.  Behind the staging area:
// the quad ring — three rounds of 16 blocks x 64 lanes (or four of 12), filled by the emit waves (which load the input and keep the quads for the emit),
// drained by the chain waves —, its words {ready[3], -, freed[3], -}, and one mail box per pair of waves: the signatures of a committed round
// (lane j's 8 bytes), then {sequence word, stream position, -, -}, then {taken, -, -, -}
constexpr uint32_t kEncRing = kEncLds, kEncRingSync = kEncRing + 12288u, kEncMbox = kEncRingSync + 32u, kMboxBytes = 160u;
constexpr uint32_t ring_slots(uint32_t R) { return 12288u / (R * 256u); }          // 3 rounds of 16 blocks, 4 of 12
constexpr uint32_t kEncLdsSplit = kEncMbox + 8u * kMboxBytes;
static_assert(kEncLdsSplit <= 160u * 1024u && kEncRing % 16u == 0, "LDS budget of the split encoder");
static_assert(kEncLds <= 160u * 1024u && dec_lds_bytes(8) <= 160u * 1024u && dec_lds_bytes(12) <= 160u * 1024u && dec_lds_bytes(16) <= 160u * 1024u, "LDS budget");
static_assert(dec_zmap_in_lds(12) && !dec_zmap_in_lds(8), "where the decoder's zero-entry map lives");


typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x2 u32x2_u __attribute__((aligned(1)));

// token polls: every lane reads the same address (broadcast), the caller takes lane 0's copy
__device__ __forceinline__ u32x2 lds_peek2(uint32_t addr) {
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ u32x4 lds_peek4(uint32_t addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
// two consecutive 16-byte lines in one round trip (the D line and the O line of the sync block)
__device__ __forceinline__ void lds_peek4x2(uint32_t addr, u32x4& a, u32x4& b) {
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_peek1(uint32_t addr) {
    uint32_t v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ uint32_t rlane_u(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ uint32_t rlane(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ void lds_poke(uint32_t addr, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
// the same at a compile-time offset from a base register (16-bit field): R addresses from ONE register — per-lane addresses that differ by constants would
// otherwise be hoisted out of the round loop one register each, rare paths included, and sit on the common path's register budget
#define DENSITY_LDS_POKE_AT(base, off, v) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(base), "v"(v), "n"(off) : "memory")
__device__ __forceinline__ void lds_poke2(uint32_t addr, uint32_t a, uint32_t b) {
    const u32x2 v = {a, b};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// the work-group barrier of the (rare) abort protocol and of the kernel's end: own LDS traffic retired first
__device__ __forceinline__ void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The critical section of a round: 8 ordered exchanges from prepared registers, then the token for the next round written
// behind them (lane 0 writes the token word, the other lanes a sink, so the store has no bank conflict), then the answers.
// One asm statement: the answers are valid when it ends, nothing the compiler does can touch a register still in flight.
#define DENSITY_ROT_XCHG8                                   \
    "ds_mskor_rtn_b32 %0, %8, %16, %24\n\t"                 \
    "ds_mskor_rtn_b32 %1, %9, %17, %25\n\t"                 \
    "ds_mskor_rtn_b32 %2, %10, %18, %26\n\t"                \
    "ds_mskor_rtn_b32 %3, %11, %19, %27\n\t"                \
    "ds_mskor_rtn_b32 %4, %12, %20, %28\n\t"                \
    "ds_mskor_rtn_b32 %5, %13, %21, %29\n\t"                \
    "ds_mskor_rtn_b32 %6, %14, %22, %30\n\t"                \
    "ds_mskor_rtn_b32 %7, %15, %23, %31\n\t"
#define DENSITY_ROT_OPERANDS(ret, addr, mask, val, tokaddr, tokval)                                                                     \
    : "=&v"(ret[0]), "=&v"(ret[1]), "=&v"(ret[2]), "=&v"(ret[3]), "=&v"(ret[4]), "=&v"(ret[5]), "=&v"(ret[6]), "=&v"(ret[7])              \
    : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(addr[4]), "v"(addr[5]), "v"(addr[6]), "v"(addr[7]),                     \
      "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]), "v"(mask[7]),                     \
      "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]), "v"(tokaddr), "v"(tokval)  \
    : "memory"
__device__ __forceinline__ void exchange_round(uint32_t (&ret)[kR], const uint32_t (&addr)[kR], const uint32_t (&mask)[kR], const uint32_t (&val)[kR],
                                               uint32_t tokaddr, uint32_t tokval, bool token_after_answers) {
    if (!token_after_answers) {
        asm volatile(DENSITY_ROT_XCHG8 "ds_write_b32 %32, %33\n\ts_waitcnt lgkmcnt(0)" DENSITY_ROT_OPERANDS(ret, addr, mask, val, tokaddr, tokval));
    } else {   // tuning / fall-back form: the token leaves only after the last answer is back
        asm volatile(DENSITY_ROT_XCHG8 "s_waitcnt lgkmcnt(0)\n\tds_write_b32 %32, %33" DENSITY_ROT_OPERANDS(ret, addr, mask, val, tokaddr, tokval));
    }
}

// The same with the answer returned in place of the address (one register per block less) for rounds of 8 or 16 blocks.
#define DENSITY_ROT_X8 \
    "ds_mskor_rtn_b32 %0, %0, %8, %16\n\t" \
    "ds_mskor_rtn_b32 %1, %1, %9, %17\n\t" \
    "ds_mskor_rtn_b32 %2, %2, %10, %18\n\t" \
    "ds_mskor_rtn_b32 %3, %3, %11, %19\n\t" \
    "ds_mskor_rtn_b32 %4, %4, %12, %20\n\t" \
    "ds_mskor_rtn_b32 %5, %5, %13, %21\n\t" \
    "ds_mskor_rtn_b32 %6, %6, %14, %22\n\t" \
    "ds_mskor_rtn_b32 %7, %7, %15, %23\n\t" \
    ""
#define DENSITY_ROT_X8_OPS : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]) \
    : "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]), "v"(mask[7]), "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]), "v"(tokaddr), "v"(tokval) : "memory"
#define DENSITY_ROT_X16 \
    "ds_mskor_rtn_b32 %0, %0, %16, %32\n\t" \
    "ds_mskor_rtn_b32 %1, %1, %17, %33\n\t" \
    "ds_mskor_rtn_b32 %2, %2, %18, %34\n\t" \
    "ds_mskor_rtn_b32 %3, %3, %19, %35\n\t" \
    "ds_mskor_rtn_b32 %4, %4, %20, %36\n\t" \
    "ds_mskor_rtn_b32 %5, %5, %21, %37\n\t" \
    "ds_mskor_rtn_b32 %6, %6, %22, %38\n\t" \
    "ds_mskor_rtn_b32 %7, %7, %23, %39\n\t" \
    "ds_mskor_rtn_b32 %8, %8, %24, %40\n\t" \
    "ds_mskor_rtn_b32 %9, %9, %25, %41\n\t" \
    "ds_mskor_rtn_b32 %10, %10, %26, %42\n\t" \
    "ds_mskor_rtn_b32 %11, %11, %27, %43\n\t" \
    "ds_mskor_rtn_b32 %12, %12, %28, %44\n\t" \
    "ds_mskor_rtn_b32 %13, %13, %29, %45\n\t" \
    "ds_mskor_rtn_b32 %14, %14, %30, %46\n\t" \
    "ds_mskor_rtn_b32 %15, %15, %31, %47\n\t" \
    ""
#define DENSITY_ROT_X16_OPS : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]), "+v"(ra[8]), "+v"(ra[9]), "+v"(ra[10]), "+v"(ra[11]), "+v"(ra[12]), "+v"(ra[13]), "+v"(ra[14]), "+v"(ra[15]) \
    : "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]), "v"(mask[7]), "v"(mask[8]), "v"(mask[9]), "v"(mask[10]), "v"(mask[11]), "v"(mask[12]), "v"(mask[13]), "v"(mask[14]), "v"(mask[15]), "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]), "v"(val[8]), "v"(val[9]), "v"(val[10]), "v"(val[11]), "v"(val[12]), "v"(val[13]), "v"(val[14]), "v"(val[15]), "v"(tokaddr), "v"(tokval) : "memory"
#define DENSITY_ROT_PF16 \
    "global_load_dword v240, %0, off offset:0\n\t" \
    "global_load_dword v241, %0, off offset:256\n\t" \
    "global_load_dword v242, %0, off offset:512\n\t" \
    "global_load_dword v243, %0, off offset:768\n\t" \
    "global_load_dword v244, %0, off offset:1024\n\t" \
    "global_load_dword v245, %0, off offset:1280\n\t" \
    "global_load_dword v246, %0, off offset:1536\n\t" \
    "global_load_dword v247, %0, off offset:1792\n\t" \
    "global_load_dword v248, %0, off offset:2048\n\t" \
    "global_load_dword v249, %0, off offset:2304\n\t" \
    "global_load_dword v250, %0, off offset:2560\n\t" \
    "global_load_dword v251, %0, off offset:2816\n\t" \
    "global_load_dword v252, %0, off offset:3072\n\t" \
    "global_load_dword v253, %0, off offset:3328\n\t" \
    "global_load_dword v254, %0, off offset:3584\n\t" \
    "global_load_dword v255, %0, off offset:3840\n\t" \
    ""
#define DENSITY_ROT_MV16 \
    "v_mov_b32 %0, v240\n\t" \
    "v_mov_b32 %1, v241\n\t" \
    "v_mov_b32 %2, v242\n\t" \
    "v_mov_b32 %3, v243\n\t" \
    "v_mov_b32 %4, v244\n\t" \
    "v_mov_b32 %5, v245\n\t" \
    "v_mov_b32 %6, v246\n\t" \
    "v_mov_b32 %7, v247\n\t" \
    "v_mov_b32 %8, v248\n\t" \
    "v_mov_b32 %9, v249\n\t" \
    "v_mov_b32 %10, v250\n\t" \
    "v_mov_b32 %11, v251\n\t" \
    "v_mov_b32 %12, v252\n\t" \
    "v_mov_b32 %13, v253\n\t" \
    "v_mov_b32 %14, v254\n\t" \
    "v_mov_b32 %15, v255\n\t" \
    ""
#define DENSITY_ROT_MV16_OUTS "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3]), "=v"(q[4]), "=v"(q[5]), "=v"(q[6]), "=v"(q[7]), "=v"(q[8]), "=v"(q[9]), "=v"(q[10]), "=v"(q[11]), "=v"(q[12]), "=v"(q[13]), "=v"(q[14]), "=v"(q[15])
#define DENSITY_ROT_STAGE16 "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255"

// the same for the 12-wave kernels (168 registers a wave): staging in v(168-R)..v167

// element j (wave-uniform, not a compile-time constant) of a register array, for the rolled loops of the rare paths: a chain of
// selects, so the array stays in registers (a dynamically indexed copy would live in scratch memory, and the compiler's waits for
// its loads would also hold the common path at the top of every round)
template <int R>
__device__ __forceinline__ uint32_t pick(const uint32_t (&a)[R], uint32_t j) {
    uint32_t v = a[0];
#pragma unroll
    for (uint32_t k = 1; k < (uint32_t)R; ++k) {
        uint32_t jj = j;
        asm volatile("" : "+s"(jj));                                              // (opaque: or the compiler turns the chain back into a table in scratch)
        v = jj == k ? a[k] : v;
    }
    return v;
}
// Next round's quads, fetched by hand: R dword loads (one 256-byte block each) the compiler does not see as memory operations, so
// it places no wait of its own between them and the stores that follow.  They land in the R highest registers of the wave
// (v(256-R)..v255, named in the statements and declared clobbered), which the compiler, allocating upwards from v0, never reaches in
// these kernels (tools/check_isa.py checks that no other instruction names them), so nothing can read or move them early.  (The
// accumulation registers would be the natural staging area, but a kernel that names one has its register file split in halves.)
// `quads_landed` waits — every load is older than the `kYounger` memory operations the
// caller guarantees to have issued since (vmcnt counts a wave's loads and stores in order) — and reads them into `q`.
template <int R, int W>
__device__ __forceinline__ void prefetch_quads(const uint8_t*) {}                // (geometries without kept quads: never called)
template <>
__device__ __forceinline__ void prefetch_quads<16, 8>(const uint8_t* p) { asm volatile(DENSITY_ROT_PF16 : : "v"(p) : "memory", DENSITY_ROT_STAGE16); }
template <int R, int W, bool kDrained>
__device__ __forceinline__ void quads_landed(uint32_t (&)[R]) {}                // (likewise)
template <>
__device__ __forceinline__ void quads_landed<16, 8, false>(uint32_t (&q)[16]) { asm volatile("s_waitcnt vmcnt(16)\n\t" DENSITY_ROT_MV16 : DENSITY_ROT_MV16_OUTS : : DENSITY_ROT_STAGE16); }
template <>
__device__ __forceinline__ void quads_landed<16, 8, true>(uint32_t (&q)[16]) { asm volatile("s_waitcnt vmcnt(0)\n\t" DENSITY_ROT_MV16 : DENSITY_ROT_MV16_OUTS : : DENSITY_ROT_STAGE16); }

#define DENSITY_ROT_X12 \
    "ds_mskor_rtn_b32 %0, %0, %12, %24\n\t" \
    "ds_mskor_rtn_b32 %1, %1, %13, %25\n\t" \
    "ds_mskor_rtn_b32 %2, %2, %14, %26\n\t" \
    "ds_mskor_rtn_b32 %3, %3, %15, %27\n\t" \
    "ds_mskor_rtn_b32 %4, %4, %16, %28\n\t" \
    "ds_mskor_rtn_b32 %5, %5, %17, %29\n\t" \
    "ds_mskor_rtn_b32 %6, %6, %18, %30\n\t" \
    "ds_mskor_rtn_b32 %7, %7, %19, %31\n\t" \
    "ds_mskor_rtn_b32 %8, %8, %20, %32\n\t" \
    "ds_mskor_rtn_b32 %9, %9, %21, %33\n\t" \
    "ds_mskor_rtn_b32 %10, %10, %22, %34\n\t" \
    "ds_mskor_rtn_b32 %11, %11, %23, %35\n\t" \
    ""
#define DENSITY_ROT_X12_OPS : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]), "+v"(ra[8]), "+v"(ra[9]), "+v"(ra[10]), "+v"(ra[11]) \
    : "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]), "v"(mask[7]), "v"(mask[8]), "v"(mask[9]), "v"(mask[10]), "v"(mask[11]), "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]), "v"(val[8]), "v"(val[9]), "v"(val[10]), "v"(val[11]), "v"(tokaddr), "v"(tokval) : "memory"

template <int R>
__device__ __forceinline__ void exchange_tied(uint32_t (&ra)[R], const uint32_t (&mask)[R], const uint32_t (&val)[R], uint32_t tokaddr, uint32_t tokval, bool token_after_answers);
template <>
__device__ __forceinline__ void exchange_tied<8>(uint32_t (&ra)[8], const uint32_t (&mask)[8], const uint32_t (&val)[8], uint32_t tokaddr, uint32_t tokval, bool token_after_answers) {
    if (!token_after_answers) asm volatile(DENSITY_ROT_X8 "ds_write_b32 %24, %25\n\ts_waitcnt lgkmcnt(0)" DENSITY_ROT_X8_OPS);
    else asm volatile(DENSITY_ROT_X8 "s_waitcnt lgkmcnt(0)\n\tds_write_b32 %24, %25" DENSITY_ROT_X8_OPS);
}
template <>
__device__ __forceinline__ void exchange_tied<12>(uint32_t (&ra)[12], const uint32_t (&mask)[12], const uint32_t (&val)[12], uint32_t tokaddr, uint32_t tokval, bool token_after_answers) {
    if (!token_after_answers) asm volatile(DENSITY_ROT_X12 "ds_write_b32 %36, %37\n\ts_waitcnt lgkmcnt(0)" DENSITY_ROT_X12_OPS);
    else asm volatile(DENSITY_ROT_X12 "s_waitcnt lgkmcnt(0)\n\tds_write_b32 %36, %37" DENSITY_ROT_X12_OPS);
}
template <>
__device__ __forceinline__ void exchange_tied<16>(uint32_t (&ra)[16], const uint32_t (&mask)[16], const uint32_t (&val)[16], uint32_t tokaddr, uint32_t tokval, bool token_after_answers) {
    if (!token_after_answers) asm volatile(DENSITY_ROT_X16 "ds_write_b32 %48, %49\n\ts_waitcnt lgkmcnt(0)" DENSITY_ROT_X16_OPS);
    else asm volatile(DENSITY_ROT_X16 "s_waitcnt lgkmcnt(0)\n\tds_write_b32 %48, %49" DENSITY_ROT_X16_OPS);
}
// The exchanges of an ORDERED round (encoder, below) in one statement: block j's is skipped if bit j of `idle` is set (a final block, a predicted
// raw copy); no token behind them.
#define DENSITY_ROT_XC16 \
    "s_bitcmp1_b32 %[idle], 0\n\ts_cbranch_scc1 .Lskip0_%=\n\tds_mskor_rtn_b32 %0, %0, %16, %32\n.Lskip0_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 1\n\ts_cbranch_scc1 .Lskip1_%=\n\tds_mskor_rtn_b32 %1, %1, %17, %33\n.Lskip1_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 2\n\ts_cbranch_scc1 .Lskip2_%=\n\tds_mskor_rtn_b32 %2, %2, %18, %34\n.Lskip2_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 3\n\ts_cbranch_scc1 .Lskip3_%=\n\tds_mskor_rtn_b32 %3, %3, %19, %35\n.Lskip3_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 4\n\ts_cbranch_scc1 .Lskip4_%=\n\tds_mskor_rtn_b32 %4, %4, %20, %36\n.Lskip4_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 5\n\ts_cbranch_scc1 .Lskip5_%=\n\tds_mskor_rtn_b32 %5, %5, %21, %37\n.Lskip5_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 6\n\ts_cbranch_scc1 .Lskip6_%=\n\tds_mskor_rtn_b32 %6, %6, %22, %38\n.Lskip6_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 7\n\ts_cbranch_scc1 .Lskip7_%=\n\tds_mskor_rtn_b32 %7, %7, %23, %39\n.Lskip7_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 8\n\ts_cbranch_scc1 .Lskip8_%=\n\tds_mskor_rtn_b32 %8, %8, %24, %40\n.Lskip8_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 9\n\ts_cbranch_scc1 .Lskip9_%=\n\tds_mskor_rtn_b32 %9, %9, %25, %41\n.Lskip9_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 10\n\ts_cbranch_scc1 .Lskip10_%=\n\tds_mskor_rtn_b32 %10, %10, %26, %42\n.Lskip10_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 11\n\ts_cbranch_scc1 .Lskip11_%=\n\tds_mskor_rtn_b32 %11, %11, %27, %43\n.Lskip11_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 12\n\ts_cbranch_scc1 .Lskip12_%=\n\tds_mskor_rtn_b32 %12, %12, %28, %44\n.Lskip12_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 13\n\ts_cbranch_scc1 .Lskip13_%=\n\tds_mskor_rtn_b32 %13, %13, %29, %45\n.Lskip13_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 14\n\ts_cbranch_scc1 .Lskip14_%=\n\tds_mskor_rtn_b32 %14, %14, %30, %46\n.Lskip14_%=:\n\t" \
    "s_bitcmp1_b32 %[idle], 15\n\ts_cbranch_scc1 .Lskip15_%=\n\tds_mskor_rtn_b32 %15, %15, %31, %47\n.Lskip15_%=:\n\t"
template <int R>
__device__ __forceinline__ void exchange_some(uint32_t (&ra)[R], const uint32_t (&mask)[R], const uint32_t (&val)[R], uint32_t idle) {
    // (one statement per block WITH its wait: an answer still in flight at the end of a conditional statement would be the compiler's to copy)
#pragma unroll
    for (uint32_t j = 0; j < (uint32_t)R; ++j) {
        if (!((idle >> j) & 1u)) asm volatile("ds_mskor_rtn_b32 %0, %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "+v"(ra[j]) : "v"(mask[j]), "v"(val[j]) : "memory");
    }
}
template <>
__device__ __forceinline__ void exchange_some<16>(uint32_t (&ra)[16], const uint32_t (&mask)[16], const uint32_t (&val)[16], uint32_t idle) {
    asm volatile(DENSITY_ROT_XC16 "s_waitcnt lgkmcnt(0)"
                 : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]), "+v"(ra[8]), "+v"(ra[9]), "+v"(ra[10]), "+v"(ra[11]), "+v"(ra[12]), "+v"(ra[13]), "+v"(ra[14]), "+v"(ra[15])
                 : "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]), "v"(mask[7]), "v"(mask[8]), "v"(mask[9]), "v"(mask[10]), "v"(mask[11]), "v"(mask[12]), "v"(mask[13]), "v"(mask[14]), "v"(mask[15]), "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]), "v"(val[8]), "v"(val[9]), "v"(val[10]), "v"(val[11]), "v"(val[12]), "v"(val[13]), "v"(val[14]), "v"(val[15]), [idle] "s"(idle)
                 : "memory", "scc");
}
// the same for a round that RUNS AHEAD (below): the dictionary token — its run-ahead words {state predicted for the next round, 1}, then the token
// word — is written behind the exchanges in the same statement, by lane 0 alone, before their answers are waited for
template <int R>
__device__ __forceinline__ void exchange_some_ahead(uint32_t (&ra)[R], const uint32_t (&mask)[R], const uint32_t (&val)[R], uint32_t idle, uint32_t dline, uint32_t state, uint32_t token) {
    exchange_some<R>(ra, mask, val, idle);
    if ((threadIdx.x & 63u) == 0) { lds_poke2(dline + 8u, state, 1u); lds_poke(dline, token); }
}
template <>
__device__ __forceinline__ void exchange_some_ahead<16>(uint32_t (&ra)[16], const uint32_t (&mask)[16], const uint32_t (&val)[16], uint32_t idle, uint32_t dline, uint32_t state, uint32_t token) {
    const u32x2 words = {state, 1u};
    const uint32_t dline2 = dline + 8u;
    asm volatile(DENSITY_ROT_XC16
                 "s_mov_b64 exec, 1\n\t"
                 "ds_write_b64 %[d2], %[w]\n\t"
                 "ds_write_b32 %[d], %[t]\n\t"
                 "s_mov_b64 exec, -1\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]), "+v"(ra[8]), "+v"(ra[9]), "+v"(ra[10]), "+v"(ra[11]), "+v"(ra[12]), "+v"(ra[13]), "+v"(ra[14]), "+v"(ra[15])
                 : "v"(mask[0]), "v"(mask[1]), "v"(mask[2]), "v"(mask[3]), "v"(mask[4]), "v"(mask[5]), "v"(mask[6]), "v"(mask[7]), "v"(mask[8]), "v"(mask[9]), "v"(mask[10]), "v"(mask[11]), "v"(mask[12]), "v"(mask[13]), "v"(mask[14]), "v"(mask[15]), "v"(val[0]), "v"(val[1]), "v"(val[2]), "v"(val[3]), "v"(val[4]), "v"(val[5]), "v"(val[6]), "v"(val[7]), "v"(val[8]), "v"(val[9]), "v"(val[10]), "v"(val[11]), "v"(val[12]), "v"(val[13]), "v"(val[14]), "v"(val[15]), [idle] "s"(idle),
                   [d2] "v"(dline2), [w] "v"(words), [d] "v"(dline), [t] "v"(token)
                 : "memory", "scc");
}
// (keeps a set of operands from being scheduled past this point, i.e. into the critical section behind the token wait)
template <int R>
__device__ __forceinline__ void pin_operands(uint32_t (&ra)[R], uint32_t (&mask)[R], uint32_t (&val)[R]) {
#pragma unroll
    for (int j = 0; j < R; ++j) asm volatile("" : "+v"(ra[j]), "+v"(mask[j]), "+v"(val[j]));
}
__device__ __forceinline__ uint32_t exchange_block(uint32_t addr, uint32_t mask, uint32_t val) {
    uint32_t ret;
    asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(ret) : "v"(addr), "v"(mask), "v"(val) : "memory");
    return ret;
}

// Commit payload {stream position (32 bits: the launcher bounds the chunk), FSM state}: penalty [0,8) | start-1 [8,16) | prev [16] | counter&15 [17,21)
// (protection_state.rs: copy_penalty, copy_penalty_start are u8, only counter & 0xf is ever tested).  Calm state with start == 1: low 17 bits 0.
__device__ __forceinline__ uint32_t pack_guard(const Guard& g) { return (g.penalty & 0xffu) | (((g.start - 1u) & 0xffu) << 8) | ((g.prev & 1u) << 16) | ((g.counter & 15u) << 17); }
__device__ __forceinline__ Guard unpack_guard(uint32_t w) {
    Guard g;
    g.penalty = w & 0xffu; g.start = ((w >> 8) & 0xffu) + 1u; g.prev = (w >> 16) & 1u; g.counter = (w >> 17) & 15u;
    return g;
}

// The FSM of an ORDERED round (below), protection_state.rs:19-47 on packed states (pack_guard), wave-uniform (scalar registers).
// fsm_verify: blocks j0..R-1 from the state `st` in front of block j0; every block's raw-copy status must be bit j of `raw_old` (what the last
// exchange assumed), a coded block is incompressible iff bit j of `inc` (its signature).  Returns the first block that is not what was assumed
// (R: none; then `state` is the state behind the round), the state in front of it and whether it is a raw copy — a raw copy where none was
// expected starts an incompressible stretch, a coded block where a copy was expected ends one.
template <int R>
__device__ __forceinline__ uint32_t fsm_verify(uint32_t st, uint32_t j0, uint32_t inc, uint32_t raw_old, uint32_t& state, uint32_t& is_copy) {
    Guard g = unpack_guard(st);
    is_copy = 0;
    uint32_t j = j0;
#pragma nounroll
    for (; j < (uint32_t)R; ++j) {
        const uint32_t copy = g.penalty != 0 ? 1u : 0u;                            // (what block_is_copy will say: the halving in it does not change that)
        if (copy != ((raw_old >> j) & 1u)) { is_copy = copy; break; }
        (void)g.block_is_copy();                                                   // codec.rs:35
        if (copy) g.decay();                                                       // codec.rs:36-37
        else g.update((inc >> j) & 1u);                                            // codec.rs:68
    }
    state = pack_guard(g);
    return j;
}
// fsm_predict: the raw-copy blocks among j0..R-1 (`raw`: bits below j0 as given) and the state behind the round if every coded block from j0
// on is incompressible (`storm`) or none is — the FSM is then a function of its state alone, taken run by run instead of block by block:
// a run of raw copies (penalty blocks, protection_state.rs:30-35), one coded block that triggers the next (:38-47), ...; `start` is halved at
// the one block of the stretch whose counter is a multiple of 16 (:19-27: before that block's own decay or trigger).
template <int R>
__device__ __forceinline__ void fsm_predict(uint32_t st, uint32_t j0, uint32_t storm, uint32_t raw_below, uint32_t& raw, uint32_t& end_state) {
    uint32_t p = st & 0xffu, s = ((st >> 8) & 0xffu) + 1u, v = (st >> 16) & 1u, c = (st >> 17) & 15u;
    uint32_t j = j0;
    raw = raw_below & ((1u << j0) - 1u);
#pragma nounroll
    while (j < (uint32_t)R) {
        const uint32_t kh = (16u - c) & 15u;                                       // blocks in front of the next halving point
        if (p) {                                                                   // a run of raw copies
            uint32_t L = (uint32_t)R - j;
            L = p < L ? p : L;
            raw |= ((1u << L) - 1u) << j;
            if (kh < L && s > 1u) s >>= 1;
            c = (c + L) & 15u; p -= L; j += L;
            if (p == 0) s = (s + 1u) & 0xffu;
        } else if (storm) {                                                        // one coded, incompressible block
            if (kh == 0 && s > 1u) s >>= 1;
            c = (c + 1u) & 15u; ++j;
            if (v) p = s;
            v = 1;
        } else {                                                                   // coded blocks to the round's end, none of them incompressible
            const uint32_t n = (uint32_t)R - j;
            if (kh < n && s > 1u) s >>= 1;
            c = (c + n) & 15u; j = (uint32_t)R; v = 0;
        }
    }
    end_state = (p & 0xffu) | (((s - 1u) & 0xffu) << 8) | (v << 16) | (c << 17);
}

// ordered rounds without unrest before the encoder speculates ACROSS rounds again: 2, and 2 more (at most 7) with every abort the chunk has seen —
// its count, up to 3, rides in bits 24..25 of the commit payload.  (Same box, round 5: leaving after 2 quiet rounds is as fast as round 4's library
// on text at 1 GiB and 6 % faster at 10 MB — the cold start of every 64 KiB chunk is ordered rounds now, not block-by-block walks —, waiting for
// 6 always costs text 1.3 % / 10 % / 13 % at 1 GiB / 100 MB / 10 MB; data that flips every few KiB aborts a work-group per flip when it leaves early.)
__device__ __forceinline__ uint32_t quiet_rounds(uint32_t P1) { const uint32_t q = 2u + 2u * ((P1 >> 24) & 3u); return q > 7u ? 7u : q; }
constexpr uint32_t kStormRounds = 3;                                              // ordered rounds of one unbroken incompressible stretch before the dictionary token runs ahead of the commit
constexpr uint32_t kSpinLimit = 1u << 22, kPoison = 0xfffffffeu, kErrWatchdog = 16u;
__device__ __forceinline__ void wave_exit() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_endpgm" ::: "memory"); }
__device__ __forceinline__ void watchdog(uint32_t& spins, uint32_t sync_base, uint32_t* err, uint32_t lane) {
    if (__builtin_expect(++spins > kSpinLimit, 0)) {
        if (lane == 0) {
            if (err) atomicOr(err, kErrWatchdog);
            lds_poke(sync_base + kSyD, kPoison); lds_poke(sync_base + kSyD + 4, kPoison);
            lds_poke(sync_base + kSyO, kPoison); lds_poke(sync_base + kSyO + 4, kPoison);
            lds_poke(sync_base + kSyZ, kPoison);
        }
        wave_exit();
    }
}

// optional cycle accounting (DENSITY_HIP_PROF=1): work-group 0 reports, per wave, the cycles spent in each phase of its iterations
constexpr uint32_t kProfRounds = 2048;
template <bool ON>
struct PhaseClock;
template <>
struct PhaseClock<false> {
    __device__ __forceinline__ explicit PhaseClock(uint64_t*) {}
    __device__ __forceinline__ void start() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void stamp(uint32_t, uint32_t, uint32_t) {}
    __device__ __forceinline__ void note(uint32_t, uint32_t, uint32_t) {}
    __device__ __forceinline__ void flush(uint32_t, uint32_t) {}
    __device__ __forceinline__ void count(int, uint32_t) {}
};
template <>
struct PhaseClock<true> {
    uint64_t* out; uint64_t t0 = 0; uint32_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // (32-bit sums: the profiling instances are as short of registers as the shipped ones)
    __device__ __forceinline__ explicit PhaseClock(uint64_t* o) : out(o) {}
    __device__ __forceinline__ void start() { if (out) t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void mark(int k) { if (out) { const uint64_t t = __builtin_readcyclecounter(); ph[k] += (uint32_t)(t - t0); t0 = t; } }
    // per-round time stamps of the D chain (rounds < kProfRounds): 0 = started polling, 1 = token seen, 2 = exchanges + token done, 3 = round finished
    __device__ __forceinline__ void stamp(uint32_t r, uint32_t what, uint32_t lane) {
        if (out && r < kProfRounds && lane == 0) out[128 + 4 * r + what] = __builtin_readcyclecounter();
    }
    // a mark on a round (bit 0: it went through the zero-entry path)
    __device__ __forceinline__ void note(uint32_t r, uint32_t v, uint32_t lane) {
        if (out && r < kProfRounds && lane == 0) out[128 + 4 * kProfRounds + r] = v;
    }
    __device__ __forceinline__ void flush(uint32_t wave, uint32_t lane) { if (out && lane == 0) for (int k = 0; k < 8; ++k) out[8 * wave + k] = ph[k]; }
    // event counters of work-group 0 (encoder: 0 fast rounds committed, 1 ordered rounds that held, 2 ordered rounds taken back, 3 rounds walked in order, 4 aborts raised, 5 ordered rounds that ran ahead)
    __device__ __forceinline__ void count(int k, uint32_t lane) { if (out && lane == 0) atomicAdd(reinterpret_cast<unsigned long long*>(out + 128 + 5 * kProfRounds + k), 1ull); }
};

// Waiting for a token.  The wave whose turn is next (or next but one) polls in a loop of five instructions; waves further away sleep
// for most of the distance first (a round hand-off takes a few hundred cycles), so that the LDS and the issue slots stay with
// the waves that work.
__device__ __forceinline__ void backoff(uint32_t dist) {
    if (dist <= 2) return;
    if (dist > 8) __builtin_amdgcn_s_sleep(24);
    else if (dist > 4) __builtin_amdgcn_s_sleep(8);
    else __builtin_amdgcn_s_sleep(3);
}
// the same between ORDERED rounds, whose hand-offs take a thousand cycles and more: only the next wave polls, the others nap for most of their distance
// (seven waves polling two 16-byte lines each kept the LDS busy enough to triple the round trip of the holder's own look-ups)
__device__ __forceinline__ void backoff_ordered(uint32_t dist) {
    if (dist <= 1) return;
    if (dist > 4) __builtin_amdgcn_s_sleep(40);
    else if (dist > 2) __builtin_amdgcn_s_sleep(20);
    else __builtin_amdgcn_s_sleep(8);
}
// up to `tries` back-to-back polls of one token word for one value
// (Written out: the compiled loop kept its counter in a vector register and took ten instructions per poll — every one of them between
// "the token is there" and the first exchange of the new holder.  Five here: read, wait, lane 0's copy, compare, branch; the count-down is
// issued while the read is in flight.)
__device__ __forceinline__ bool poll_word(uint32_t addr, uint32_t want, uint32_t tries) {
    uint32_t v, seen;
    asm volatile(
        "1:\n\t"
        "ds_read_b32 %[v], %[a]\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %[s], %[v]\n\t"
        "s_cmp_eq_u32 %[s], %[w]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "2:"
        : [v] "=&v"(v), [s] "=&s"(seen), [n] "+s"(tries)
        : [a] "v"(addr), [w] "s"(want)
        : "scc", "memory");
    return seen == want;
}

// the same on a whole 16-byte line whose first word is the token: the line as it was when the token matched, or as last seen (four one-word reads in
// one round trip: a tuple register cannot be named word by word in an asm statement)
__device__ __forceinline__ bool poll_line(uint32_t addr, uint32_t want, uint32_t tries, u32x4& line) {
    uint32_t seen, w0, w1, w2, w3;
    asm volatile(
        "1:\n\t"
        "ds_read_b32 %[w0], %[a]\n\t"
        "ds_read_b32 %[w1], %[a] offset:4\n\t"
        "ds_read_b32 %[w2], %[a] offset:8\n\t"
        "ds_read_b32 %[w3], %[a] offset:12\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_readfirstlane_b32 %[s], %[w0]\n\t"
        "s_cmp_eq_u32 %[s], %[w]\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "2:"
        : [w0] "=&v"(w0), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [s] "=&s"(seen), [n] "+s"(tries)
        : [a] "v"(addr), [w] "s"(want)
        : "scc", "memory");
    line = u32x4{w0, w1, w2, w3};
    return seen == want;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// encode: Codec::encode / encode_block (codec/codec.rs:34-80), Chameleon::encode_quad (chameleon.rs:88-100)
// ---------------------------------------------------------------------------------------------------------------
template <int R, int W, bool kProf, bool KEEP = (W == 8), bool EARLY = false, bool PAGED = false, bool SPLIT = false>
__global__ __launch_bounds__(SPLIT ? 2 * W * 64 : W * 64) void chameleon_encode_rot(const uint8_t* __restrict__ in, uint64_t total, uint64_t chunk_bytes,
                                                                   uint8_t* __restrict__ out, uint64_t out_stride, uint64_t* __restrict__ sizes,
                                                                   uint8_t* __restrict__ index, uint32_t* __restrict__ err, SegArgs seg,
                                                                   uint64_t* __restrict__ prof) {
    static_assert((R == 8 || R == 12 || R == 16) && (W == 8 || W == 12 || W == 16), "round = 8, 12 or 16 blocks; 8, 12 or 16 waves");
    static_assert(!KEEP || (R == 16 && W == 8), "kept quads: staging registers exist for rounds of 16 on 8 waves (12 waves: the compiler needs them itself, DESIGN.md 4.3)");
    static_assert(!SPLIT || ((R == 16 || R == 12) && W == 8 && !KEEP && !EARLY), "split encoder: rounds of 12 or 16, 8 chain + 8 emit waves, quads from the ring (no hand-fetched loads)");
    constexpr uint32_t kRingSlots = ring_slots(R), kSlotBytes = R * 256u;
    constexpr uint32_t kThreads = SPLIT ? 2u * W * 64u : W * 64u;
    // (rounds of 16 on 16 waves fit the 128 registers a wave then has because nothing but the exchange operands is kept across the wait for the
    // dictionary token: the quads themselves are loaded again — from L2 — once the exchanges are out)
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint64_t chunk = blockIdx.x;
    PhaseClock<kProf> clk(blockIdx.x == 0 ? prof : nullptr);   // phases: 0 hash, 1 D wait, 2 exchange, 3 signatures, 4 O wait + commit, 5 load wait, 6 emit, 7 in-order rounds
    const uint8_t* src = in + chunk * chunk_bytes;
    const uint64_t len = (total - chunk * chunk_bytes) < chunk_bytes ? (total - chunk * chunk_bytes) : chunk_bytes;
    // PAGED (round 5): no slot per chunk — `out` is page 0 of the container, stream positions are absolute offsets from it, and the stream moves to
    // a fresh page (one shared counter) whenever a round's records would not fit the rest of its page (below: page_place)
    uint8_t* dst = PAGED ? out : out + chunk * out_stride;
    uint8_t* idx = index ? index + chunk * (chunk_bytes / kBlock) : nullptr;     // this chunk's slice of the block index
    const uint32_t nfull = (uint32_t)(len / kBlock);                              // whole blocks (the launcher bounds len)
    const uint32_t nrounds = nfull / R;
    uint32_t* dir = PAGED ? seg.page_dir + chunk * seg.page_dir_words : nullptr;   // this chunk's page directory                                            // whole rounds: these rotate; the rest (< R blocks + a ragged one) is the epilogue
    // the table sits at LDS address 0 (this kernel has no static LDS): slot addresses need no base
    const uint32_t sy = kEncSync;
    const ZmapLds zmap{kEncZmap};

    {   // fresh state per chunk (chameleon.rs:45-48): zero table, zero-entry map, tokens: round 0 in slow mode, nothing committed
        // (a segment of a longer stream — SegArgs — starts from the dictionary image and FSM state it is given instead, and in
        // speculation mode if its predecessor ended calm)
        uint4* p = reinterpret_cast<uint4*>(smem);
        const uint4 z = make_uint4(0, 0, 0, 0);
        const uint4* image = seg.init_images ? reinterpret_cast<const uint4*>(seg.init_images + chunk * kSegImageBytes) : nullptr;
        for (uint32_t i = threadIdx.x; i < (kTableBytes + kZmapBytes) / 16; i += kThreads) p[i] = image ? image[i] : z;
        if (threadIdx.x == 0) {
            const uint32_t g0 = seg.init_guard ? seg.init_guard[chunk] : pack_guard(Guard{});
            uint32_t pos0 = 0;
            if (PAGED) {                                                          // this chunk's first page and its spare: {page base, stream bytes in earlier pages, spare page, pages so far}
                const uint32_t pg = atomicAdd(seg.page_counter, 2u);
                if (pg + 2u > seg.page_limit && err) atomicOr(err, 2u);           // (cannot happen: the launcher's bound is every chunk's worst case)
                pos0 = pg << kPageShift;
                *reinterpret_cast<uint4*>(smem + kEncSync + kSyPage) = make_uint4(pos0, 0u, pg + 1u, 1u);
                *reinterpret_cast<uint4*>(dir + 4) = make_uint4(pg, 0u, 0u, 0u);
            }
            *reinterpret_cast<uint4*>(smem + kEncSync + kSyD) = make_uint4((g0 >> 31) ? 0u : 1u, kNone, 0u, 0u);
            *reinterpret_cast<uint4*>(smem + kEncSync + kSyO) = make_uint4(0u, kNone, pos0, g0 & 0x7fffffffu);
            if (lds_addr(smem) != 0 && err) atomicOr(err, kErrWatchdog);           // (cannot happen: see above)
        }
        if (threadIdx.x < kMemoEntries) *reinterpret_cast<uint4*>(smem + kEncSync + kSyMemo + 16u * threadIdx.x) = make_uint4(kNone, 0u, 0u, 0u);
        if (SPLIT) {                                                              // ring words and mail boxes: nothing filled, nothing drained, nothing posted, nothing taken
            if (threadIdx.x < 2) *reinterpret_cast<uint4*>(smem + kEncRingSync + 16u * threadIdx.x) = z;
            if (threadIdx.x >= 64 && threadIdx.x < 64 + 16) *reinterpret_cast<uint4*>(smem + kEncMbox + kMboxBytes * ((threadIdx.x - 64) >> 1) + 128u + 16u * (threadIdx.x & 1u)) = z;
        }
    }
    __syncthreads();

    // 8 waves have 256 registers each: the quads stay in registers across the waits and the next round's are fetched a round ahead;
    // 12 and 16 waves load them again instead
    constexpr bool kKeepQuads = KEEP;                                             // (split: a chain wave takes its quads from the ring and keeps them only up to the exchange operands — the rare paths that want them again load them from L2, like the 12- and 16-wave geometries)
    uint32_t cur_round = 0;                                                       // (split: the round whose quads such a path loads)
    constexpr bool kKeepHash = KEEP && W == 8;                                    // (12 waves have 168 registers each: the hash product is made again for the emit)
    uint32_t q[R], hp[R];                                                         // hp: the quads' hash products (kept with them)
#pragma unroll
    for (uint32_t j = 0; j < R; ++j) hp[j] = 0;
    auto load_round = [&](uint32_t (&d)[R], uint32_t r) {
        if (r < nrounds) {
            const uint8_t* p = src + (uint64_t)r * (R * kBlock);
#pragma unroll
            for (uint32_t j = 0; j < R; ++j) d[j] = *reinterpret_cast<const uint32_t*>(p + j * kBlock + 4u * lane);
        }
    };
    // (split, rare paths of a chain wave: the round's quads again, unconditionally — a guarded load would keep the old values alive across the common path)
    // Every such path loads into an array of ITS OWN (one merged with `q` would have the compiler keep two sets of quads alive in the common path).
    auto reload_quads = [&](uint32_t (&t)[R], uint32_t r) {
        const uint8_t* p = src + (uint64_t)r * (R * kBlock);
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) t[j] = *reinterpret_cast<const uint32_t*>(p + j * kBlock + 4u * lane);
    };
    // quad -> exchange operands {dword address, half mask, entry << 16*half} (chameleon.rs:89, chameleon_dev.hpp)
    auto operands = [&](uint32_t qv, uint32_t& a, uint32_t& m, uint32_t& v, uint32_t* keep = nullptr) {
        const uint32_t P = qv * kHashMul;
        if (keep) *keep = P;
        const uint32_t sh = (P >> 12) & 16u;                                      // (h & 1) << 4
        a = (P >> 15) & 0x1fffcu;                                                 // (h >> 1) << 2
        m = 0xffffu << sh;
        v = stored_entry(qv, P) << sh;
    };
    // one record (codec.rs:39-67, io/write_buffer.rs) or raw block (codec.rs:35-37) to its place in the stream
    auto emit_block = [&](uint8_t* rec, uint32_t qv, uint64_t sg, bool raw) {
        if (raw) {
            st32u(rec + 4u * lane, qv);
        } else {
            const uint32_t off = kSig + 4u * lane - 2u * mbcnt64(sg);
            if (lane < 2) st32u(rec + 4u * lane, lane ? (uint32_t)(sg >> 32) : (uint32_t)sg);   // codec.rs:24-26
            if ((sg >> lane) & 1ull) st16u(rec + off, (qv * kHashMul) >> 16); else st32u(rec + off, qv);
        }
    };
    // one block in order: FSM, then either a raw copy or the dictionary step with the zero-entry map (slow rounds, epilogue)
    // a register array parked in the staging area (array 0 or 1), and element j (wave-uniform, not a compile-time constant) of it:
    // what the rolled loops of the rare paths index instead of registers (a dynamically indexed register array would live in scratch
    // memory, whose loads the compiler waits for at the top of every round, common path included)
    auto park = [&](uint32_t which, const uint32_t (&a)[R]) {
        uint32_t base = kEncStage + which * 4096u + 4u * lane;
        asm volatile("" : "+v"(base));                                            // (made here, on the rare path: not an invariant of the round loop)
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) DENSITY_LDS_POKE_AT(base, j * 256u, a[j]);
    };
    auto parked = [&](uint32_t which, uint32_t j) -> uint32_t { return lds_peek1(kEncStage + which * 4096u + j * 256u + 4u * lane); };
    auto block_in_order = [&](Guard& g, uint32_t qv, uint32_t& a, uint32_t m, uint32_t v, uint64_t& sg, bool& raw) {
        sg = 0;
        raw = g.block_is_copy();                                                  // codec.rs:35
        if (raw) { g.decay(); return; }
        const uint32_t old = exchange_block(a, m, v);
        a = old;                                                                  // (like the fast path: the answer replaces the address)
        const bool susp = v == 0 && qv != 0;                                       // stored entry 0 outside slot 0 (entry 0 in slot 0 is the zero quad)
        const uint32_t zbit = zmap_claim_in_order(zmap, susp, (qv * kHashMul) >> 16, lane);
        sg = ballot64(((old ^ v) & m) == 0 && (!susp || zbit));                    // chameleon.rs:90-99
        g.update((uint32_t)__builtin_popcountll(sg) <= 4u);                        // codec.rs:68: 8 + 256 - 2*hits >= 256
    };

    uint32_t ra[R], mask[R], val[R];                                              // per block: address, then (after the exchange) the answer; half mask; entry
#pragma unroll
    for (uint32_t j = 0; j < R; ++j) { q[j] = 0; ra[j] = 0; mask[j] = 0; val[j] = 0; }
    // The records of a round without raw blocks, straight-line: the signatures and the index bytes leave from lanes 0..R-1 in one
    // store each (lane j: record j, offsets by a DPP prefix over the record lengths); per block the MAP lanes store the 2-byte slot
    // index (the upper half of the hash product), the PLAIN lanes the quad, through an SGPR base (io/write_buffer.rs:13-27).
    const uint32_t minus_2lane = 0u - 2u * lane;
    auto emit_round_coded = [&](const uint32_t (&qq)[R], uint32_t pos0, uint8_t* idxp, uint32_t slo, uint32_t shi) {
        const uint32_t nhv = (uint32_t)(__builtin_popcount(slo) + __builtin_popcount(shi));
        const uint32_t lenv = kSig + kBlock - 2u * nhv;
        uint32_t incl = lenv;                                                                 // prefix within a row of 16 lanes
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
        if (R > 8) incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
        const uint32_t posv = pos0 + incl - lenv;                                                 // lane j: where record j starts
        if (lane < R) {
            *reinterpret_cast<u32x2_u*>(dst + posv) = u32x2{slo, shi};                             // codec.rs:24-26
            if (idxp) idxp[lane] = (uint8_t)nhv;
        }
        const uint32_t itemsv = posv + kSig;                                                      // lane j: where record j's items start
        // ONE store per record: every lane writes four bytes at its item's place — a PLAIN lane its quad, a MAP lane its 16-bit hash and, behind
        // it, the two bytes that FOLLOW its item in the stream: the first two of the next lane's item, or (lane 63) of the next record's
        // signature.  Neighbours then write the same bytes twice, with the same values.  (A second, 2-byte store for the MAP lanes cost the
        // texture path as much as the first: the encoder was 13 % faster without it.)  Only the round's last record, whose successor is
        // another wave's, keeps two masked stores.
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            const uint64_t sg = ((uint64_t)rlane_u(shi, (int)j) << 32) | rlane_u(slo, (int)j);
            const uint64_t plain = ~sg;
            const uint32_t pos = rlane_u(itemsv, (int)j);                         // (one read instead of a scalar running sum: popcount, shift, subtract, add)
            // the item's place: 4*lane - 2*(MAP lanes below) from the record's items on — the signature itself is the mask that is counted (no
            // complement to make), the count seeded with -2*lane, times -2 and added in one instruction
            const uint32_t P = kKeepHash ? hp[j] : qq[j] * kHashMul;              // (the hash is the MAP item: chameleon.rs:92)
            if (j + 1 < R) {
                const uint32_t nsig = rlane_u(slo, (int)j + 1);                    // the next record's first bytes: its signature's low word
                uint32_t val, off;
                asm volatile(
                    "s_mov_b64 vcc, %[sg]\n\t"
                    "v_cndmask_b32_sdwa %[v], %[q], %[P], vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"   // an item's first two bytes: MAP the hash, PLAIN the quad's low half
                    "v_mbcnt_lo_u32_b32 %[o], %[sl], %[ln]\n\t"                        // MAP lanes below - 2 * lane (the count seeded with -2 * lane) ...
                    "v_mbcnt_hi_u32_b32 %[o], %[sh], %[o]\n\t"                        // (two instructions between the select and the lane shift that reads it: the wait states a DPP source needs)
                    "v_mov_b32_dpp %[v], %[v] wave_shl:1 row_mask:0xf bank_mask:0xf\n\t"   // ... of the NEXT lane's item (lane 63: replaced below)
                    "v_mad_i32_i24 %[o], %[o], -2, %[pos]\n\t"                        // ... times -2, from the record's items on: 4 * lane - 2 * (MAP lanes below)
                    "v_writelane_b32 %[v], %[ns], 63\n\t"
                    "v_perm_b32 %[v], %[v], %[P], %[sel]\n\t"                         // hash | following bytes << 16
                    "v_cndmask_b32 %[v], %[q], %[v], vcc\n\t"                         // PLAIN lanes: the quad
                    "global_store_dword %[o], %[v], %[dst]"
                    : [v] "=&v"(val), [o] "=&v"(off)
                    : [P] "v"(P), [q] "v"(qq[j]), [sg] "s"(sg), [sl] "s"((uint32_t)sg), [sh] "s"((uint32_t)(sg >> 32)), [ln] "v"(minus_2lane), [pos] "s"(pos),
                      [ns] "s"(nsig), [sel] "s"(0x05040302u), [dst] "s"(dst)
                    : "memory", "vcc");
            } else {
                const uint32_t off = pos + 2u * __builtin_amdgcn_mbcnt_hi((uint32_t)(plain >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)plain, lane));
                asm volatile(
                    "s_nop 4\n\t"                                                   // (an SGPR a VALU instruction has just written — the compiler reloading a spilled base — needs 5 wait states before a memory instruction reads it: its own code sees to that, an asm statement must)
                    "s_mov_b64 exec, %4\n\t"
                    "global_store_dword %0, %2, %3\n\t"
                    "s_not_b64 exec, exec\n\t"
                    "global_store_short_d16_hi %0, %1, %3\n\t"
                    "s_mov_b64 exec, -1"
                    ::"v"(off), "v"(P), "v"(qq[j]), "s"(dst), "s"(plain) : "memory", "scc");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // undo the exchanges of this wave's round, last block first: the lowest lane of a slot holds the pre-block entry, so the
    // answers go back lane-reversed in ONE ds_write_b16 per block (ascending lane service order: the highest physical lane =
    // the lowest original lane wins)
    // (`skip`: blocks of the round that exchanged nothing — the raw copies an ordered round predicted, below)
    auto rollback_round = [&](uint32_t skip = 0u) {
        if constexpr (SPLIT) { uint32_t t[R]; reload_quads(t, cur_round); park(0, t); } else park(0, q);
        park(1, ra);                                                  // (a rolled loop: this path is rare, its code must not weigh on the common one)
#pragma nounroll
        for (int j = (int)R - 1; j >= 0; --j) {
            if ((skip >> j) & 1u) continue;
            const uint32_t P = parked(0, (uint32_t)j) * kHashMul, srj = parked(1, (uint32_t)j);
            const uint32_t hi = (P >> 16) & 1u;                                    // 1: the slot is the upper half of its dword
            const uint32_t a16 = ((P >> 15) & 0x1fffcu) + 2u * hi;
            const uint32_t prev = hi ? (srj >> 16) : (srj & 0xffffu);
            const uint32_t ar = bperm(63u - lane, a16), pr = bperm(63u - lane, prev);
            dict_store(ar, pr);
        }
    };
    // Abort protocol (all 16 waves; `holding`: this wave has exchanged `hold_round` and not committed it).  After the first
    // barrier nobody is inside a critical section, D says how far the dictionary got (rounds < d exchanged), A which round
    // failed; rounds d-1 .. A are rolled back one per barrier step by their owners, then the chain restarts at A in slow mode.
    uint32_t hold_skip = 0;                                                       // blocks of the round this wave holds that exchanged nothing (a run-ahead round's predicted raw copies)
    auto abort_sync = [&](bool holding, uint32_t hold_round) {
        wg_barrier();
        const u32x2 v = lds_peek2(sy + kSyD);
        const uint32_t d = rfl(v.x) >> 1, a = rfl(v.y);
        for (uint32_t x = d; x-- > a;) {
            if (holding && hold_round == x) rollback_round(hold_skip);
            wg_barrier();
        }
        if (wave == a % W && lane == 0) {
            lds_poke(sy + kSyD + 12, 0u);                                         // (no run-ahead behind an abort: the restarted round waits for its payload)
            lds_poke(sy + kSyD, (a << 1) | 1u);
            lds_poke(sy + kSyD + 4, kNone);
            lds_poke(sy + kSyO + 4, kNone);
        }
        wg_barrier();
    };

    // PAGED: where a round of `need` bytes goes whose turn it is at stream position `pos` — there, if it ends INSIDE the page (strictly: a position on
    // a page boundary is then always a fresh page's start), else at the start of the spare page, which becomes the stream's page (io/write_buffer.rs:
    // 29-31's running total moves on in the directory instead: bytes used per page).  Called by the holder of the commit token only; `refill`: the
    // spare was taken, a new one is fetched once the tokens have been passed on.
    bool refill = false;
    auto page_place = [&](uint32_t pos, uint32_t need, uint32_t first_block) -> uint32_t {
        if (__builtin_expect((pos & (kPageBytes - 1u)) + need < kPageBytes, 1)) return pos;
        const u32x4 st = lds_peek4(sy + kSyPage);
        const uint32_t base = rfl(st.x), before = rfl(st.y), count = rfl(st.w);
        uint32_t spare = rfl(st.z);
        if (spare == kNone) spare = rfl(lane == 0 ? atomicAdd(seg.page_counter, 1u) : 0u);          // (the refill has not come back yet: rare)
        if (spare >= seg.page_limit) { if (err && lane == 0) atomicOr(err, 2u); spare = seg.page_limit - 1u; }   // (cannot happen, see above; never write past the output)
        if (lane == 0) {
            dir[4u * count + 2u] = pos - base;                                    // bytes of stream in the page that is left
            *reinterpret_cast<uint4*>(dir + 4u * (count + 1u)) = make_uint4(spare, first_block, 0u, 0u);
            const u32x4 v = {spare << kPageShift, before + (pos - base), kNone, count + 1u};
            asm volatile("ds_write_b128 %0, %1" ::"v"(sy + kSyPage), "v"(v) : "memory");
        }
        // a new spare only if the stream is LIKELY to outgrow this page: what is left of the chunk at the bytes per block the stream has had so far, and
        // an eighth on top.  The last page of a chunk mostly needs none, and a spare nobody uses is 64 KiB of the container (one per chunk until round 6:
        // 2.4 % of the headline blob).  If the guess is wrong the next change of pages takes its page from the counter itself (above: spare == kNone).
        refill = (uint64_t)(nfull - first_block + 1u) * (before + (pos - base)) * 9u >= (uint64_t)first_block * (8u * kPageBytes);
        return spare << kPageShift;
    };
    auto page_refill = [&]() {
        if (lane == 0) { const uint32_t pg = atomicAdd(seg.page_counter, 1u); lds_poke(sy + kSyPage + 8u, pg); }
        refill = false;
    };
    // (the last whole round also keeps room for what follows it on one wave: the blocks of the partial round and the ragged block — so that no page
    // starts inside the decoder's in-order tail)
    const uint32_t tail_need = PAGED ? (nfull - nrounds * R + 1u) * (kSig + kBlock) : 0u;
    // ---- SPLIT: the ring and the mail boxes (constants above) ----
    // Every spin of either role looks for a raised abort — the work-group barrier of the protocol counts all sixteen waves; an emit wave never holds an
    // uncommitted round, a chain wave none at these places — and for the watchdog's poison.
    auto join_abort = [&]() {
        const u32x2 v = lds_peek2(sy + kSyD);
        const uint32_t A = rfl(v.y);
        if (__builtin_expect(A != kNone, 0)) { if (A == kPoison) wave_exit(); abort_sync(false, 0); }
    };
    // chain wave: round r's quads out of the ring (its LDS reads execute in issue order: whoever sees the `freed` word may overwrite the slot)
    auto ring_take = [&](uint32_t (&d)[R], uint32_t r) {
        const uint32_t slot = r % kRingSlots;
        for (uint32_t spins = 0; !poll_word(kEncRingSync + 4u * slot, r + 1u, 4);) { join_abort(); __builtin_amdgcn_s_sleep(1); watchdog(spins, sy, err, lane); }
        clk.mark(5);
        const uint32_t at = kEncRing + slot * kSlotBytes + 4u * lane;
        // (one statement, issue to wait: an answer of an asynchronous LDS read exists for the compiler only when the statement ends)
        if constexpr (R == 16) asm volatile("ds_read_b32 %0, %16 offset:0\n\t"
                     "ds_read_b32 %1, %16 offset:256\n\t"
                     "ds_read_b32 %2, %16 offset:512\n\t"
                     "ds_read_b32 %3, %16 offset:768\n\t"
                     "ds_read_b32 %4, %16 offset:1024\n\t"
                     "ds_read_b32 %5, %16 offset:1280\n\t"
                     "ds_read_b32 %6, %16 offset:1536\n\t"
                     "ds_read_b32 %7, %16 offset:1792\n\t"
                     "ds_read_b32 %8, %16 offset:2048\n\t"
                     "ds_read_b32 %9, %16 offset:2304\n\t"
                     "ds_read_b32 %10, %16 offset:2560\n\t"
                     "ds_read_b32 %11, %16 offset:2816\n\t"
                     "ds_read_b32 %12, %16 offset:3072\n\t"
                     "ds_read_b32 %13, %16 offset:3328\n\t"
                     "ds_read_b32 %14, %16 offset:3584\n\t"
                     "ds_read_b32 %15, %16 offset:3840\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11]), "=&v"(d[12]), "=&v"(d[13]), "=&v"(d[14]), "=&v"(d[15])
                     : "v"(at) : "memory");
        if constexpr (R == 12) asm volatile("ds_read_b32 %0, %12 offset:0\n\t"
                     "ds_read_b32 %1, %12 offset:256\n\t"
                     "ds_read_b32 %2, %12 offset:512\n\t"
                     "ds_read_b32 %3, %12 offset:768\n\t"
                     "ds_read_b32 %4, %12 offset:1024\n\t"
                     "ds_read_b32 %5, %12 offset:1280\n\t"
                     "ds_read_b32 %6, %12 offset:1536\n\t"
                     "ds_read_b32 %7, %12 offset:1792\n\t"
                     "ds_read_b32 %8, %12 offset:2048\n\t"
                     "ds_read_b32 %9, %12 offset:2304\n\t"
                     "ds_read_b32 %10, %12 offset:2560\n\t"
                     "ds_read_b32 %11, %12 offset:2816\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]), "=&v"(d[8]), "=&v"(d[9]), "=&v"(d[10]), "=&v"(d[11])
                     : "v"(at) : "memory");
        if (lane == 0) lds_poke(kEncRingSync + 16u + 4u * slot, r + 1u);
    };
    // emit wave: round r's quads into the ring, once the chain wave of round r - kRingSlots has drained the slot
    auto ring_put = [&](const uint32_t (&d)[R], uint32_t r) {
        const uint32_t slot = r % kRingSlots;
        // (this wave is idle most of a round — it waits here for about five hand-offs of the chain —: long naps at the lowest priority, so that its
        // polls take neither LDS cycles from the exchanges nor issue slots from the waves it waits for)
        if (r >= kRingSlots) {
            __builtin_amdgcn_s_setprio(0);
            for (uint32_t spins = 0; !poll_word(kEncRingSync + 16u + 4u * slot, r + 1u - kRingSlots, 1);) { join_abort(); __builtin_amdgcn_s_sleep(12); watchdog(spins, sy, err, lane); }
            __builtin_amdgcn_s_setprio(1);
        }
        const uint32_t base = kEncRing + slot * kSlotBytes + 4u * lane;
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) DENSITY_LDS_POKE_AT(base, j * 256u, d[j]);
        if (lane == 0) lds_poke(kEncRingSync + 4u * slot, r + 1u);
    };
    // chain wave `wave`, round r committed: its signatures and stream position to the partner (`skip`: the chain wave wrote the round out itself —
    // rounds with raw-copy blocks); the box is free once the partner has taken the pair's previous round
    auto mbox_post = [&](uint32_t r, uint32_t pos, uint32_t lo, uint32_t hi, uint32_t skip) {
        const uint32_t mb = kEncMbox + kMboxBytes * wave;
        if (r >= (uint32_t)W)
            for (uint32_t spins = 0; !poll_word(mb + 144u, r + 1u - W, 2);) { join_abort(); __builtin_amdgcn_s_sleep(1); watchdog(spins, sy, err, lane); }
        if (lane < R) lds_poke2(mb + 8u * lane, lo, hi);
        if (lane == 0) { lds_poke(mb + 132u, pos); lds_poke(mb + 128u, ((r + 1u) << 1) | skip); }
    };
    auto mbox_wait = [&](uint32_t e, uint32_t r, uint32_t& pos, uint32_t& lo, uint32_t& hi) -> bool {
        const uint32_t mb = kEncMbox + kMboxBytes * e;
        uint32_t seq;
        __builtin_amdgcn_s_setprio(0);
        for (uint32_t spins = 0;;) {
            seq = rfl(lds_peek1(mb + 128u));
            if ((seq >> 1) == r + 1u) break;
            join_abort(); __builtin_amdgcn_s_sleep(6); watchdog(spins, sy, err, lane);
        }
        __builtin_amdgcn_s_setprio(1);
        pos = rfl(lds_peek1(mb + 132u));
        const u32x2 sg = lds_peek2(mb + 8u * (lane < R ? lane : 0u));
        lo = lane < R ? sg.x : 0u; hi = lane < R ? sg.y : 0u;
        if (lane == 0) lds_poke(mb + 144u, r + 1u);
        return (seq & 1u) != 0;
    };
    if (SPLIT && wave >= (uint32_t)W) {
        // ---- emit wave e: loads the rounds e, e + 8, ..., hands their quads to the chain wave e through the ring — a round ahead of the one it is
        // about to write out — and writes a round's records once the chain wave has committed it (mail box).  Three register sets: the round waiting
        // for its commit, the next one (on its way into the ring) and the one after that, whose loads are issued BEFORE the wait for the commit and
        // the emit, so that their latency lies under both.
        // The memory queue and the compiler: its bookkeeping cannot see the record stores (issued inside asm statements) and is conservative across
        // the loop's edge, so a wait it places for a LOAD also waits for stores it does not know of.  Inside the emit that was ruinous (measured:
        // 4600 instead of 2600 cycles per round — from the ninth record on, every record waited for the acknowledgement of an older record's store),
        // so the quads are "laundered" once they have landed: an empty statement that redefines them, after which the compiler attaches no pending
        // load to them and the emit runs without a wait.  What is left is one over-long wait per round, in front of the ring transfer (the loads it
        // waits for were issued before the last emit's stores: it sits that emit's stores out too), where this wave has slack.  (Loads issued by hand,
        // out of the compiler's sight, were tried: it copies the "defined" registers at the loop's edge before they have landed — tools/check_isa.py
        // finds such copies.)
        // Priority 1 like a chain wave that hashes: below the chain's critical steps (2, 3), above the waves that only poll for this one's work (0).
        __builtin_amdgcn_s_setprio(1);
        const uint32_t e = wave - W;
        auto launder = [&](uint32_t (&d)[R]) {
#pragma unroll
            for (uint32_t j = 0; j < R; ++j) asm volatile("" : "+v"(d[j]));
        };
        uint32_t qa[R], qb[R], qc[R];
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) { qa[j] = 0; qb[j] = 0; qc[j] = 0; }
        load_round(qa, e);
        if (e < nrounds) ring_put(qa, e);
        launder(qa);
        load_round(qb, e + W);
        auto step = [&](uint32_t (&cur)[R], uint32_t (&nxt)[R], uint32_t (&fut)[R], uint32_t r) {
            clk.start();
            if (r + W < nrounds) ring_put(nxt, r + W);
            launder(nxt);
            clk.mark(1);
            load_round(fut, r + 2u * W);
            uint32_t pos, lo, hi;
            const bool skip = mbox_wait(e, r, pos, lo, hi);
            clk.mark(4);
            if (!skip) emit_round_coded(cur, pos, idx ? idx + (uint64_t)r * R : nullptr, lo, hi);
            clk.mark(6);
        };
        for (uint32_t r = e; r < nrounds; r += 3u * W) {
            step(qa, qb, qc, r);
            if (r + W < nrounds) step(qb, qc, qa, r + W);
            if (r + 2u * W < nrounds) step(qc, qa, qb, r + 2u * W);
        }
    } else {
    if constexpr (KEEP) {
        // (by hand like every later fetch: a load the compiler can see ahead of the loop would make it wait, at the top of every
        // iteration, until all but a few of the previous round's record stores have been acknowledged)
        if (wave < nrounds) { prefetch_quads<R, W>(src + (uint64_t)wave * (R * kBlock) + 4u * lane); quads_landed<R, W, true>(q); }
    } else if (!SPLIT) {
        load_round(q, wave);
    }
    uint32_t poll_tries = 16;                                                     // polls for the FAST token before a look at the whole D line: few while this wave's rounds are ordered ones
    for (uint32_t r = wave; r < nrounds; r += W) {
        clk.start();
        __builtin_amdgcn_s_setprio(1);                                   // (see the priorities note at the exchange)
        if (SPLIT) { cur_round = r; ring_take(q, r); clk.mark(7); }
        uint32_t slo = 0, shi = 0;                                                // lane j: the signature of block j (codec.rs:24-26)
        uint32_t copy_mask = 0, opos = 0;
        bool fast_commit = false, prefetched = false;
        // an ORDERED round (below): its commit payload, and how far it is final — blocks below it_j0, the FSM state in front of it_j0, the raw-copy
        // blocks (final below it_j0, predicted from there on), the prediction and the state behind the round if it holds
        uint32_t P0 = 0, P1 = 0, it_j0 = kNone, it_state = 0, it_raw = 0, it_mode = 0, it_end = 0;
        bool have_turn = false, ahead = false;
        hold_skip = 0;
      for (bool reentered = false;; reentered = true) {   // (re-entered after an abort, and by an ordered round whose prediction failed: the answers have replaced the addresses, so the operands are made again)
        uint32_t zblocks = 0, zq = 0;
        bool zsusp = false;
        auto prepare = [&](const uint32_t (&qq)[R]) {
        uint32_t zmin = 0xffffffffu;
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            operands(qq[j], ra[j], mask[j], val[j], kKeepHash ? &hp[j] : nullptr);
            zmin = val[j] < zmin ? val[j] : zmin;
            __builtin_amdgcn_sched_barrier(0);                                    // block by block: short live ranges, not maximal overlap
        }
        // Blocks with a quad that needs the zero-entry map (about one quad in 64 Ki): found here, ahead of the waits, together with the
        // first such block's quads, so that the commit — which holds up every later round — has next to nothing left to look up.
        if (__builtin_expect(ballot64(zmin == 0) != 0, 0)) {                      // a stored entry 0: the zero quad (harmless) or one outside slot 0
            asm volatile("");                                                     // (nothing of this block is worth computing ahead of the test: sixteen compares of the common path otherwise)
#pragma unroll
            for (uint32_t j = 0; j < R; ++j) zblocks |= (ballot64(val[j] == 0 && qq[j] != 0) != 0 ? 1u : 0u) << j;
            if (zblocks) {
                const uint32_t j0 = (uint32_t)__builtin_ctz(zblocks);
                zq = pick<R>(qq, j0);
                zsusp = pick<R>(val, j0) == 0 && zq != 0;
            }
        }
        };
        if (SPLIT && reentered) reload_quads(q, r);
        prepare(q);
        const bool zero_round = zblocks != 0;
        uint32_t tokaddr = lane == 0 ? sy + kSyD : sy + kSySink + 4u * lane;
        uint32_t tokval = (r + 1u) << 1;                                          // (in its register before the wait, like the operands)
        asm volatile("" : "+v"(tokval));
        pin_operands<R>(ra, mask, val);                                           // complete before the wait for the token

        clk.mark(0);
        clk.stamp(r, 0, lane);
        __builtin_amdgcn_s_setprio(2);
        {
            // ---- D chain: wait for this round's turn ----
            uint32_t slow = 1, dS = 0, dF = 0;                                    // dS, dF: the D line's run-ahead words (below: ordered rounds that run ahead)
            bool got_payload = false;
            // (the memo of FSM predictions as it stands now, read AHEAD of the wait: inside a stretch its entries are stable, and the look-up behind the
            // token is then a compare instead of an LDS round trip — 200 cycles of every run-ahead hop; a miss reads it again)
            u32x4 memo_early = {kNone, 0u, 0u, 0u};
            if (poll_tries != 16 && !have_turn) memo_early = lds_peek4(sy + kSyMemo + 16u * (lane & (kMemoEntries - 1u)));
            if (!have_turn)
            for (uint32_t spins = 0;;) {
                if (poll_tries != 16) {
                    // this wave's last round was an ordered one: most likely this one is too, and then it needs the commit payload as well —
                    // the D line and the O line in one look instead of one after the other (behind a few tight polls for this round's slow token:
                    // a round that runs ahead is handed over like a fast one, and the two-line look alone found it 775 cycles late)
                    // (first a few tight polls of the D line for this round's slow token — a round that runs ahead is handed over like a fast one, every
                    // LDS round trip on the way is 200 cycles of the hop: the line's run-ahead words come with the token)
                    {
                        u32x4 dl1;
                        if (poll_line(sy + kSyD, (r << 1) | 1u, 8, dl1) && rfl(dl1.y) == kNone) { slow = 1; dS = rfl(dl1.z); dF = rfl(dl1.w); break; }
                    }
                    u32x4 dl, ol;
                    lds_peek4x2(sy + kSyD, dl, ol);
                    const uint32_t D = rfl(dl.x), A = rfl(dl.y);
                    if (__builtin_expect(A != kNone, 0)) { if (A == kPoison) wave_exit(); abort_sync(false, 0); continue; }
                    if ((D >> 1) == r) {
                        slow = D & 1u; dS = rfl(dl.z); dF = rfl(dl.w);
                        if (slow && rfl(ol.x) == r) { P0 = rfl(ol.z); P1 = rfl(ol.w); got_payload = true; }
                        break;
                    }
                    if (D & 1u) backoff_ordered(r - (D >> 1)); else backoff(r - (D >> 1));
                    watchdog(spins, sy, err, lane);
                    continue;
                }
                if (poll_word(sy + kSyD, r << 1, 16)) { slow = 0; break; }         // the common hand-off: fast token for this round
                const u32x4 v = lds_peek4(sy + kSyD);
                const uint32_t D = rfl(v.x), A = rfl(v.y);
                if (__builtin_expect(A != kNone, 0)) { if (A == kPoison) wave_exit(); abort_sync(false, 0); continue; }
                if ((D >> 1) == r) { slow = D & 1u; dS = rfl(v.z); dF = rfl(v.w); break; }
                backoff(r - (D >> 1));
                watchdog(spins, sy, err, lane);
            }
            clk.mark(1);
            clk.stamp(r, 1, lane);
            if (__builtin_expect(!slow, 1)) {
                // ---- fast round: R speculative exchanges, token passed behind them ----
                // Priorities: the SIMD's arbiter prefers, at equal priority, the wave that was launched first, which leaves the last-launched
                // wave of each SIMD short of issue slots and late for its turns.  So a wave's priority follows its deadline instead: 3
                // inside the exchanges, 2 on the way to the commit and while it waits for a token, 1 while it prepares its next round, 0
                // while it writes records out (nobody waits for those).
                __builtin_amdgcn_s_setprio(3);
                exchange_tied<R>(ra, mask, val, tokaddr, tokval, false);
                __builtin_amdgcn_s_setprio(2);
                clk.mark(2);
                clk.stamp(r, 2, lane);
                // EARLY: the next round's quads are asked for HERE, a signature pass and a commit wait earlier than behind the commit (their
                // latency under load is of the order of a whole emit); once per round, whatever becomes of it (an abort re-enters the loop)
                if constexpr (EARLY && KEEP) if (!prefetched && r + W < nrounds) { prefetch_quads<R, W>(src + (uint64_t)(r + W) * (R * kBlock) + 4u * lane); prefetched = true; }
                // The signatures (chameleon.rs:90-99: MAP flag = 1 iff the slot held this quad), block j's into lane j of slo / shi.  gfx950: an SGPR
                // written by a VALU instruction — the compare — needs 2 wait states before a VALU instruction — the lane write — reads it, which the
                // compiler sees to in its own code but not inside an asm statement: so block j's two lane writes go out behind block j + 1's compare
                // (and block j's own: three instructions in between), four instructions per block with no idle one.
                uint64_t sgp;
                {
                    const uint32_t x0 = (ra[0] ^ val[0]) & mask[0];
                    asm volatile("v_cmp_eq_u32_e64 %0, 0, %1" : "=s"(sgp) : "v"(x0));
                }
#pragma unroll
                for (uint32_t j = 1; j < R; ++j) {
                    const uint32_t xj = (ra[j] ^ val[j]) & mask[j];
                    uint64_t sgn;
                    if (j == 1) {                                                 // (block 0's compare has no lane writes behind it: one idle state)
                        asm volatile("v_cmp_eq_u32_e64 %2, 0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %4, %6\n\tv_writelane_b32 %1, %5, %6"
                                     : "+v"(slo), "+v"(shi), "=&s"(sgn) : "v"(xj), "s"((uint32_t)sgp), "s"((uint32_t)(sgp >> 32)), "n"(0));
                    } else {
                        asm volatile("v_cmp_eq_u32_e64 %2, 0, %3\n\tv_writelane_b32 %0, %4, %6\n\tv_writelane_b32 %1, %5, %6"
                                     : "+v"(slo), "+v"(shi), "=&s"(sgn) : "v"(xj), "s"((uint32_t)sgp), "s"((uint32_t)(sgp >> 32)), "n"(j - 1));
                    }
                    sgp = sgn;
                }
                asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, %4\n\tv_writelane_b32 %1, %3, %4" : "+v"(slo), "+v"(shi) : "s"((uint32_t)sgp), "s"((uint32_t)(sgp >> 32)), "n"(R - 1));
                // everything the commit needs that does not depend on the token: incompressible records (codec.rs:68: 8 + 256 - 2*hits >= 256) and the
                // bytes of the round — a sum over the lanes' record lengths instead of a scalar count and add per block
                const uint32_t nhv = (uint32_t)(__builtin_popcount(slo) + __builtin_popcount(shi));
                uint32_t inc = (uint32_t)ballot64(lane < R && nhv <= 4u);
                uint32_t sum;
                {
                    uint32_t acc = kSig + kBlock - 2u * nhv;                                              // (lanes >= R hold no signature: their slo / shi are 0, and they are not summed)
                    acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x111, 0xf, 0xf, true);     // row_shr:1
                    acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x112, 0xf, 0xf, true);     // row_shr:2
                    acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x114, 0xf, 0xf, true);     // row_shr:4
                    if (R > 8) acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x118, 0xf, 0xf, true);   // row_shr:8
                    sum = rlane_u(acc, (int)R - 1);
                }
                uint32_t hits = 0;                                                                        // (only the rare zero-entry path below wants the count itself)
                asm volatile("" : "+s"(inc), "+s"(sum));                        // computed HERE: left to itself the compiler sinks both — and the 16 signatures they need — below the token wait, into the commit
                if (!kKeepQuads && !SPLIT) load_round(q, r);              // the quads again (from L2): not kept across the wait for the token (split: the emit wave has them)
                clk.mark(3);
                // ---- O chain: commit ----
                bool aborted = false;
                for (uint32_t spins = 0;;) {
                    u32x4 v = lds_peek4(sy + kSyO);
                    for (uint32_t i = 0; i < 16 && rfl(v.x) != r; ++i) v = lds_peek4(sy + kSyO);   // token and payload in one read
                    const uint32_t O = rfl(v.x), A = rfl(v.y);
                    if (O == r) { P0 = rfl(v.z); P1 = rfl(v.w); break; }
                    if (__builtin_expect(A != kNone, 0)) { if (A == kPoison) wave_exit(); abort_sync(true, r); aborted = true; break; }
                    backoff(r - O);
                    watchdog(spins, sy, err, lane);
                }
                if (aborted) continue;
                // Zero-entry map, in stream order (this wave holds the commit token): every quad whose stored entry is 0 outside slot 0 marks
                // its slot; its MAP flag — the slot read 0 — stands only if the slot had been marked before, i.e. really held this entry
                // and not just never anything.  `flipped`: the marks this round set itself (taken back if the round is rolled back).
                uint32_t flipped = 0;
                if (__builtin_expect(zero_round, 0)) {
                    uint32_t qz_[R];
                    if constexpr (SPLIT) reload_quads(qz_, r);
                    const uint32_t (&qz)[R] = *(SPLIT ? &qz_ : &q);
                    clk.note(r, 1, lane);
                    hits = (R * (kSig + kBlock) - sum) >> 1;
                    bool first = true;
                    for (uint32_t zb = zblocks; zb; zb &= zb - 1u, first = false) {
                        const uint32_t j = (uint32_t)__builtin_ctz(zb);
                        const uint32_t qv = first ? zq : pick<R>(qz, j);
                        const bool susp = first ? zsusp : (pick<R>(val, j) == 0 && qv != 0);
                        const uint32_t zbit = zmap_claim_in_order(zmap, susp, (qv * kHashMul) >> 16, lane);
                        flipped |= (susp && !zbit ? 1u : 0u) << j;
                        const uint64_t sg = ((uint64_t)rlane(shi, j) << 32) | rlane(slo, j);
                        const uint64_t lost = ballot64(susp && !zbit) & sg;
                        slo = lane == j ? (uint32_t)(sg & ~lost) : slo;
                        shi = lane == j ? (uint32_t)((sg & ~lost) >> 32) : shi;
                        hits -= (uint32_t)__builtin_popcountll(lost);
                    }
                    inc = (uint32_t)ballot64(lane < R && (uint32_t)(__builtin_popcount(slo) + __builtin_popcount(shi)) <= 4u);
                    sum = R * (kSig + kBlock) - 2u * hits;
                }
                // which blocks the FSM would have turned into raw copies: block j+1 iff inc[j] && prev[j] (protection_state.rs:38-47)
                const uint32_t t = inc & ((inc << 1) | ((P1 >> 16) & 1u));
                if (__builtin_expect((P1 & 0xffu) != 0 || (t & ((1u << (R - 1)) - 1u)) != 0, 0)) {
                    if (ballot64(flipped != 0)) {
                        uint32_t qz_[R];
                        if constexpr (SPLIT) reload_quads(qz_, r);
                        const uint32_t (&qz)[R] = *(SPLIT ? &qz_ : &q);
                        for (uint32_t zb = zblocks; zb; zb &= zb - 1u) {
                            const uint32_t j = (uint32_t)__builtin_ctz(zb);
                            if ((flipped >> j) & 1u) zmap.clear((pick<R>(qz, j) * kHashMul) >> 16);
                        }
                    }
                    // (the chunk's abort count, for the ordered rounds' patience: this wave holds the commit token, the payload is its to amend)
                    if (lane == 0) { lds_poke(sy + kSyO + 12, ((P1 >> 24) & 3u) < 3u ? P1 + 0x01000000u : P1); lds_poke(sy + kSyD + 4, r); lds_poke(sy + kSyO + 4, r); }
                    clk.count(4, lane);
                    abort_sync(true, r);
                    continue;
                }
                uint32_t g_out;
                if (__builtin_expect((P1 & 0x1ffffu) == 0 && inc == 0, 1)) {
                    g_out = (P1 & 0xe3e1ffffu) | ((((P1 >> 17) + R) & 15u) << 17);    // calm, start == 1: only the block counter moves (and no incompressible stretch is running: its count, bits 26..28, goes)
                } else {
                    Guard g = unpack_guard(P1);
#pragma unroll
                    for (uint32_t j = 0; j < R; ++j) (void)g.block_is_copy();        // no block was a copy: bookkeeping only (:19-27)
                    g.penalty = ((t >> (R - 1)) & 1u) ? g.start : 0u;
                    g.prev = (inc >> (R - 1)) & 1u;
                    g_out = pack_guard(g) | (P1 & 0x03000000u);
                }
                opos = PAGED ? page_place(P0, sum + (r + 1u == nrounds ? tail_need : 0u), r * R) : P0;
                if (lane == 0) {
                    lds_poke2(sy + kSyO + 8, opos + sum, g_out);
                    lds_poke(sy + kSyO, r + 1u);
                }
                copy_mask = 0;
                __builtin_amdgcn_s_setprio(0);
                fast_commit = true;
                if (PAGED && refill) page_refill();
                if constexpr (KEEP) if (!prefetched && r + W < nrounds) prefetch_quads<R, W>(src + (uint64_t)(r + W) * (R * kBlock) + 4u * lane);   // next round's quads: in flight behind the commit, landed by the end of the emit
                clk.mark(4);
                clk.count(0, lane);
                poll_tries = 16;
                break;
            }
            // ---- ordered round (round 5): everything before it is final first, then the round in batches ----
            // A round behind an abort or behind unrest does not speculate ACROSS rounds: it waits for its commit payload, so the FSM state at its
            // first block is known.  INSIDE the round the raw-copy blocks are predicted — calm state: none; inside an incompressible stretch
            // (penalty running, or the last coded block incompressible): every coded block incompressible, which makes the FSM a function of its
            // state alone (protection_state.rs:19-47) —, the blocks predicted coded exchange in one go like a fast round's, and the FSM walked over
            // the signatures they produce must arrive at the predicted raw blocks: by induction, block by block, the round is then exactly the
            // sequential one.  Where it does not — block jm — everything below jm IS final; this wave takes back its exchanges from jm on (nobody
            // has seen them: the dictionary token leaves only with the commit), predicts again from the exact state at jm — the other way round:
            // a raw copy where none was expected starts an incompressible stretch, a coded block where a copy was expected ends one — and
            // exchanges the rest of the round again; jm only grows.  No barrier, no other wave involved: data that flips between compressible and
            // incompressible every few KiB costs a round a second batch, not an abort of the work-group per flip; incompressible data runs in
            // batches too.
            bool ordered = false;
            {
                if (!have_turn) {
                    bool aborted = false;
                    // RUN-AHEAD (round 5): inside a long incompressible stretch — random input, data that is compressed already — the state behind a
                    // round is its prediction round after round (every coded block incompressible: the FSM is a function of its state alone), so the
                    // dictionary token need not wait for the commit: the predecessor passed it on right behind its exchanges, with the state it
                    // PREDICTS for this round (D line, words 2 and 3).  This round predicts from that, exchanges, passes the token on the same way,
                    // and only then waits for its commit payload — which must show the state it assumed, and its own signatures the stretch going on;
                    // if not, the abort protocol takes back what ran ahead, as for a fast round, and the chain restarts here without run-ahead.
                    // An ordinary ordered round starts it after kStormRounds rounds of an unbroken stretch (payload bits 26..28).
                    ahead = dF == 1u && !zero_round && (dS & 0x100ffu) != 0;
                    if (!got_payload && !ahead)
                    for (uint32_t spins = 0;;) {
                        const u32x4 v = lds_peek4(sy + kSyO);
                        const uint32_t O = rfl(v.x), A = rfl(v.y);
                        if (O == r) { P0 = rfl(v.z); P1 = rfl(v.w); break; }
                        if (A != kNone) { if (A == kPoison) wave_exit(); abort_sync(false, 0); aborted = true; break; }
                        backoff(r - O);
                        watchdog(spins, sy, err, lane);
                    }
                    if (aborted) continue;
                    have_turn = true;
                    if (!kKeepQuads && !SPLIT) load_round(q, r);                  // (as in the fast path: the quads are not kept across the waits; split: every use below loads its own)
                    // (a fresh chunk's first round is the cold start — raw copies for certain, nothing to predict —, and the rare zero-entry quads
                    // are settled block by block: those rounds are walked in order, below)
                    if (!zero_round && !(r == 0 && !seg.init_images)) {
                        it_j0 = 0; it_state = ahead ? dS & 0x1fffffu : P1 & 0x1fffffu; it_raw = 0;
                        it_end = (it_state & ~0x1e0000u) | ((((it_state >> 17) + R) & 15u) << 17);   // (calm, start == 1, no incompressible block: only the counter moves)
                        it_mode = (it_state & 0x100ffu) != 0 ? 1u : 0u;              // penalty running or the last coded block incompressible
                        if ((it_state & 0x1ffffu) != 0) {                                // (calm, start == 1: no raw copy while no block is incompressible, only the counter moves: the check below)
                            // Inside an incompressible stretch the state in front of a round repeats with a period of a few rounds (the counter moves
                            // by R = 16 a round, penalty and start go round a short cycle), and the prediction is a function of that state alone: a
                            // memo of eight in the sync block, touched only by the holder of the commit token, saves the walk — a few hundred scalar
                            // instructions in the one place where every later round waits.
                            u32x4 e = memo_early;                                    // lane l: entry l mod 8; the round's number picks the one to replace
                            uint64_t found = ballot64(e.x == it_state);
                            if (it_mode && found) clk.count(6, lane);
                            if (!(it_mode && found)) { e = lds_peek4(sy + kSyMemo + 16u * (lane & (kMemoEntries - 1u))); found = ballot64(e.x == it_state); if (it_mode && found) clk.count(7, lane); }
                            if (it_mode && found) {
                                const uint32_t l0 = (uint32_t)__builtin_ctzll(found);
                                it_raw = rlane(e.y, l0); it_end = rlane(e.z, l0);
                            } else {
                                fsm_predict<R>(it_state, 0u, it_mode, 0u, it_raw, it_end);
                                if (it_mode && lane == 0) { const u32x4 v = {it_state, it_raw, it_end, 0u}; asm volatile("ds_write_b128 %0, %1" ::"v"(sy + kSyMemo + 16u * (r & (kMemoEntries - 1u))), "v"(v) : "memory"); }
                            }
                        }
                    }
                }
                ordered = it_j0 != kNone;
                if (ordered) {
                }
            }
            bool batched = false;
            uint32_t osum = 0, ounrest = 0;
            Guard og;
            if (ordered) {
                // only the blocks from it_j0 on that are predicted coded exchange (final blocks and raw copies — codec.rs:35-37 — touch no state); no
                // token behind them: it leaves with the commit
                const uint32_t keep_lo = slo, keep_hi = shi;                       // (final blocks keep their signatures)
                if (ahead) {                                                      // the token behind the exchanges (LDS order), with the state predicted for the next round
                    exchange_some_ahead<R>(ra, mask, val, rfl(it_raw), sy + kSyD, it_end, ((r + 1u) << 1) | 1u);
                    hold_skip = it_raw;
                    clk.stamp(r, 2, lane);
                } else
                exchange_some<R>(ra, mask, val, rfl(it_raw | ((1u << it_j0) - 1u)));
#pragma unroll
                for (uint32_t j = 0; j < R; ++j) {                                // chameleon.rs:90-99 (an idle block's "signature" is never looked at)
                    const uint64_t sg = ballot64(((ra[j] ^ val[j]) & mask[j]) == 0);
                    slo = lane == j ? (uint32_t)sg : slo;
                    shi = lane == j ? (uint32_t)(sg >> 32) : shi;
                }
                // ---- do the signatures lead the FSM to the predicted raw copies? ----
                {
                    const uint32_t below = (1u << it_j0) - 1u, all = (1u << R) - 1u;
                    slo = lane < it_j0 ? keep_lo : slo;                            // (final blocks keep their signatures; theirs of this pass are of idle lanes)
                    shi = lane < it_j0 ? keep_hi : shi;
                    const uint32_t nh2 = (uint32_t)(__builtin_popcount(slo) + __builtin_popcount(shi));
                    const uint32_t inc_all = (uint32_t)ballot64(lane < R && nh2 <= 4u) & ~it_raw;   // codec.rs:68, coded blocks
                    const uint32_t coded_new = all & ~it_raw & ~below;
                    bool done = ((inc_all ^ (it_mode ? all : 0u)) & coded_new) == 0;   // every block behaved as predicted: the prediction's end state stands
                    if (ahead) {
                        // the commit turn: only now is the state in front of this round known — it must be the one that was assumed
                        bool aborted = false;
                        for (uint32_t spins = 0;;) {
                            const u32x4 v = lds_peek4(sy + kSyO);
                            const uint32_t O = rfl(v.x), A = rfl(v.y);
                            if (O == r) { P0 = rfl(v.z); P1 = rfl(v.w); break; }
                            if (A != kNone) { if (A == kPoison) wave_exit(); abort_sync(true, r); aborted = true; break; }
                            backoff_ordered(r - O);
                            watchdog(spins, sy, err, lane);
                        }
                        if (!aborted && (!done || (P1 & 0x1fffffu) != it_state)) {
                            if (lane == 0) { lds_poke(sy + kSyO + 12, ((P1 >> 24) & 3u) < 3u ? P1 + 0x01000000u : P1); lds_poke(sy + kSyD + 4, r); lds_poke(sy + kSyO + 4, r); }
                            clk.count(4, lane);
                            abort_sync(true, r);
                            aborted = true;
                        }
                        if (aborted) { ahead = false; have_turn = false; it_j0 = kNone; hold_skip = 0; continue; }
                    } else
                    if (!done) {
                        uint32_t sm, mm;
                        const uint32_t jm = fsm_verify<R>(it_state, it_j0, inc_all, it_raw, sm, mm);
                        if (jm == (uint32_t)R) { done = true; it_end = sm; }          // (single incompressible blocks in a calm round: no raw copy came of them)
                        else {
                            clk.count(2, lane);
                            rollback_round(it_raw | ((1u << jm) - 1u));             // the exchanges from jm on, last block first
                            uint32_t raw_new;
                            fsm_predict<R>(sm, jm, mm, it_raw, raw_new, it_end);
                            it_raw = raw_new; it_j0 = jm; it_state = sm; it_mode = mm;
                            continue;
                        }
                    }
                    batched = true;
                    clk.count(1, lane);
                    og = unpack_guard(it_end);
                    uint32_t acc = ((it_raw >> lane) & 1u) ? kBlock : kSig + kBlock - 2u * nh2;              // lane j < R: bytes of block j
                    acc = lane < R ? acc : 0u;
                    acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x111, 0xf, 0xf, true);     // row_shr:1
                    acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x112, 0xf, 0xf, true);     // row_shr:2
                    acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x114, 0xf, 0xf, true);     // row_shr:4
                    acc += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)acc, 0x118, 0xf, 0xf, true);     // row_shr:8
                    osum = rlane_u(acc, 15);
                    ounrest = inc_all != 0 ? 1u : 0u;
                }
            }
            // ---- in-order round: everything before it is final (the wait above); walk the blocks with the FSM ----
            Guard g = batched ? og : unpack_guard(P1);
            uint32_t sum = osum, unrest = ounrest;
            copy_mask = batched ? it_raw : 0u;
            if (!batched) {
                clk.count(3, lane);
                slo = 0; shi = 0;
                if constexpr (SPLIT) { uint32_t t[R]; reload_quads(t, r); park(0, t); } else park(0, q);   // (a rolled loop, as in rollback_round)
#pragma nounroll
                for (uint32_t j = 0; j < R; ++j) {
                    const uint32_t qv = parked(0, j);
                    uint32_t a, m, v;
                    operands(qv, a, m, v);
                    bool raw;
                    uint64_t sg;
                    block_in_order(g, qv, a, m, v, sg, raw);
                    slo = lane == j ? (uint32_t)sg : slo;
                    shi = lane == j ? (uint32_t)(sg >> 32) : shi;
                    copy_mask |= (raw ? 1u : 0u) << j;
                    unrest |= g.prev;
                    sum += raw ? kBlock : kSig + kBlock - 2u * (uint32_t)__builtin_popcountll(sg);
                }
            }
            opos = PAGED ? page_place(P0, sum + (r + 1u == nrounds ? tail_need : 0u), r * R) : P0;
            if (seg.raw_blocks && copy_mask && lane == 0) atomicAdd(seg.raw_blocks + chunk, (uint32_t)__builtin_popcount(copy_mask));
            // back to speculation ACROSS rounds only after quiet_rounds() rounds in a row without an incompressible or raw block (the count rides in
            // bits 21..23 of the commit payload): a mis-speculated fast round costs a work-group barrier and the roll-back of every round that ran
            // ahead — dozens of ordered rounds' worth
            const uint32_t streak = (g.penalty | copy_mask | unrest) != 0 ? 0u : (((P1 >> 21) & 7u) < 7u ? ((P1 >> 21) & 7u) + 1u : 7u);
            const uint32_t stay_slow = streak < quiet_rounds(P1) ? 1u : 0u;
            // (rounds in a row that were one incompressible stretch, as predicted from their first block on: run-ahead starts behind kStormRounds of them)
            const uint32_t storm = batched && it_mode && it_j0 == 0 ? (((P1 >> 26) & 7u) < 7u ? ((P1 >> 26) & 7u) + 1u : 7u) : 0u;
            if (lane == 0) {
                lds_poke2(sy + kSyO + 8, opos + sum, pack_guard(g) | (streak << 21) | (P1 & 0x03000000u) | (storm << 26));
                lds_poke(sy + kSyO, r + 1u);
                if (!ahead) {                                                     // (a round that ran ahead passed the dictionary token on behind its exchanges)
                    lds_poke2(sy + kSyD + 8, pack_guard(g), storm >= kStormRounds && stay_slow ? 1u : 0u);
                    lds_poke(sy + kSyD, ((r + 1u) << 1) | stay_slow);
                }
            }
            hold_skip = 0;
            if (ahead) clk.count(5, lane);
            if constexpr (KEEP) if (!prefetched && r + W < nrounds) prefetch_quads<R, W>(src + (uint64_t)(r + W) * (R * kBlock) + 4u * lane);   // (as behind a fast commit)
            poll_tries = 2;
            if (PAGED && refill) page_refill();
            clk.mark(7);
            break;
        }
      }

        clk.mark(5);

        // ---- emit: records of this round and their block-index bytes ----
        if (SPLIT && copy_mask == 0) {
            __builtin_amdgcn_s_setprio(1);
            mbox_post(r, opos, slo, shi, 0u);                                     // the partner writes the records (it has the quads)
        } else if (__builtin_expect(copy_mask == 0, 1)) {
            emit_round_coded(q, opos, idx ? idx + (uint64_t)r * R : nullptr, slo, shi);
        } else {
            // (unrolled since round 5 — ordered rounds made incompressible data a common case: the rolled loop picked every block's quads out of
            // the registers by a chain of selects, ten thousand cycles a round)
            uint8_t* rec = dst + opos;
            uint32_t qe_[R];
            if constexpr (SPLIT) reload_quads(qe_, r);
            const uint32_t (&qe)[R] = *(SPLIT ? &qe_ : &q);
            if (idx && lane < R) idx[(uint64_t)r * R + lane] = (uint8_t)(((copy_mask >> lane) & 1u) ? kIdxCopy : (uint32_t)(__builtin_popcount(slo) + __builtin_popcount(shi)));
#pragma unroll
            for (uint32_t j = 0; j < R; ++j) {
                const bool raw = (copy_mask >> j) & 1u;
                const uint64_t sg = ((uint64_t)rlane_u(shi, (int)j) << 32) | rlane_u(slo, (int)j);
                emit_block(rec, qe[j], sg, raw);
                rec += raw ? kBlock : kSig + kBlock - 2u * (uint32_t)__builtin_popcountll(sg);
            }
            if (SPLIT) mbox_post(r, opos, slo, shi, 1u);                          // (the partner drops its copy of the round)
        }
        if constexpr (SPLIT) {
            // (the next round's quads come out of the ring at the top of the loop)
        } else if constexpr (KEEP) {
            // (both ways out of the round have asked for the next one's quads)
            if (r + W < nrounds) {
                // behind a fast commit at least R stores are younger than the R loads (emit_round_coded: one store per record, one or two for the
                // last — a store none of whose lanes is active is not counted — and the signatures go out in one more)
                if (fast_commit) quads_landed<R, W, false>(q); else quads_landed<R, W, true>(q);
            }
        } else {
            load_round(q, r + W);                                                 // next round's quads (their latency is this wave's slack, not the chain's)
        }
        clk.mark(6);
        clk.stamp(r, 3, lane);
    }
    }
    clk.flush(wave, lane);

    // ---- end of the chunk: every round committed (no abort can follow) ----
    for (uint32_t spins = 0;;) {
        const u32x4 v = lds_peek4(sy + kSyO);
        if (rfl(v.x) == nrounds) break;
        if (rfl(v.y) == kPoison) wave_exit();
        if (rfl(v.y) != kNone) abort_sync(false, 0); else __builtin_amdgcn_s_sleep(4);
        watchdog(spins, sy, err, lane);
    }
    wg_barrier();
    // ---- epilogue on one wave: the blocks of the last, partial round in order, then the ragged block (codec.rs:51-63) ----
    if (wave == 0) {
        const u32x4 v = lds_peek4(sy + kSyO);
        Guard g = unpack_guard(rfl(v.w));
        uint64_t opos = rfl(v.z);
        for (uint32_t b = nrounds * R; b < nfull; ++b) {
            const uint32_t qv = *reinterpret_cast<const uint32_t*>(src + (uint64_t)b * kBlock + 4u * lane);
            uint32_t a, m, vv;
            operands(qv, a, m, vv);
            uint64_t sg;
            bool raw;
            block_in_order(g, qv, a, m, vv, sg, raw);
            emit_block(dst + opos, qv, sg, raw);
            const uint32_t nh = (uint32_t)__builtin_popcountll(sg);
            if (idx && lane == 0) idx[b] = (uint8_t)(raw ? kIdxCopy : nh);
            if (seg.raw_blocks && raw && lane == 0) atomicAdd(seg.raw_blocks + chunk, 1u);
            opos += raw ? kBlock : kSig + kBlock - 2u * nh;
        }
        if (seg.final_guard && lane == 0) seg.final_guard[chunk] = pack_guard(g) | (g.penalty == 0 ? 0x80000000u : 0u);   // (a segment that is not the stream's last ends on a whole block; bit 31: the next one may start speculating)
        const uint64_t end = encode_ragged_block(src, len, nfull, dst, opos, g, idx, 0u, zmap, lane);
        if (PAGED) {                                                              // the stream's length is the bytes used over its pages; the last page's share and the count go into the directory
            const u32x4 st = lds_peek4(sy + kSyPage);
            const uint32_t base = rfl(st.x), before = rfl(st.y), count = rfl(st.w);
            if (lane == 0) {
                dir[4u * count + 2u] = (uint32_t)end - base;
                *reinterpret_cast<uint4*>(dir) = make_uint4(count, 0u, 0u, 0u);
                sizes[chunk] = (uint64_t)before + ((uint32_t)end - base);
            }
        } else
        if (lane == 0) sizes[chunk] = end;
    }
    if (seg.final_images) {                                                        // the dictionary as this chunk leaves it
        wg_barrier();
        uint4* image = reinterpret_cast<uint4*>(seg.final_images + chunk * kSegImageBytes);
        const uint4* p = reinterpret_cast<const uint4*>(smem);
        for (uint32_t i = threadIdx.x; i < (kTableBytes + kZmapBytes) / 16; i += kThreads) image[i] = p[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Segments of one long stream (whole-stream-exact parallel encode, api.hip::run_stream_encode_segmented)
// ---------------------------------------------------------------------------------------------------------------
// "Last writers": the dictionary image a FRESH table has after every block of a chunk went through it (no raw-copy blocks: what
// the segmented encode speculates for every segment but the first).  The D chain of the encoder and nothing else: rounds of 16
// blocks rotate over 16 waves, each wave issues its round's ordered exchanges behind the token and drops the answers; zero-entry
// quads mark their slot (the marks need no order: a stale mark under a non-zero entry is never consulted).  Whole rounds only.
__global__ __launch_bounds__(1024) void chameleon_lastwriters_rot(const uint8_t* __restrict__ in, uint64_t chunk_bytes, uint8_t* __restrict__ images,
                                                                   uint32_t* __restrict__ err) {
    constexpr int R = 16, W = 16;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint64_t chunk = blockIdx.x;
    const uint8_t* src = in + chunk * chunk_bytes;
    const uint32_t nrounds = (uint32_t)(chunk_bytes / (R * kBlock));
    const uint32_t sy = kEncSync;
    const ZmapLds zmap{kEncZmap};
    {
        uint4* p = reinterpret_cast<uint4*>(smem);
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint32_t i = threadIdx.x; i < (kTableBytes + kZmapBytes) / 16; i += W * 64) p[i] = z;
        if (threadIdx.x == 0) *reinterpret_cast<uint4*>(smem + kEncSync + kSyD) = make_uint4(0u, kNone, 0u, 0u);
    }
    __syncthreads();
    uint32_t q[R], ra[R], mask[R], val[R];
    for (uint32_t r = wave; r < nrounds; r += W) {
        const uint8_t* p = src + (uint64_t)r * (R * kBlock);
        bool zero_entry = false;
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) q[j] = *reinterpret_cast<const uint32_t*>(p + j * kBlock + 4u * lane);
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            const uint32_t P = q[j] * kHashMul;
            const uint32_t sh = (P >> 12) & 16u;
            ra[j] = (P >> 15) & 0x1fffcu;
            mask[j] = 0xffffu << sh;
            val[j] = stored_entry(q[j], P) << sh;
            zero_entry |= val[j] == 0;
        }
        const uint32_t tokaddr = lane == 0 ? sy + kSyD : sy + kSySink + 4u * lane;
        pin_operands<R>(ra, mask, val);
        for (uint32_t spins = 0;;) {
            if (poll_word(sy + kSyD, r, 16)) break;
            const uint32_t D = rfl(lds_peek1(sy + kSyD));
            if (D == r) break;
            if (D == kPoison) wave_exit();
            backoff(r - D);
            watchdog(spins, sy, err, lane);
        }
        __builtin_amdgcn_s_setprio(3);
        exchange_tied<R>(ra, mask, val, tokaddr, r + 1u, false);
        __builtin_amdgcn_s_setprio(0);
        if (__builtin_expect(ballot64(zero_entry) != 0, 0)) {
#pragma unroll
            // (the zero quad in slot 0 included: here the mark also says "this chunk wrote the slot", which an entry of 0 alone does not;
            // nothing ever consults slot 0's mark)
            for (uint32_t j = 0; j < R; ++j) if (val[j] == 0) (void)zmap.test_and_set((q[j] * kHashMul) >> 16);
        }
    }
    wg_barrier();
    uint4* image = reinterpret_cast<uint4*>(images + chunk * kSegImageBytes);
    const uint4* lp = reinterpret_cast<const uint4*>(smem);
    for (uint32_t i = threadIdx.x; i < (kTableBytes + kZmapBytes) / 16; i += W * 64) image[i] = lp[i];
}

// Start images: slot by slot, the base image with the last-writer images of the following chunks laid over it one after the other
// (a slot counts as written by a chunk if its entry is non-zero or its zero-entry mark is set).  One thread per slot; the output
// marks are OR-ed into pre-zeroed words.
__global__ __launch_bounds__(256) void merge_images_kernel(const uint8_t* __restrict__ base, const uint8_t* __restrict__ lastwriters,
                                                           uint8_t* __restrict__ start, uint32_t count) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;                  // 0 .. 65535
    uint32_t e = reinterpret_cast<const uint16_t*>(base)[slot];
    uint32_t z = (reinterpret_cast<const uint32_t*>(base + kTableBytes)[slot >> 5] >> (slot & 31u)) & 1u;
    for (uint32_t k = 0; k < count; ++k) {
        uint8_t* out = start + (uint64_t)k * kSegImageBytes;
        reinterpret_cast<uint16_t*>(out)[slot] = (uint16_t)e;
        if (z) atomicOr(reinterpret_cast<uint32_t*>(out + kTableBytes) + (slot >> 5), 1u << (slot & 31u));
        if (k + 1 == count) break;                                                // (the last chunk has no successor: its last writers were never computed)
        const uint8_t* lw = lastwriters + (uint64_t)k * kSegImageBytes;
        const uint32_t le = reinterpret_cast<const uint16_t*>(lw)[slot];
        const uint32_t lz = (reinterpret_cast<const uint32_t*>(lw + kTableBytes)[slot >> 5] >> (slot & 31u)) & 1u;
        if (le != 0 || lz) { e = le; z = lz; }
    }
}

// where the streams of segments [first, first + count) go: one behind the other from *carry on, which moves to their end (write_buffer.rs:29-31's
// running total, a few dozen sizes at a time)
__global__ void scan_offsets_kernel(const uint64_t* __restrict__ sizes, uint32_t first, uint32_t count, uint64_t* __restrict__ carry, uint64_t* __restrict__ offsets) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t at = *carry;
    for (uint32_t i = 0; i < count; ++i) { offsets[first + i] = at; at += sizes[first + i]; }
    *carry = at;
}

// 16 bytes per lane: aligned stores, unaligned loads (the streams start at any even offset of the destination); head and tail by bytes
__global__ __launch_bounds__(256) void compact_bytes_kernel(const uint8_t* __restrict__ src, uint64_t src_stride, const uint64_t* __restrict__ sizes,
                                                            const uint64_t* __restrict__ offsets, uint8_t* __restrict__ dst) {
    typedef uint4 uint4_u __attribute__((aligned(1)));
    const uint64_t chunk = blockIdx.y;
    const uint64_t n = sizes[chunk];
    const uint8_t* s = src + chunk * src_stride;
    uint8_t* d = dst + offsets[chunk];
    uint64_t head = (16 - (reinterpret_cast<uintptr_t>(d) & 15)) & 15;
    if (head > n) head = n;
    const uint64_t vecs = (n - head) / 16;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < vecs; i += (uint64_t)gridDim.x * blockDim.x)
    {
        const uint4_u* sp = reinterpret_cast<const uint4_u*>(s + head + 16 * i);
        const uint4 v = {sp->x, sp->y, sp->z, sp->w};
        *reinterpret_cast<uint4*>(d + head + 16 * i) = v;
    }
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) d[threadIdx.x] = s[threadIdx.x];
        const uint64_t tail0 = head + 16 * vecs;
        if (tail0 + threadIdx.x < n && threadIdx.x < 16) d[tail0 + threadIdx.x] = s[tail0 + threadIdx.x];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// decode (index-fed): Codec::decode (codec/codec.rs:82-126), Chameleon::decode_plain / decode_map (chameleon.rs:56-68)
// ---------------------------------------------------------------------------------------------------------------
// The raw-copy bits of a chunk's block index against the blow-up protection (protection_state.rs:19-47, codec.rs:89-91), WITHOUT walking
// the FSM over the chunk: called for block i only where it is raw or incompressible (a coded record of 256 bytes or more: at most 4 MAP
// flags), it checks what the FSM implies locally —
//   * a run of raw blocks starts right behind a TRIGGER: an incompressible coded block whose nearest earlier coded block (looking through
//     a raw run) was incompressible too (:38-47: `update` is not called for raw blocks, `prev` survives them);
//   * every trigger is followed by a raw block (unless the chunk ends there);
//   * the run is `copy_penalty_start` blocks long (or cut by the chunk's end).  That value is 1 at the chunk's start, grows by one at the end of
//     every run (:30-35) and is halved at every 16th block while above 1 (:19-27) — so it follows from the PREVIOUS run alone (whose
//     length is its own value when it was triggered, checked by that run's thread), and is back at 1 if no run ended within the last 8
//     sixteen-block boundaries (a u8 halves to 1 in at most 8 steps).
// Every thread checks its own blocks against the index copy in LDS; all of them passing is equivalent to the FSM walk
// (tests/test_index_fsm_model.py holds the same rules, in numpy, against the oracle's FSM).  A chunk starts with a fresh FSM.
__device__ __forceinline__ bool index_fsm_consistent(const uint8_t* ix, uint32_t i, uint32_t nblk) {
    auto raw = [&](uint32_t b) -> bool { return (ix[b] & kIdxCopy) != 0; };
    auto inc = [&](uint32_t b) -> bool { return ix[b] <= 4u; };                   // coded, at most 4 MAP flags (a ragged block says 0x7f)
    auto mult16 = [](uint32_t lo, uint32_t hi) -> uint32_t { return hi / 16u + 1u - (lo + 15u) / 16u; };   // multiples of 16 in [lo, hi], lo <= hi + 1
    auto halve = [](uint32_t s, uint32_t k) -> uint32_t { const uint32_t h = k < 32u ? s >> k : 0u; return s > 1u ? (h ? h : 1u) : s; };
    if (!raw(i)) {
        // an incompressible coded block: a trigger iff the coded block before it was incompressible as well
        uint32_t u = i;
        while (u > 0 && raw(u - 1)) --u;                                          // (u - 1: the nearest earlier coded block, if any)
        const bool trigger = u > 0 && inc(u - 1);
        return !trigger || i + 1 >= nblk || raw(i + 1);
    }
    if (i > 0 && raw(i - 1)) return true;                                         // inside a run: the run's first block answers for it
    if (i == 0 || !inc(i - 1)) return false;                                      // a run must start behind an incompressible coded block ...
    const uint32_t t = i - 1;
    uint32_t u = t;
    while (u > 0 && raw(u - 1)) --u;
    if (u == 0 || !inc(u - 1)) return false;                                      // ... whose coded predecessor was incompressible too
    uint32_t L = 1;
    while (i + L < nblk && raw(i + L)) ++L;
    // copy_penalty_start when t triggered: from the previous run, if one ended within reach
    uint32_t s = 1;
    const uint32_t reach = t > 143u ? t - 143u : 0u;
    uint32_t e = t;                                                               // (search (reach, t) backwards for a raw block: the previous run's last)
    while (e > reach && !raw(e - 1)) --e;
    if (e > reach) {
        const uint32_t last = e - 1;
        uint32_t a = last;
        while (a > 0 && raw(a - 1) && last - a < 255u) --a;                       // its first block; its trigger is a - 1
        const uint32_t Lp = last - a + 1u;
        const uint32_t s_end = (halve(Lp, a <= last ? mult16(a, last) : 0u) + 1u) & 0xffu;   // halvings at the run's own blocks, then + 1 at its end
        s = halve(s_end, mult16(last + 1u, t));
    }
    return L == s || (L < s && i + L == nblk);
}

template <int R, int W, bool kProf, bool PAGED = false>
__global__ __launch_bounds__(W * 64) void chameleon_decode_rot(const uint8_t* __restrict__ in, const uint64_t* __restrict__ offsets,
                                                              const uint64_t* __restrict__ sizes, uint8_t* __restrict__ out,
                                                              uint64_t out_stride, uint64_t out_total, uint32_t flags,
                                                              const uint8_t* __restrict__ index, uint32_t* __restrict__ zmap_words,
                                                              uint64_t* __restrict__ produced, uint32_t* __restrict__ err, SegArgs seg,
                                                              uint64_t* __restrict__ prof) {
    static_assert((R == 8 || R == 12 || R == 16) && (W == 8 || W == 12 || W == 16), "round = 8, 12 or 16 records; 8, 12 or 16 waves");
    constexpr uint32_t kThreads = W * 64, kScanThreads = W == 16 ? 1024 : 512, kPerThread = kRotMaxBlocks / kScanThreads;   // position scan: 16 or 32 index entries per thread
    // flags: bit 0 = the output length is known exactly (container decode); bits 8..11 / 16..19 = how long a wave sleeps per hand-off still to
    // come / once it has seen the token reach its predecessor, in units of 64 cycles (the launcher's choice per round length)
    const uint32_t exact = flags & 1u, nap_far = (flags >> 8) & 15u, nap_near = (flags >> 16) & 15u;
    auto nap = [](uint32_t n) {                                                   // s_sleep takes an immediate: 64 cycles per unit, in binary
        if (n & 8u) __builtin_amdgcn_s_sleep(8);
        if (n & 4u) __builtin_amdgcn_s_sleep(4);
        if (n & 2u) __builtin_amdgcn_s_sleep(2);
        if (n & 1u) __builtin_amdgcn_s_sleep(1);
    };

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint64_t chunk = blockIdx.x;
    PhaseClock<kProf> clk(blockIdx.x == 0 ? prof : nullptr);   // phases: 0 stage A, 1 stage B, 2 operands, 3 D wait, 4 exchange, 5 quads, 6 zero-entry map, 7 stores + rotate
    const uint8_t* src = in + offsets[chunk];
    const uint8_t* idx = index + chunk * (out_stride / kBlock);                 // this chunk's slice of the block index (4-byte aligned: launcher)
    const uint64_t elen64 = sizes[chunk];
    uint8_t* dst = out + chunk * out_stride;
    const uint64_t room_all = out_total - chunk * out_stride;
    const uint64_t cap = room_all < out_stride ? room_all : out_stride;
    const uint32_t elen = elen64 > 0xfff00000ull ? 0xfff00000u : (uint32_t)elen64;   // 32-bit stream offsets in the pipeline; the in-order loop finishes longer streams
    const uint32_t nblk = (uint32_t)((cap + kBlock - 1) / kBlock);               // <= kRotMaxBlocks (launcher)
    constexpr uint32_t dSync = dec_sync_at(R);
    constexpr bool kZmapLds = dec_zmap_in_lds(R);
    typedef typename std::conditional<kZmapLds, ZmapLds, ZmapGlobal>::type Zmap;
    Zmap zmap;
    if constexpr (kZmapLds) zmap = Zmap{dec_zmap_at(R)}; else zmap = Zmap{zmap_words + chunk * (kZmapBytes / 4)};
    // the table sits at LDS address 0 (this kernel has no static LDS): slot addresses need no base
    const uint32_t sy = dSync;

    {   // fresh dictionary, this chunk's zero-entry map, the block index into LDS (a segment of a longer stream — SegArgs — starts from
        // the dictionary image it is given instead)
        uint4* p = reinterpret_cast<uint4*>(smem);
        const uint4 z = make_uint4(0, 0, 0, 0);
        const uint4* image = seg.init_images ? reinterpret_cast<const uint4*>(seg.init_images + chunk * kSegImageBytes) : nullptr;
        for (uint32_t i = threadIdx.x; i < kTableBytes / 16; i += kThreads) p[i] = image ? image[i] : z;
        if constexpr (kZmapLds) { for (uint32_t i = threadIdx.x; i < kZmapBytes / 16; i += kThreads) reinterpret_cast<uint4*>(smem + dec_zmap_at(R))[i] = image ? image[kTableBytes / 16 + i] : z; }
        else { for (uint32_t i = threadIdx.x; i < kZmapBytes / 16; i += kThreads) reinterpret_cast<uint4*>(zmap_words + chunk * (kZmapBytes / 4))[i] = image ? image[kTableBytes / 16 + i] : z; }
        const uint32_t* iw = reinterpret_cast<const uint32_t*>(idx);
        uint32_t* lw = reinterpret_cast<uint32_t*>(smem + kDecIdx);
        for (uint32_t i = threadIdx.x; i < kRotMaxBlocks / 4; i += kThreads) lw[i] = i < (nblk + 3u) / 4u ? iw[i] : 0x7f7f7f7fu;   // beyond the chunk: "ragged" = stop
        if (threadIdx.x == 0) {
            *reinterpret_cast<uint4*>(smem + dSync + kSyD) = make_uint4(0u, kNone, 0u, 0u);
            *reinterpret_cast<uint64_t*>(smem + dSync + kSyEnd) = ~0ull;
            if (lds_addr(smem) != 0) atomicOr(err, kErrWatchdog);                 // (cannot happen: see above)
        }
        if (threadIdx.x < W) { *reinterpret_cast<uint32_t*>(smem + dSync + kSyZdone + 4u * threadIdx.x) = 0u; *reinterpret_cast<uint32_t*>(smem + dSync + kSyZset + 4u * threadIdx.x) = 0u; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // the map is used through L2 atomics by this work-group only (chameleon.hip)
    }
    __syncthreads();

    uint32_t bad_index = 0;
    // ---- record positions of the whole chunk: one prefix sum over the index (consecutive entries per thread).  A record is
    // pipelined only if it is complete and followed by at least 2 more stream bytes (a MAP item is fetched as a dword); the first
    // one that is not (ragged block, end of the stream, end of the output, an index that disagrees with the stream length) and
    // everything behind it is finished by the in-order loop (codec.rs:102-123).
    {
        uint32_t* wave_sums = reinterpret_cast<uint32_t*>(smem + dSync + kSyWsum);
        const bool scans = threadIdx.x < kScanThreads;                            // (with 12 waves the first 8 do the scan)
        const uint32_t first = threadIdx.x * kPerThread;
        auto rec_len = [&](uint32_t ent) -> uint32_t { return (ent & kIdxCopy) ? kBlock : kSig + kBlock - 2u * (ent & 0x7fu); };
        uint32_t mine = 0;
        if (scans) {
#pragma unroll
            for (uint32_t k = 0; k < kPerThread; ++k) mine += rec_len(smem[kDecIdx + first + k]);
        }
        uint32_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = bperm(lane >= (uint32_t)d ? lane - d : lane, incl);
            if (lane >= (uint32_t)d) incl += o;
        }
        if (lane == 63 && scans) wave_sums[wave] = incl;
        __syncthreads();
        if (scans) {
            uint32_t pos = incl - mine;
            for (uint32_t w = 0; w < wave; ++w) pos += wave_sums[w];
            uint64_t stop_key = ~0ull;
#pragma unroll
            for (uint32_t k = 0; k < kPerThread; ++k) {
                const uint32_t i = first + k, ent = smem[kDecIdx + i], l = rec_len(ent);
                // the raw-copy flags must be what the blow-up protection would have decided (below): looked at only where a block is raw or incompressible
                if (exact && i < nblk && __builtin_expect((ent & kIdxCopy) != 0 || ent <= 4u, 0) && !index_fsm_consistent(smem + kDecIdx, i, nblk)) bad_index = 1;
                if (i % R == 0) *reinterpret_cast<uint32_t*>(smem + kDecPos + (i / R) * 4u) = pos;
                const bool stop = (ent & 0x7fu) == kIdxRagged || i >= nblk || ((uint64_t)i + 1) * kBlock > cap || pos >= elen || elen - pos < l + 2u;
                if (stop && stop_key == ~0ull) stop_key = ((uint64_t)i << 33) | ((uint64_t)((ent & kIdxCopy) && i < nblk ? 1u : 0u) << 32) | pos;
                pos += l;
            }
            if (threadIdx.x == kScanThreads - 1 && stop_key == ~0ull) stop_key = ((uint64_t)kRotMaxBlocks << 33) | pos;
            if (stop_key != ~0ull) atomicMin(reinterpret_cast<unsigned long long*>(smem + dSync + kSyEnd), (unsigned long long)stop_key);
        }
    }
    __syncthreads();
    // ---- PAGED (round 5): the stream lives in pages (include/density_hip.h); positions so far are positions in the STREAM.  The chunk's directory is
    // checked against them — a page starts at a multiple of 16 blocks, pages follow one another in block order, the stream position of a page's first
    // block is the bytes of the pages before it, no page holds more than a page, every page lies inside the container — and then every round's
    // position is turned into an offset from page 0; bit 0 (record positions are even) marks the rounds a page change falls into. ----
    uint32_t* pg_first = reinterpret_cast<uint32_t*>(smem + dec_pages_at(R));
    uint32_t* pg_delta = pg_first + kDecMaxPages;
    uint32_t n_pages = 0;
    if constexpr (PAGED) {
        const uint32_t* dirp = seg.page_dir + chunk * seg.page_dir_words;
        n_pages = rfl(dirp[0]);
        const bool dir_ok = n_pages >= 1 && n_pages <= kDecMaxPages && 4u * (n_pages + 1u) <= seg.page_dir_words;
        if (!dir_ok) n_pages = 0;
        uint32_t used = 0, page = 0, first = 0;
        if (threadIdx.x < n_pages) { const uint4 e = *reinterpret_cast<const uint4*>(dirp + 4u * (threadIdx.x + 1u)); page = e.x; first = e.y; used = e.z; pg_first[threadIdx.x] = first; pg_delta[threadIdx.x] = used; }
        if (threadIdx.x == 0) *reinterpret_cast<uint32_t*>(smem + dSync + kSyEnd + 8) = 0u;   // (the verdict word)
        __syncthreads();
        uint32_t before = 0;
        if (threadIdx.x < n_pages) {
            const uint32_t k = threadIdx.x;
            for (uint32_t m = 0; m < k; ++m) before += pg_delta[m];               // bytes of stream in the pages before this one
            bool ok = page < seg.page_limit && used <= kPageBytes && first % 16u == 0 && first < nblk && (k == 0 ? first == 0 : first > pg_first[k - 1]);
            if (ok) {
                // the stream position of block `first`: the position of its round and the index entries in front of it inside the round
                uint32_t at = *reinterpret_cast<const uint32_t*>(smem + kDecPos + (first / R) * 4u);
                for (uint32_t b = first / R * R; b < first; ++b) { const uint32_t ent = smem[kDecIdx + b]; at += (ent & kIdxCopy) ? kBlock : kSig + kBlock - 2u * (ent & 0x7fu); }
                ok = at == before;
            }
            // the pages hold the chunk's stream and nothing else: the last page ends where the size table says the stream ends — which also keeps
            // every stream position below `elen` inside a page of the directory (no read through a directory that is shorter than its stream)
            if (k + 1u == n_pages && (uint64_t)before + used != elen64) ok = false;
            if (!ok) bad_index = 1;
        }
        if (!dir_ok) bad_index = 1;
        if (bad_index) atomicOr(reinterpret_cast<uint32_t*>(smem + dSync + kSyEnd + 8), 1u);
        __syncthreads();
        if (threadIdx.x < n_pages) pg_delta[threadIdx.x] = (page << kPageShift) - before;
        const bool dead = *reinterpret_cast<const uint32_t*>(smem + dSync + kSyEnd + 8) != 0;   // a directory (or index) that lies: nothing is read through it
        __syncthreads();
        for (uint32_t x = threadIdx.x; x <= kRotMaxBlocks / R; x += kThreads) {
            const uint32_t b = x * R;
            uint32_t lo = 0, hi = n_pages ? n_pages - 1u : 0u;                     // the last page whose first block is <= b
            while (lo < hi) { const uint32_t mid = (lo + hi + 1u) >> 1; if (pg_first[mid] <= b) lo = mid; else hi = mid - 1u; }
            uint32_t* slot = reinterpret_cast<uint32_t*>(smem + kDecPos + x * 4u);
            const bool change = lo + 1u < n_pages && pg_first[lo + 1u] < b + R;
            *slot = dead ? 0u : (*slot + pg_delta[lo]) | (change ? 1u : 0u);
        }
        if (dead) { if (threadIdx.x == 0) { atomicOr(err, 8u); *reinterpret_cast<uint64_t*>(smem + dSync + kSyEnd) = 0; } }   // no record is followed: the in-order tail reports the rest
        __syncthreads();
    }
    const uint64_t end_key = *reinterpret_cast<const uint64_t*>(smem + dSync + kSyEnd);
    const uint32_t nvalid = rfl((uint32_t)(end_key >> 33));                      // records [0, nvalid) are complete and followed by more data
    const uint32_t npr = nvalid / R;                                              // whole rounds: these rotate; the rest (< R records + the ragged end) is the epilogue

    // ---- three-stage software pipeline per wave: A(x + 2W) signature loads | B(x + W) item loads | C(x) dictionary + stores ----
    // Per round in flight: lane j < R holds record j's position and (one 8-byte load) its signature; after stage B every lane
    // holds its R items and its R MAP/PLAIN flags (bit j of `hits`).  All rounds are whole, so every stage is straight-line code:
    // the loads of a stage leave back to back and nothing waits for a store.
    struct Meta { uint32_t posv, cnt; u32x2 sgv; uint32_t copy_mask; };
    // (Rounds past the end are clamped to the last one instead of skipped — a few redundant loads at the end of a chunk — so that the
    // number and order of memory operations per iteration is fixed and the compiler's waits count exactly.)
    auto stage_a = [&](uint32_t xr, Meta& m) {                                   // positions of round x; signatures requested
        const uint32_t x = xr < npr ? xr : npr - 1u;
        const uint32_t e = smem[kDecIdx + x * R + (lane < R ? lane : 0u)];        // lane j < R: entry of record j
        uint32_t base = rfl(lds_peek1(kDecPos + x * 4u));
        const uint32_t mylen = (e & kIdxCopy) ? kBlock : kSig + kBlock - 2u * (e & 0x7fu);
        uint32_t hop = 0;                                                         // PAGED: what the records behind a page change inside this round are further on
        if constexpr (PAGED) {
            if (__builtin_expect(base & 1u, 0)) {                                 // (a page change falls into this round: some forty times per 4 MiB chunk)
                uint32_t k = 0;
                while (k + 1u < n_pages && pg_first[k + 1u] <= x * R) ++k;        // the page of the round's first record; the next one starts inside the round
                const uint32_t j0 = pg_first[k + 1u] - x * R;
                hop = lane >= j0 ? pg_delta[k + 1u] - pg_delta[k] : 0u;
                base &= ~1u;
            }
        }
        uint32_t incl = mylen;                                                    // prefix within rows of 16 lanes
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
        incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
        if (R > 8) incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
        m.posv = base + incl - mylen + hop;
        m.copy_mask = (uint32_t)ballot64((e & kIdxCopy) != 0 && lane < R);
        m.sgv = *reinterpret_cast<const u32x2_u*>(src + ((lane < R && !(e & kIdxCopy)) ? m.posv : base));   // codec.rs:28-31 (idle lanes: any valid address)
        m.cnt = e & 0x7fu;                                                        // the entry's MAP count: checked against the signature in stage B
    };
    const uint32_t minus_2lane = 0u - 2u * lane;
    auto stage_b = [&](const Meta& m, uint32_t& hits, uint32_t (&item)[R]) {    // signatures -> MAP/PLAIN flags, item loads
        // the index must agree with the stream it describes: a record's MAP count is its signature's popcount (lane j < R: record j)
        bad_index |= (lane < R && !((m.copy_mask >> lane) & 1u) && (uint32_t)(__builtin_popcount(m.sgv.x) + __builtin_popcount(m.sgv.y)) != m.cnt) ? 1u : 0u;
        hits = 0;
        // (lane j < R prepares record j for all lanes at once — a raw record has no signature: no MAP flags, its 256 bytes are its "items" —
        // so that the loop below is three lane reads per record and no scalar arithmetic)
        const uint32_t codedv = ((m.copy_mask >> lane) & 1u) ? 0u : ~0u;             // all ones, or 0 for 256 raw bytes without a signature (codec.rs:89-91)
        const uint32_t sxv = m.sgv.x & codedv, syv = m.sgv.y & codedv, pbv = m.posv + (codedv & kSig);
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {                                        // (straight-line: selects, no branches)
            const uint32_t slo = rlane_u(sxv, (int)j), shi = rlane_u(syv, (int)j), pos = rlane_u(pbv, (int)j);
            uint32_t bit;                                                         // this lane's flag: one select on the signature as a lane mask
            asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(bit) : "s"(((uint64_t)shi << 32) | slo));
            hits |= bit << j;
            // this lane's item sits 4 bytes further per PLAIN lane below it and 2 per MAP lane: 4*lane - 2*(MAP lanes below), from the record's items on
            const uint32_t t = __builtin_amdgcn_mbcnt_hi(shi, __builtin_amdgcn_mbcnt_lo(slo, minus_2lane));   // MAP lanes below - 2*lane
            uint32_t off;                                                         // (one multiply-add — left to the compiler: a shift pair and a subtract; the stream's base is the load's scalar operand)
            asm("v_mad_i32_i24 %0, %1, -2, %2" : "=v"(off) : "v"(t), "s"(pos));
            item[j] = ld32u(src + off);
        }
    };

    Meta ma, mb, mc;
    ma.posv = mb.posv = mc.posv = 0; ma.cnt = mb.cnt = mc.cnt = 0; ma.copy_mask = mb.copy_mask = mc.copy_mask = 0;
    ma.sgv = mb.sgv = mc.sgv = u32x2{0u, 0u};
    uint32_t itemb[R], itemc[R], hitsb = 0, hitsc = 0;
#pragma unroll
    for (uint32_t j = 0; j < R; ++j) { itemb[j] = 0; itemc[j] = 0; }
    // prologue: B(w) needs A(w); A(w + W) goes out behind it
    if (npr) {
        stage_a(wave, mb);
        stage_b(mb, hitsc, itemc);
        mc = mb;
        stage_a(wave + W, mb);
        // Everything asked for so far is waited for HERE, once, visibly to the compiler: with nothing pending at the top of the loop the
        // waits it places inside count from the loop's own order of loads and stores (a signature load is followed by the round's 12
        // record stores, so the next round's stage B waits for "all but the last 12"); with loads still pending from out here it would
        // settle for the common bound of both ways in — zero — and every round would begin by waiting for its predecessor's stores.
        asm volatile("" : : "v"(mb.sgv.x), "v"(mb.sgv.y), "v"(mb.posv), "v"(mc.sgv.x), "v"(mc.sgv.y));
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) asm volatile("" : : "v"(itemc[j]));
    }

    uint32_t ra[R], mask[R], val[R];
    for (uint32_t x = wave; x < npr; x += W) {
        clk.start();
        __builtin_amdgcn_s_setprio(1);                                   // (priorities: see the encoder's exchange)
        // (B first: what it waits for — the signatures requested one iteration ago — is older than anything issued since, so the
        // wait does not cover a load that has just left)
        stage_b(mb, hitsb, itemb);
        clk.mark(1);
        stage_a(x + 2 * W, ma);
        clk.mark(0);

        // ---- C: operands of the dictionary step ----
        // (Instruction count is this kernel's time: a wave whose iteration is longer than W hand-offs arrives late for its turn, and every
        // late arrival stalls the chain — six instructions per record less made the kernel 17 % faster.  Hence: the loop below treats every
        // record as coded and a rare branch behind it takes the raw-copy records' operands back (a chunk's cold start; incompressible
        // data), and the rare zero-entry candidates cost one compare per record each way, their lanes collected in scalar registers.)
        const uint32_t coded_mask = ((1u << R) - 1u) & ~mc.copy_mask;             // records that go through the dictionary
        const uint32_t hit_mask = seg.lastwriters_only ? 0u : hitsc;              // MAP quads that are looked up (raw records have no hit bits: stage B)
        // zplain: lanes with a zero-entry CANDIDATE that writes — a PLAIN quad whose stored entry is 0 (those that read 0: zm[] below)
        uint64_t zplain = 0;
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            const uint32_t qv = itemc[j];
            const uint32_t P = qv * kHashMul;
            // MAP: the item is the slot (chameleon.rs:64-68); PLAIN: the upper half of the hash product — one select with a half-word pick per
            // side; `em`: 0xffff for the lanes that write (PLAIN: chameleon.rs:56-61), 0 for those that only read (MAP); `mm`: the MAP lanes
            uint32_t h, em;
            uint64_t mm;
            asm("v_and_b32 %0, %5, %3\n\t"
                "v_cmp_ne_u32 vcc, 0, %0\n\t"
                "v_cndmask_b32_sdwa %0, %4, %6, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n\t"
                "v_cndmask_b32 %1, %7, 0, vcc\n\t"
                "s_mov_b64 %2, vcc"
                : "=&v"(h), "=&v"(em), "=s"(mm) : "v"(hitsc), "v"(P), "n"(1u << j), "v"(qv), "v"(0xffffu) : "vcc");
            const uint32_t sh = h << 4;                                           // (a shift takes the low five bits of its count: (h & 1) << 4)
            // stored_entry(qv, P) for the lanes that write, 0 for a MAP lane (`em` is 0xffff or 0: the salt needs no mask of its own)
            const uint32_t e = (((P & 0xfffeu) | (qv >> 31)) ^ __umul24(P >> 16, kSaltMul)) & em;
            ra[j] = (h >> 1) << 2;
            mask[j] = em << sh;
            val[j] = e << sh;
            zplain |= ballot64(e == 0) & ~mm;                                     // a PLAIN quad whose stored entry is 0 (one compare; the rest is scalar)
        }
        // A round that will MARK the zero-entry map (a PLAIN quad whose entry is 0: about four per 4 MiB of text) says so before its exchanges:
        // rounds behind it that only LOOK a slot up in the map (every recurrence of such a quad: one round in 25) then wait for nothing but
        // earlier rounds that have said so — almost never — instead of for every earlier round to finish.
        // (round 5) ... unless it is the ZERO quad, whose entry 0 in slot 0 is no candidate (a 0 there IS the zero quad, written or not): the first
        // zero quad behind anything else that hashed to slot 0 — once per incompressible patch of mixed data — used to announce a mark, and a
        // marking round waits for every earlier round to be through.  Looked at exactly, in a rare branch (a last-writers pass does mark slot 0:
        // there the mark says "written").
        uint64_t zreal = zplain;
        if (__builtin_expect(zplain != 0, 0) && !seg.lastwriters_only) {
            zreal = 0;
#pragma nounroll
            for (uint32_t j = 0; j < R; ++j) {                                    // (rolled, over select chains, like every rare path of this kernel)
                const uint32_t it = pick<R>(itemc, j), P = it * kHashMul;
                zreal |= ballot64(!((hitsc >> j) & 1u) && ((coded_mask >> j) & 1u) && (P >> 16) != 0 && stored_entry(it, P) == 0);
            }
        }
        const bool marks = zreal != 0;
        if (__builtin_expect(marks, 0)) { if (lane == 0) lds_poke(sy + kSyZset + 4u * wave, x + 1u); }
        if (__builtin_expect(mc.copy_mask != 0, 0)) {
            // raw-copy records (codec.rs:89-91) touch no state: their lanes read a harmless conflict-free word instead
#pragma unroll
            for (uint32_t j = 0; j < R; ++j) {
                const bool raw = (mc.copy_mask >> j) & 1u;
                ra[j] = raw ? 4u * lane : ra[j];
                mask[j] = raw ? 0u : mask[j];
                val[j] = raw ? 0u : val[j];
            }
        }
        const uint32_t tokaddr = lane == 0 ? sy + kSyD : sy + kSySink + 4u * lane;
        uint32_t tokval = x + 1u;                                                 // (in its register before the wait: nothing but the priority change between the token and the exchanges)
        asm volatile("" : "+v"(tokval));
        pin_operands<R>(ra, mask, val);                                           // complete before the wait for the token
        clk.mark(2);
        clk.stamp(x, 0, lane);
        __builtin_amdgcn_s_setprio(2);
        // ---- D chain ----
        // (Every poll is an LDS instruction in the queue the token holder's exchanges go through.  A wave two or more turns away sleeps for most
        // of the hand-offs still to come — one takes 600 cycles and more —, the next in line polls; when it has SEEN the token reach its
        // predecessor it sleeps through the first part of that critical section too.)
        for (uint32_t spins = 0, seen = ~0u;;) {
            const uint32_t D = rfl(lds_peek1(sy + kSyD));
            if (D == x) break;
            if (D == kPoison) wave_exit();
            const uint32_t dist = x - D;
            // (a hand-off is ~480 cycles + ~19 per record — profiles/r04_*: 690 for rounds of 12, 780 for 16; the sleeps cover about half of one)
            if (dist >= 2) { for (uint32_t k = 1; k < dist && k < 6; ++k) nap(nap_far); }   // 320 cycles (rounds of 12) per hand-off to come
            else {
                if (seen != ~0u && seen != D) nap(nap_near);                      // 192 cycles of a critical section of 450 and more (12 records)
                if (poll_word(sy + kSyD, x, 8)) break;
            }
            seen = D;
            watchdog(spins, sy, err, lane);
        }
        clk.mark(3);
        clk.stamp(x, 1, lane);
        __builtin_amdgcn_s_setprio(3);
        exchange_tied<R>(ra, mask, val, tokaddr, tokval, false);
        __builtin_amdgcn_s_setprio(0);
        clk.mark(4);
        clk.stamp(x, 2, lane);

        // ---- what each slot holds at this lane's turn -> quads (in place of the answers) ----
        // (a MAP quad that read 0 — never written, or a genuine zero entry? — is a lane of zm[j]: the compare costs what the running minimum
        // it replaces cost, its answer lands in scalar registers, and the rare path below knows record and lanes without working them out again)
        uint64_t zany = 0;
        uint32_t zrec = 0;                                                        // the records that have such a lane: one scalar bit per record (a lane mask per record was 2 R scalar registers)
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) {
            const bool maps = (hit_mask >> j) & 1u;
            const uint32_t h = itemc[j] & 0xffffu;
            const uint32_t cur = __builtin_amdgcn_ubfe(ra[j], itemc[j] << 4, 16);  // the slot's half of the word ((h & 1) << 4: a bit-field offset is five bits)
            const uint64_t mm = ballot64(maps);                                   // the MAP lanes as a lane mask: for the select below and, in scalar registers, for
            const uint64_t zj = ballot64(cur == 0) & mm;                          // "MAP of a slot holding 0": never written, or a genuine zero entry?
            zany |= zj;
            zrec |= (zj != 0 ? 1u : 0u) << j;
            const uint32_t mq = entry_to_quad(h, cur);                            // (for every lane, then one select: cheaper than an exec mask around it)
            asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(ra[j]) : "v"(itemc[j]), "v"(mq), "s"(mm));
        }
        clk.mark(5);
        // ---- zero-entry map, in stream order (rare: stored entries are salted).  A round with no such quad only reports "done"; one that
        // has any first waits until every earlier round has reported (wave w' owns the rounds = w' mod W). ----
        if (__builtin_expect(marks || zany != 0, 0)) {
          if (!marks) {
            // (round 5) Slot 0 needs no map: its entry 0 IS the zero quad, written or not (chameleon.rs:41,88-100) — and the zero quad is the
            // commonest quad of real data (zero pages, padding).  Its MAP lanes read 0 like a candidate's, so all-zero input took every record
            // of every round through the select chains below, low-entropy data every other round.  Here, still in the rare branch (rounds
            // without any 0 read never get here), the candidates are made exact again: a MAP lane that read 0 — its quad is the one an entry of
            // 0 stands for in its slot, entry -> quad being one-to-one per slot — in a slot other than 0.
            uint32_t real = 0;
            for (uint32_t zb = zrec; zb; zb &= zb - 1u) {                         // (rolled, over select chains)
                const uint32_t j = (uint32_t)__builtin_ctz(zb);
                const uint32_t h = pick<R>(itemc, j) & 0xffffu;
                real |= (ballot64(((hit_mask >> j) & 1u) != 0 && h != 0 && pick<R>(ra, j) == entry_to_quad(h, 0)) != 0 ? 1u : 0u) << j;
            }
            zrec &= real;
            if (zrec == 0) zany = 0;
          }
          if (marks || zany != 0) {
            clk.note(x, 1, lane);
            for (uint32_t spins = 0;;) {
                const uint32_t wv = lane % W;
                if (marks) {
                    // marks must not be seen by look-ups of earlier rounds: every earlier round has finished its zero-entry phase
                    const uint32_t d = (wave + W - wv) % W;                       // wave wv's last round before x is x - d
                    const uint32_t done = lds_peek1(sy + kSyZdone + 4u * wv);    // (rounds finished: last round + 1)
                    if (ballot64(d != 0 && x >= d && done < x - d + 1u) == 0) break;
                } else {
                    // look-ups only: the marks of earlier rounds must be in — those rounds said so before their exchanges, i.e. before ours
                    const uint32_t pending = lds_peek1(sy + kSyZset + 4u * wv);  // (round + 1, 0: none)
                    if (ballot64(pending != 0 && pending - 1u < x) == 0) break;
                }
                if (rfl(lds_peek1(sy + kSyD)) == kPoison) wave_exit();
                watchdog(spins, sy, err, lane);
            }
            if (kZmapLds && !marks) {
                // Look-ups only (about one round in 25 on repetitive text: every recurrence of a quad whose entry is 0): nothing in this round
                // changes the map, so its look-ups need no order among themselves — all lanes of a record at once, usually one lane of one record
                // The records concerned — usually one — one by one, in a ROLLED loop over select chains.  (Round 4: every rare path of this kernel
                // is rolled now.  Unrolled, their per-record temporaries were all live at once and set the kernel's register need — 160 for rounds
                // of 12, spills for anything longer — although the common path needs ~120; rolled, rounds of 16 and 20 fit 12 waves' 168.)  Which of
                // the record's lanes read 0 is worked out again: the quad such a lane holds is the one an entry of 0 stands for in its slot, and
                // entry -> quad is one-to-one per slot.
                for (uint32_t zb = zrec; zb; zb &= zb - 1u) {
                    const uint32_t j = (uint32_t)__builtin_ctz(zb);
                    const uint32_t it = pick<R>(itemc, j), an = pick<R>(ra, j);
                    const uint32_t h = it & 0xffffu;
                    const bool t = ((hit_mask >> j) & 1u) && an == entry_to_quad(h, 0) && h != 0;   // (slot 0: "never written" and its zero entry both stand for the zero quad)
                    uint32_t bit = 1;
                    if (t) bit = zmap.test(h);
                    const uint32_t outv = (t && !bit) ? 0u : an;                  // chameleon.rs:64-68 on a never-written (zero) word
#pragma unroll
                    for (uint32_t k = 0; k < R; ++k) {
                        uint32_t jj = j;
                        asm volatile("" : "+s"(jj));                              // (opaque, as in pick)
                        ra[k] = jj == k ? outv : ra[k];
                    }
                }
            } else {
            // which records have such a quad — from what is still in registers, a few instructions per record — then those records one by one, usually one
            uint32_t zblocks = 0;
#pragma nounroll
            for (uint32_t j = 0; j < R; ++j) {                                    // (rolled, over select chains: see above)
                // a PLAIN quad with stored entry 0 (from the item again: the exchange operands are dead by now, and keeping them alive for this path
                // cost the common one registers), or a MAP quad whose slot gave the quad that an entry of 0 stands for
                const uint32_t it = pick<R>(itemc, j), an = pick<R>(ra, j);
                const bool wrote0 = !((hitsc >> j) & 1u) && ((coded_mask >> j) & 1u) && stored_entry(it, it * kHashMul) == 0;
                const bool read0 = ((hit_mask >> j) & 1u) && an == entry_to_quad(it & 0xffffu, 0);
                zblocks |= (ballot64(wrote0 || read0) != 0 ? 1u : 0u) << j;
            }
            zblocks &= coded_mask;
            for (uint32_t zb = zblocks; zb; zb &= zb - 1u) {
                const uint32_t j = (uint32_t)__builtin_ctz(zb);
                const bool coded = (coded_mask >> j) & 1u;
                const bool hit = (hitsc >> j) & 1u;
                const uint32_t qv = pick<R>(itemc, j), cur = pick<R>(ra, j);
                const uint32_t P = qv * kHashMul;
                const uint32_t h = hit ? (qv & 0xffffu) : (P >> 16);
                const bool zset = coded && !hit && stored_entry(qv, P) == 0 && (h != 0 || seg.lastwriters_only);
                const bool ztest = coded && hit && h != 0 && cur == entry_to_quad(h, 0) && !seg.lastwriters_only;
                uint64_t todo = ballot64(zset || ztest);
                uint32_t out = cur;
                while (todo) {                                                    // ascending lane == stream order
                    const uint32_t l = (uint32_t)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    if (lane == l) {
                        if (zset) zmap.set(h);
                        else if (!zmap.test(h)) out = 0;                          // chameleon.rs:64-68 on a never-written (zero) word
                    }
                }
#pragma unroll
                for (uint32_t k = 0; k < R; ++k) {
                    uint32_t jj = j;
                    asm volatile("" : "+s"(jj));                                  // (opaque, as in pick)
                    ra[k] = jj == k ? out : ra[k];
                }
            }
            }
          }
        }
        if (lane == 0) {
            if (marks) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lds_poke(sy + kSyZset + 4u * wave, 0u); }   // (behind the marks: LDS operations of a wave execute in order; in L2: ZmapGlobal::set consumed the atomics' answers)
            lds_poke(sy + kSyZdone + 4u * wave, x + 1u);
        }
        clk.mark(6);

        // ---- stores: 256 coalesced bytes per record ----
        uint8_t* base = dst + (uint64_t)x * R * kBlock;
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) *reinterpret_cast<uint32_t*>(base + j * kBlock + 4u * lane) = ra[j];   // (a last-writers pass stores too: a branch here would cost stage B its exact waits)
        // ---- rotate the pipeline ----
        mc = mb; mb = ma; hitsc = hitsb;
#pragma unroll
        for (uint32_t j = 0; j < R; ++j) itemc[j] = itemb[j];
        clk.mark(7);
        clk.stamp(x, 3, lane);
    }
    clk.flush(wave, lane);

    if (ballot64(bad_index != 0) != 0 && lane == 0) atomicOr(err, 8u);             // (lane j < R holds record j's verdict: any lane's counts)
    wg_barrier();
    // ---- epilogue on one wave, in order: the records of the last, partial round — one call per record, the block's raw-copy flag
    // from the index standing in for the FSM — then the ragged end of the stream (codec.rs:102-123) ----
    if (wave == 0) {
        Guard g;
        // (PAGED: no page starts inside this tail — the encoder keeps room for it in the last round's page —, so one offset turns its stream positions
        // into offsets from page 0; a directory that says otherwise is malformed)
        uint32_t tail_delta = 0;
        bool bad = false;
        if constexpr (PAGED) {
            if (n_pages == 0 || *reinterpret_cast<const uint32_t*>(smem + dSync + kSyEnd + 8) != 0 || pg_first[n_pages - 1u] > npr * R) bad = true;
            else tail_delta = pg_delta[n_pages - 1u];
        }
        const uint32_t end_at = (uint32_t)end_key + tail_delta;
        // (32-bit arithmetic like every page offset: the last page may lie BELOW the stream bytes in front of it — a producer may number its pages in any
        // order; this library's encoder never does, its counter only grows —, and the sum must wrap like the offsets it is compared with)
        const uint64_t elen_at = (uint32_t)((uint32_t)elen64 + tail_delta);
        uint64_t ip = npr * R < nvalid ? (rfl(lds_peek1(kDecPos + npr * 4u)) & (PAGED ? ~1u : ~0u)) : end_at, op = (uint64_t)npr * R * kBlock;
        for (uint32_t i = npr * R; i < nvalid && !bad; ++i) {
            const uint32_t ent = smem[kDecIdx + i];
            const uint64_t rec_end = ip + ((ent & kIdxCopy) ? kBlock : kSig + kBlock - 2u * (ent & 0x7fu));
            g.penalty = (ent & kIdxCopy) ? 1u : 0u; g.start = 1; g.prev = 0; g.counter = 1;
            // The record's signature must say what the index says (as the rotating rounds check it): lengths alone do not — a corrupted signature with MORE
            // MAP flags makes the record 8 or more bytes shorter than the index has it, and the bytes left over pass for a signature with no items behind it
            // (codec.rs:102-123 on an exhausted buffer), where the reference reads the next record from the wrong place (tools/gpu_fuzz_tail.py, round 6).
            if (!(ent & kIdxCopy)) {
                if (rec_end > elen_at || ip + kSig > rec_end) bad = true;
                else {
                    const uint64_t sig = (uint64_t)rfl(ld32u(src + ip)) | ((uint64_t)rfl(ld32u(src + ip + 4)) << 32);
                    if ((uint32_t)__builtin_popcountll(sig) != (ent & 0x7fu)) bad = true;
                }
                if (bad) break;
            }
            bad = !decode_in_order(src, rec_end, dst, cap, g, ip, op, 0u, zmap, lane, seg.lastwriters_only != 0) || ip != rec_end;
        }
        g.penalty = (uint32_t)(end_key >> 32) & 1u; g.start = 1; g.prev = 0; g.counter = 1;    // the stopping block's raw-copy flag is all that is left of the FSM
        if (!bad && (ip != end_at || op != (uint64_t)nvalid * kBlock)) bad = true;
        if (!bad) bad = !decode_in_order(src, elen_at, dst, cap, g, ip, op, 0u, zmap, lane, seg.lastwriters_only != 0);
        if (exact && !bad && op != cap) bad = true;
        if (lane == 0) {
            produced[chunk] = op;
            if (bad) atomicOr(err, 1u);
        }
    }
    if (seg.final_images) {                                                        // the dictionary as this chunk leaves it
        __threadfence();
        wg_barrier();
        uint4* image = reinterpret_cast<uint4*>(seg.final_images + chunk * kSegImageBytes);
        const uint4* p = reinterpret_cast<const uint4*>(smem);
        for (uint32_t i = threadIdx.x; i < kTableBytes / 16; i += kThreads) image[i] = p[i];
        if constexpr (kZmapLds) {
            for (uint32_t i = threadIdx.x; i < kZmapBytes / 16; i += kThreads) image[kTableBytes / 16 + i] = reinterpret_cast<const uint4*>(smem + dec_zmap_at(R))[i];
        } else {
            const uint32_t* zw = zmap_words + chunk * (kZmapBytes / 4);
            for (uint32_t i = threadIdx.x; i < kZmapBytes / 16; i += kThreads)
                image[kTableBytes / 16 + i] = make_uint4(__hip_atomic_load(zw + 4 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(zw + 4 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                                          __hip_atomic_load(zw + 4 * i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(zw + 4 * i + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Start-up self-test of what these kernels assume about the LDS (density_hip_selftest / acquire_ctx):
//  (1) ds_mskor_rtn_b32 services the lanes of one instruction in ascending lane order — same half-dword, alternating halves of
//      one dword, mask-0 readers between writers, back-to-back instructions;
//  (2) a plain ds_write_b32 issued behind a wave's exchanges is not visible before them (the token hand-off), checked by 16
//      waves rotating exactly like the kernels do, all hammering the same few dictionary slots;
//  (3) the lane-reversed ds_write_b16 restores a block (rollback).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kRotThreads) void rotor_selftest_kernel(uint32_t* __restrict__ fail, uint32_t tune) {
    __shared__ __attribute__((aligned(16))) uint32_t cell[64];
    __shared__ __attribute__((aligned(16))) uint32_t syn[kSyBytes / 4];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    uint32_t bad = 0;
    if (threadIdx.x < 64) cell[threadIdx.x] = 0;
    if (threadIdx.x == 0) { syn[kSyD / 4] = 0; syn[kSyD / 4 + 1] = kNone; }
    __syncthreads();
    const uint32_t c0 = lds_addr(cell), sy = lds_addr(syn);
    if (wave == 0) {
        // (1a) all lanes, one half-dword, two instructions back to back: lane l must get lane l-1's entry
        uint32_t r0, r1, r2, r3;
        const uint32_t e0 = (lane + 1u) << 16, e1 = (lane + 101u) << 16;
        asm volatile("ds_mskor_rtn_b32 %0, %2, %3, %4\n\tds_mskor_rtn_b32 %1, %2, %3, %5\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1) : "v"(c0), "v"(0xffff0000u), "v"(e0), "v"(e1) : "memory");
        if ((r0 >> 16) != lane) bad |= 1u;
        if ((r1 >> 16) != (lane == 0 ? 64u : lane + 100u)) bad |= 2u;
        // (1b) alternating halves of one dword: even lanes the low half, odd lanes the high half; each half is its own chain
        const uint32_t sh = (lane & 1u) << 4;
        asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r2) : "v"(c0 + 4u), "v"(0xffffu << sh), "v"((lane + 1u) << sh) : "memory");
        if (((r2 >> sh) & 0xffffu) != (lane < 2 ? 0u : lane - 1u)) bad |= 4u;
        // (1c) mask-0 readers between writers: lanes = 3 (mod 4) write, the others read what the last writer below them left
        const bool wr = (lane & 3u) == 3u;
        asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r3) : "v"(c0 + 8u), "v"(wr ? 0xffffu : 0u), "v"(wr ? lane + 1u : 0u) : "memory");
        if ((r3 & 0xffffu) != (lane < 4 ? 0u : (lane & ~3u))) bad |= 8u;
        // (3) rollback of (1a)'s second instruction, then of its first: the cell must read 0 again
        uint32_t back;
        dict_store(bperm(63u - lane, c0 + 2u), bperm(63u - lane, r1 >> 16));
        dict_store(bperm(63u - lane, c0 + 2u), bperm(63u - lane, r0 >> 16));
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(back) : "v"(c0) : "memory");
        if (back != 0) bad |= 16u;
    }
    __syncthreads();
    // (2) token rotation: round r (wave r % 16) appends to four chains (two dwords x two halves); the value a lane gets back
    // must be the entry of its predecessor in that chain: previous lane of the chain, previous block, previous round
    {
        const uint32_t chain = lane & 3u;                                         // dword (chain >> 1), half (chain & 1)
        const uint32_t a = c0 + 16u + 4u * (chain >> 1), sh = (chain & 1u) << 4;
        uint32_t addr[kR], mask[kR], val[kR], ret[kR];
        constexpr uint32_t kRounds = 192;
        for (uint32_t r = wave; r < kRounds; r += kRotWaves) {
#pragma unroll
            for (uint32_t j = 0; j < kR; ++j) {
                addr[j] = a; mask[j] = 0xffffu << sh;
                val[j] = ((((r * kR + j) * 16u + (lane >> 2)) + 1u) & 0xffffu) << sh;   // position in the chain + 1 (mod 2^16)
            }
            for (uint32_t spins = 0;; ++spins) {
                const uint32_t D = rfl(lds_peek2(sy + kSyD).x);
                if (D == r) break;
                if (D == kPoison) wave_exit();
                if (spins > kSpinLimit) { if (lane == 0) { atomicOr(fail, 64u << 8); lds_poke(sy + kSyD, kPoison); } wave_exit(); }
            }
            exchange_round(ret, addr, mask, val, lane == 0 ? sy + kSyD : sy + kSySink + 4u * lane, r + 1u, (tune & 1u) != 0);
#pragma unroll
            for (uint32_t j = 0; j < kR; ++j) {
                const uint32_t want = ((r * kR + j) * 16u + (lane >> 2)) & 0xffffu;
                if (((ret[j] >> sh) & 0xffffu) != want) bad |= 32u;
            }
        }
    }
    if (bad) atomicOr(fail, bad << 8);                                           // (bits 0..7 belong to container.hip's selftest_kernel)
}

// ---------------------------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------------------------
namespace {
// DENSITY_HIP_PROF=1: per-wave, per-phase cycle accounting of work-group 0, printed to stderr after every launch (synchronises)
constexpr size_t kProfWords = 128 + 5 * kProfRounds + 8;
uint64_t* rot_prof_buffer() {
    static uint64_t* buf = nullptr;
    if (!debug_env("DENSITY_HIP_PROF")) return nullptr;
    if (!buf && hipMalloc((void**)&buf, kProfWords * sizeof(uint64_t)) != hipSuccess) buf = nullptr;
    if (buf) { (void)hipDeviceSynchronize(); (void)hipMemset(buf, 0, kProfWords * sizeof(uint64_t)); (void)hipDeviceSynchronize(); }
    return buf;
}
void rot_prof_report(const char* what, const char* phases, uint64_t* buf, hipStream_t stream, uint32_t waves = 8) {
    if (!buf) return;
    static uint64_t h[kProfWords];
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(h, buf, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return;
    if (const char* dump = debug_env("DENSITY_HIP_PROF_DUMP")) {                     // raw buffer, for offline analysis: <prefix>.<encode|decode>.bin
        char path[512];
        snprintf(path, sizeof(path), "%s.%s.bin", dump, what);
        if (FILE* f = fopen(path, "wb")) { fwrite(h, sizeof(uint64_t), kProfWords, f); fclose(f); }
    }
    fprintf(stderr, "[density_hip prof] %s work-group 0, kcycles per wave by phase (%s)\n", what, phases);
    {
        const uint64_t* c = h + 128 + 5 * kProfRounds;
        fprintf(stderr, "[density_hip prof]   events: fast rounds %llu, ordered rounds held %llu (%llu of them ran ahead) / taken back %llu, rounds walked in order %llu, aborts raised %llu\n",
                (unsigned long long)c[0], (unsigned long long)c[1], (unsigned long long)c[5], (unsigned long long)c[2], (unsigned long long)c[3], (unsigned long long)c[4]);
        fprintf(stderr, "[density_hip prof]   memo of predictions: %llu found in the early copy, %llu on a second look\n", (unsigned long long)c[6], (unsigned long long)c[7]);
    }
    for (int w = 0; w < 16; ++w) {
        uint64_t tot = 0;
        for (int k = 0; k < 8; ++k) tot += h[8 * w + k];
        fprintf(stderr, "[density_hip prof]   w%-2d", w);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %7.1f", (double)h[8 * w + k] / 1e3);
        fprintf(stderr, "  | total %8.1f\n", (double)tot / 1e3);
    }
    // D chain: hop = token seen (round r) - token seen (round r-1); critical = exchanges + token; detect = seen - max(previous token done, own arrival)
    const uint64_t* ts = h + 128;
    uint32_t n = 0, late = 0;
    double hop = 0, crit = 0, det = 0, lateness = 0;
    uint64_t hop_max = 0;
    uint32_t hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t r = 17; r < kProfRounds; ++r) {
        if (!ts[4 * r + 1] || !ts[4 * r + 2] || !ts[4 * (r - 1) + 1] || !ts[4 * (r - 1) + 2]) continue;
        const uint64_t seen = ts[4 * r + 1], prev_seen = ts[4 * (r - 1) + 1], prev_done = ts[4 * (r - 1) + 2], arrive = ts[4 * r];
        if (seen < prev_seen) continue;
        const uint64_t hp = seen - prev_seen;
        hop += (double)hp; crit += (double)(ts[4 * r + 2] - seen);
        hop_max = hp > hop_max ? hp : hop_max;
        const uint64_t ready = arrive > prev_done ? arrive : prev_done;
        det += seen > ready ? (double)(seen - ready) : 0.0;
        if (arrive > prev_done) { ++late; lateness += (double)(arrive - prev_done); }
        uint32_t b = 0; for (uint64_t v = hp / 128; v && b < 7; v >>= 1) ++b;
        ++hist[b];
        ++n;
    }
    if (n) {
        fprintf(stderr, "[density_hip prof]   D chain over %u rounds: hop %.0f (max %llu) cycles = critical section %.0f + detect %.0f; owner arrived late in %u rounds (avg lateness %.0f)\n",
                n, hop / n, (unsigned long long)hop_max, crit / n, det / n, late, late ? lateness / late : 0.0);
        fprintf(stderr, "[density_hip prof]   hop histogram (<128, <256, <512, <1k, <2k, <4k, <8k, more):");
        for (int b = 0; b < 8; ++b) fprintf(stderr, " %u", hist[b]);
        fprintf(stderr, "\n");
        // why owners are late: a wave's own iteration = arrival(r) - exchanges done(r - waves), split by what its previous round was
        const uint64_t* notes = h + 128 + 4 * kProfRounds;
        double it_plain = 0, it_zero = 0, tail_plain = 0, tail_zero = 0;
        uint32_t n_plain = 0, n_zero = 0, late_after_zero = 0, late_total = 0, zero_rounds = 0;
        for (uint32_t r = 17 + waves; r < kProfRounds; ++r) {
            if (!ts[4 * r] || !ts[4 * (r - waves) + 2] || !ts[4 * (r - waves) + 3] || !ts[4 * (r - 1) + 2]) continue;
            const bool z = (notes[r - waves] & 1u) != 0;
            const double it = (double)(ts[4 * r] - ts[4 * (r - waves) + 2]), tail = (double)(ts[4 * (r - waves) + 3] - ts[4 * (r - waves) + 2]);
            if (z) { it_zero += it; tail_zero += tail; ++n_zero; } else { it_plain += it; tail_plain += tail; ++n_plain; }
            if (ts[4 * r] > ts[4 * (r - 1) + 2]) { ++late_total; if (z) ++late_after_zero; }
            if (notes[r] & 1u) ++zero_rounds;
        }
        fprintf(stderr, "[density_hip prof]   a wave between its exchanges and its next arrival: %.0f cycles (%.0f of them up to the end of the round) after a plain round (%u), "
                        "%.0f (%.0f) after a zero-entry round (%u); %u of %u late arrivals follow a zero-entry round; %u zero-entry rounds\n",
                n_plain ? it_plain / n_plain : 0.0, n_plain ? tail_plain / n_plain : 0.0, n_plain, n_zero ? it_zero / n_zero : 0.0, n_zero ? tail_zero / n_zero : 0.0, n_zero,
                late_after_zero, late_total, zero_rounds);
    }
}
// how long the decoder's waiting waves sleep (units of 64 cycles): per hand-off still to come (bits 8..11 of the kernel's flags) and once the token has
// reached the predecessor (bits 16..19); DENSITY_HIP_NAP="far,near" overrides (tuning runs)
uint32_t decode_naps(uint32_t round_len) {
    static const char* env = debug_env("DENSITY_HIP_NAP");
    uint32_t far_ = round_len >= 16 ? 5u : 5u, near_ = round_len >= 16 ? 3u : 3u;
    if (env) { unsigned a = 0, b = 0; if (sscanf(env, "%u,%u", &a, &b) == 2) { far_ = a & 15u; near_ = b & 15u; } }
    return (far_ << 8) | (near_ << 16);
}
uint32_t rot_tune() {
    static const uint32_t t = debug_env("DENSITY_HIP_TUNE") ? (uint32_t)atoi(debug_env("DENSITY_HIP_TUNE")) : 0u;   // read once: bit 0 = token after answers
    return t;
}
}  // namespace

bool g_rotor_split = kRotorSplitDefault;
constexpr int kSplitRound = 16;                // blocks per round of the split encoder: 12 fit the 128 registers sixteen waves have (16 spill: DESIGN.md 4.3)
bool rotor_encode_eligible(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks) {
    const bool aligned = ((uintptr_t)d_in % 4 == 0) && (n_chunks == 1 || chunk_bytes % 4 == 0);
    return aligned && (n_chunks == 1 ? total : chunk_bytes) < (1ull << 31);   // 32-bit stream positions
}
hipError_t launch_rotor_encode(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                               uint64_t* d_sizes, uint8_t* d_index, uint32_t* d_err, hipStream_t stream) {
    uint64_t* prof = rot_prof_buffer();
    // geometry (DENSITY_HIP_TUNE bits 2..4): 0 = default = rounds of 16 blocks on 8 waves (the longer round amortises the hand-off, and 8
    // waves have the registers to keep their quads), 1 = 8 blocks on 16 waves, 2 = 16 blocks on 12 waves (as fast as the default, more code)
    // (12 or 16 blocks on 12 waves with kept quads do not fit: the compiler needs the staging registers / spills 43 registers)
    // The default asks for the next round's quads right behind its exchanges (EARLY: 2 % faster than behind the commit); bit 8: behind the commit
    const uint32_t sel = (rot_tune() >> 2) & 7u;
    const bool early = !((rot_tune() >> 8) & 1u);
    if (g_rotor_split && sel == 0) {
        // the split encoder (round 5): 8 chain + 8 emit waves, the quads handed over through an LDS ring (kernel variant bit 11 selects the other one)
        auto ks = prof ? chameleon_encode_rot<kSplitRound, 8, true, false, false, false, true> : chameleon_encode_rot<kSplitRound, 8, false, false, false, false, true>;
        hipError_t es = hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEncLdsSplit);
        if (es != hipSuccess) return es;
        hipLaunchKernelGGL(ks, dim3(n_chunks), dim3(1024), kEncLdsSplit, stream, d_in, total, chunk_bytes, d_out, out_stride, d_sizes, d_index, d_err, SegArgs{}, prof);
        rot_prof_report("encode (split)", "chain waves 0-7: hash | D wait | exchange | signatures | O wait+commit | ring: wait for the quads | post | ring: the reads (+ in-order rounds);  emit waves 8-15: - | ring transfer incl. the wait for the slot | - | - | mail box wait | - | emit | -", prof, stream, 16);
        return hipGetLastError();
    }
    const uint32_t waves = sel == 1 ? 16 : sel == 2 ? 12 : 8;
    auto kernel = sel == 1 ? (prof ? chameleon_encode_rot<8, 16, true> : chameleon_encode_rot<8, 16, false>)
                : sel == 2 ? (prof ? chameleon_encode_rot<16, 12, true> : chameleon_encode_rot<16, 12, false>)
                : early    ? (prof ? chameleon_encode_rot<16, 8, true, true, true> : chameleon_encode_rot<16, 8, false, true, true>)
                           : (prof ? chameleon_encode_rot<16, 8, true> : chameleon_encode_rot<16, 8, false>);
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEncLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(waves * 64), kEncLds, stream, d_in, total, chunk_bytes, d_out, out_stride, d_sizes, d_index, d_err, SegArgs{}, prof);
    rot_prof_report("encode", "hash | D wait | exchange | signatures | O wait+commit | load wait | emit | in-order rounds", prof, stream, waves);
    return hipGetLastError();
}
hipError_t launch_rotor_encode_paged(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_pages, uint32_t page_limit,
                                     uint32_t* d_page_counter, uint32_t* d_dir, uint32_t dir_words, uint64_t* d_sizes, uint8_t* d_index, uint32_t* d_err, hipStream_t stream) {
    auto kernel = g_rotor_split ? chameleon_encode_rot<kSplitRound, 8, false, false, false, true, true> : chameleon_encode_rot<16, 8, false, true, true, true>;
    const uint32_t lds = g_rotor_split ? kEncLdsSplit : kEncLds;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    SegArgs pg;
    pg.page_counter = d_page_counter; pg.page_dir = d_dir; pg.page_dir_words = dir_words; pg.page_limit = page_limit;
    hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(g_rotor_split ? 1024 : 512), lds, stream, d_in, total, chunk_bytes, d_pages, (uint64_t)0, d_sizes, d_index, d_err, pg, (uint64_t*)nullptr);
    return hipGetLastError();
}
bool rotor_decode_eligible(const uint8_t* d_out, uint32_t n_chunks, uint64_t out_stride, uint64_t out_total, const uint8_t* d_index, const uint32_t* d_zmap) {
    if (!d_index || !d_zmap || n_chunks > kMaxPipelinedChunks) return false;
    const uint64_t per_chunk = n_chunks == 1 ? (out_total < out_stride ? out_total : out_stride) : out_stride;
    if ((per_chunk + kBlock - 1) / kBlock > kRotMaxBlocks) return false;
    if ((uintptr_t)d_index % 4 != 0 || (n_chunks > 1 && (out_stride / kBlock) % 4 != 0)) return false;
    return (uintptr_t)d_out % 4 == 0 && (n_chunks == 1 || out_stride % 4 == 0);
}
hipError_t launch_rotor_decode(const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out,
                               uint64_t out_stride, uint64_t out_total, bool exact, const uint8_t* d_index, uint32_t* d_zmap,
                               uint64_t* d_produced, uint32_t* d_err, hipStream_t stream) {
    uint64_t* prof = rot_prof_buffer();
    // geometry (DENSITY_HIP_TUNE bits 5..7): 0 = default = rounds of 12 records on 12 waves (168 registers each: the longest round that does
    // not spill, i.e. the shortest chain per record), 1 = 8 records on 16 waves.  (Rounds of 16 on 12 waves and of 12 on 16 were built and
    // measured in round 3: both spill — 310 / 200 register slots — and are gone.)
    // Round 4, with the rare paths rolled (120 registers instead of 160), longer rounds and more waves build without spills: 2 = 16 records on
    // 12 waves, 4 = 12 on 16.  Measured on one box against the default's 0.381 / 0.421 ms (fast / slow box): 16 on 12 0.437 (a round's critical
    // section grows with its length — 38 cycles per record either way — so only the hand-off's ~210 cycles are spread thinner, 4 cycles per
    // record, and a wave whose 16 records take longer than 12 hand-offs is late more often than that pays); 12 on 16 0.420 (no gain: the
    // decoder waits for its chain, not for issue slots).  Also built and measured: 20 on 12 (0.426), one set of item registers with the next
    // round's loads behind the quads — 16 on 16 0.422, 12 on 12 0.398, 16 on 12 0.426 (the loads' run-up is too short: stalls of 2-8 k cycles).
    const uint32_t sel = (rot_tune() >> 5) & 7u;
    const uint32_t waves = (sel == 1 || sel == 4) ? 16 : 12;
    auto kernel = sel == 1 ? (prof ? chameleon_decode_rot<8, 16, true> : chameleon_decode_rot<8, 16, false>)
                : sel == 2 ? (prof ? chameleon_decode_rot<16, 12, true> : chameleon_decode_rot<16, 12, false>)
                : sel == 4 ? (prof ? chameleon_decode_rot<12, 16, true> : chameleon_decode_rot<12, 16, false>)
                           : (prof ? chameleon_decode_rot<12, 12, true> : chameleon_decode_rot<12, 12, false>);
    const uint32_t rlen = sel == 1 ? 8 : sel == 2 ? 16 : 12;
    const uint32_t lds = dec_lds_bytes(rlen);
    const uint32_t naps = decode_naps(rlen);
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(waves * 64), lds, stream, d_in, d_offsets, d_sizes, d_out, out_stride, out_total,
                       (exact ? 1u : 0u) | naps, d_index, d_zmap, d_produced, d_err, SegArgs{}, prof);
    rot_prof_report("decode", "stage A | stage B | operands | D wait | exchange | quads | Z chain | stores", prof, stream, waves);
    return hipGetLastError();
}
hipError_t launch_rotor_decode_paged(const uint8_t* d_pages, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                                     uint64_t out_total, const uint8_t* d_index, const uint32_t* d_dir, uint32_t dir_words, uint32_t n_pages, uint32_t* d_zmap,
                                     uint64_t* d_produced, uint32_t* d_err, hipStream_t stream) {
    auto kernel = chameleon_decode_rot<12, 12, false, true>;
    const uint32_t lds = dec_lds_bytes_paged(12);
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    SegArgs pg;
    pg.page_dir = const_cast<uint32_t*>(d_dir); pg.page_dir_words = dir_words; pg.page_limit = n_pages;
    hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(768), lds, stream, d_pages, d_offsets, d_sizes, d_out, out_stride, out_total, 1u | decode_naps(12), d_index, d_zmap,
                       d_produced, d_err, pg, (uint64_t*)nullptr);
    return hipGetLastError();
}
hipError_t launch_rotor_selftest(uint32_t* d_fail, hipStream_t stream) {
    hipLaunchKernelGGL(rotor_selftest_kernel, dim3(1), dim3(kRotThreads), 0, stream, d_fail, rot_tune());
    return hipGetLastError();
}


hipError_t launch_rotor_encode_seg(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                                   uint64_t* d_sizes, uint32_t* d_err, SegArgs seg, hipStream_t stream) {
    auto kernel = g_rotor_split ? chameleon_encode_rot<kSplitRound, 8, false, false, false, false, true> : chameleon_encode_rot<16, 8, false, true, true>;
    const uint32_t lds = g_rotor_split ? kEncLdsSplit : kEncLds;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(g_rotor_split ? 1024 : 512), lds, stream, d_in, total, chunk_bytes, d_out, out_stride, d_sizes, (uint8_t*)nullptr, d_err, seg, (uint64_t*)nullptr);
    return hipGetLastError();
}
hipError_t launch_rotor_lastwriters(const uint8_t* d_in, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_images, uint32_t* d_err, hipStream_t stream) {
    if (n_chunks == 0) return hipSuccess;
    hipError_t e = hipFuncSetAttribute((const void*)chameleon_lastwriters_rot, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kEncLds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(chameleon_lastwriters_rot, dim3(n_chunks), dim3(1024), kEncLds, stream, d_in, chunk_bytes, d_images, d_err);
    return hipGetLastError();
}
hipError_t launch_merge_images(const uint8_t* d_base, const uint8_t* d_lastwriters, uint8_t* d_start, uint32_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    // the marks are OR-ed in: clear them first (one strided fill)
    hipError_t e = hipMemset2DAsync(d_start + kTableBytes, kSegImageBytes, 0, kZmapBytes, count, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(merge_images_kernel, dim3(65536 / 256), dim3(256), 0, stream, d_base, d_lastwriters, d_start, count);
    return hipGetLastError();
}
hipError_t launch_scan_offsets(const uint64_t* d_sizes, uint32_t first, uint32_t count, uint64_t* d_carry, uint64_t* d_offsets, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(scan_offsets_kernel, dim3(1), dim3(64), 0, stream, d_sizes, first, count, d_carry, d_offsets);
    return hipGetLastError();
}
hipError_t launch_compact_bytes(const uint8_t* d_src, uint64_t src_stride, const uint64_t* d_sizes, const uint64_t* d_offsets, uint32_t n_chunks,
                                uint8_t* d_dst, hipStream_t stream) {
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(compact_bytes_kernel, dim3(16, n_chunks), dim3(256), 0, stream, d_src, src_stride, d_sizes, d_offsets, d_dst);
    return hipGetLastError();
}


hipError_t launch_rotor_decode_seg(const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out,
                                   uint64_t out_stride, uint64_t out_total, const uint8_t* d_index, uint32_t* d_zmap, uint64_t* d_produced, uint32_t* d_err,
                                   SegArgs seg, hipStream_t stream) {
    auto kernel = chameleon_decode_rot<12, 12, false>;
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dec_lds_bytes(12));
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(768), dec_lds_bytes(12), stream, d_in, d_offsets, d_sizes, d_out, out_stride, out_total, decode_naps(12), d_index, d_zmap, d_produced, d_err,
                       seg, (uint64_t*)nullptr);
    return hipGetLastError();
}

}  // namespace density
