// serial_codec.hip — Cheetah and Lion on the device: tables in global memory.
//
// Dictionary and predictor tables live in a global-memory workspace (Cheetah 768 KiB, Lion 1.75 MiB per stream:
// cheetah.rs:25-55, lion.rs:29-72 — neither fits LDS, and the predictor holds arbitrary quads, so the 16-bit packing of
// chameleon.hip does not apply to it).  Two kernel families, bit-exact with each other and with the oracle:
//   * one LANE per chunk stream (`serial_*_chunks`): the reference's scalar algorithm per lane.  Round 1's functional path; now
//     the cross-check (density_hip_set_kernel_variant(16)) and, as plain device functions, the ragged ends and the re-decode of
//     mis-speculated records of the kernels below;
//   * one WAVE per chunk stream (`cheetah_*_wave`, `lion_*_wave`, second half of this file): a record per step, a quad per lane —
//     the default.
// The code is written against the format, with the reference lines it must agree with cited per function.
#include "common.hpp"
#include "kernels.hpp"

namespace density {

bool g_force_lane_codec = false;   // density_hip_set_kernel_variant(16): Cheetah on the one-lane-per-stream kernels (cross-check)
bool g_lion_one_wave = false;      // density_hip_set_kernel_variant(32768): Lion's decode on ONE wave per stream (round 4's) instead of two

namespace {

struct Pair { uint32_t a, b; };
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// (Lion's prediction rows at a stride of 32 bytes instead of 20 — a row then never crosses a 32-byte sector — measured slower: decode 3.01 against 2.83 ms, encode 2.15
// against 2.06: the tables grow by 43 %, and so do their clearing and their footprint in the caches)
template <int ALGO> struct Geo;
template <> struct Geo<DENSITY_HIP_CHEETAH> {                 // cheetah.rs:17-23,188-196
    static constexpr uint32_t kFlagBits = 2, kSig = 8, kBlock = 128, kPredWords = 1;
};
template <> struct Geo<DENSITY_HIP_LION> {                    // lion.rs:17-27,317-325
    static constexpr uint32_t kFlagBits = 3, kSig = 6, kBlock = 64, kPredWords = 5;
};

template <int ALGO>
struct Tables {
    Pair* dict;          // 64 Ki x {a, b}
    uint32_t* pred;      // 64 Ki x kPredWords
    uint32_t last_hash;
    __device__ void clear() {
        uint4* p = reinterpret_cast<uint4*>(dict);
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint32_t i = 0; i < 65536u * sizeof(Pair) / 16; ++i) p[i] = z;
        uint4* r = reinterpret_cast<uint4*>(pred);
        for (uint32_t i = 0; i < 65536u * Geo<ALGO>::kPredWords * 4u / 16; ++i) r[i] = z;
        last_hash = 0;
    }
};

__device__ __forceinline__ uint32_t hash16(uint32_t q) { return (q * kHashMul) >> 16; }

// ---- per-quad steps.  `item` receives the bytes to emit (0, 2 or 4), the return value is the flag. ----

// cheetah.rs:123-149
__device__ __forceinline__ uint32_t enc_quad(Tables<DENSITY_HIP_CHEETAH>& t, uint32_t q, uint32_t& item, uint32_t& item_len) {
    const uint32_t h = hash16(q);
    uint32_t* guess = t.pred + t.last_hash;
    Pair* e = t.dict + h;
    // both table reads are issued before either is needed (the dictionary entry is unused for a predicted quad): one memory
    // round trip per quad instead of two
    const uint32_t predicted = *guess;
    const Pair cur = *e;
    uint32_t flag;
    if (predicted == q) { flag = 3; item_len = 0; }
    else {
        if (cur.a == q) { flag = 1; item = h; item_len = 2; }
        else {
            if (cur.b == q) { flag = 2; item = h; item_len = 2; } else { flag = 0; item = q; item_len = 4; }
            *e = Pair{q, cur.a};
        }
        *guess = q;
    }
    t.last_hash = h;
    return flag;
}

// lion.rs:50-57,211-270
__device__ __forceinline__ uint32_t enc_quad(Tables<DENSITY_HIP_LION>& t, uint32_t q, uint32_t& item, uint32_t& item_len) {
    const uint32_t h = hash16(q);
    uint32_t* p = t.pred + 5u * t.last_hash;
    Pair* e = t.dict + h;
    uint32_t n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3], n4 = p[4];
    const Pair cur = *e;                                      // issued with the predictions: one memory round trip per quad
    uint32_t flag;
    item_len = 0;
    if (n0 == q) { flag = 1; }
    else if (n1 == q) { flag = 2; p[1] = n0; p[0] = q; }
    else if (n2 == q) { flag = 3; p[2] = n1; p[1] = n0; p[0] = q; }
    else if (n3 == q) { flag = 4; p[3] = n2; p[2] = n1; p[1] = n0; p[0] = q; }
    else {
        if (n4 == q) { flag = 5; }
        else {
            if (cur.a == q) { flag = 6; item = h; item_len = 2; }
            else {
                if (cur.b == q) { flag = 7; item = h; item_len = 2; } else { flag = 0; item = q; item_len = 4; }
                *e = Pair{q, cur.a};
            }
        }
        p[4] = n3; p[3] = n2; p[2] = n1; p[1] = n0; p[0] = q;     // shift_predictions
    }
    t.last_hash = h;
    return flag;
}

__device__ __forceinline__ uint32_t item_bytes(Tables<DENSITY_HIP_CHEETAH>&, uint32_t flag) { return flag == 0 ? 4u : (flag == 3 ? 0u : 2u); }
__device__ __forceinline__ uint32_t item_bytes(Tables<DENSITY_HIP_LION>&, uint32_t flag) { return flag == 0 ? 4u : (flag >= 6 ? 2u : 0u); }

// cheetah.rs:68-103,154-163 — `p` points at the item (if any)
__device__ __forceinline__ uint32_t dec_quad(Tables<DENSITY_HIP_CHEETAH>& t, uint32_t flag, const uint8_t* p) {
    uint32_t q, h;
    if (flag == 3) { q = t.pred[t.last_hash]; h = hash16(q); }
    else {
        if (flag == 0) { q = ld32u(p); h = hash16(q); Pair* e = t.dict + h; *e = Pair{q, e->a}; }
        else {
            h = ld16u(p);
            Pair* e = t.dict + h;
            const Pair cur = *e;
            if (flag == 1) q = cur.a; else { q = cur.b; *e = Pair{q, cur.a}; }
        }
        t.pred[t.last_hash] = q;
    }
    t.last_hash = h;
    return q;
}

// lion.rs:85-186,275-290
__device__ __forceinline__ uint32_t dec_quad(Tables<DENSITY_HIP_LION>& t, uint32_t flag, const uint8_t* p5) {
    uint32_t q, h;
    uint32_t* p = t.pred + 5u * t.last_hash;
    if (flag >= 1 && flag <= 5) {
        q = p[flag - 1];
        for (uint32_t k = flag - 1; k > 0; --k) p[k] = p[k - 1];     // move to front (A: nothing moves)
        if (flag > 1) p[0] = q;
        h = hash16(q);
    } else {
        if (flag == 0) { q = ld32u(p5); h = hash16(q); Pair* e = t.dict + h; *e = Pair{q, e->a}; }
        else {
            h = ld16u(p5);
            Pair* e = t.dict + h;
            const Pair cur = *e;
            if (flag == 6) q = cur.a; else { q = cur.b; *e = Pair{q, cur.a}; }
        }
        p[4] = p[3]; p[3] = p[2]; p[2] = p[1]; p[1] = p[0]; p[0] = q;
    }
    t.last_hash = h;
    return q;
}

template <int ALGO>
__device__ __forceinline__ void store_sig(uint8_t* rec, uint64_t sig) {           // codec.rs:24-26, lion.rs:334-337
    for (uint32_t i = 0; i < Geo<ALGO>::kSig; ++i) rec[i] = (uint8_t)(sig >> (8 * i));
}

// codec.rs:28-31; lion.rs:340-351 (6 significant bytes)
template <int ALGO>
__device__ __forceinline__ uint64_t load_sig(const uint8_t* p) {
    uint64_t v = 0;
    for (uint32_t i = 0; i < Geo<ALGO>::kSig; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

// ---- Codec::encode / encode_block, codec.rs:34-80 ----
template <int ALGO>
__global__ __launch_bounds__(64) void serial_encode_chunks(const uint8_t* __restrict__ in, uint64_t total, uint64_t chunk_bytes,
                                                           uint32_t n_chunks, uint8_t* __restrict__ out, uint64_t out_stride,
                                                           uint64_t* __restrict__ sizes, uint8_t* __restrict__ tables, uint32_t n_slots) {
    // `tables` arrives zeroed (the launcher clears the slots in use with one memset), so a lane clears its own tables
    // only before its second and later chunks
    using G = Geo<ALGO>;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    constexpr uint64_t kTableBytes = 65536ull * (sizeof(Pair) + 4ull * G::kPredWords);
    Tables<ALGO> t;
    t.dict = reinterpret_cast<Pair*>(tables + slot * kTableBytes);
    t.pred = reinterpret_cast<uint32_t*>(tables + slot * kTableBytes + 65536ull * sizeof(Pair));
    for (uint64_t chunk = slot; chunk < n_chunks; chunk += n_slots) {
        const uint8_t* src = in + chunk * chunk_bytes;
        const uint64_t len = (total - chunk * chunk_bytes) < chunk_bytes ? (total - chunk * chunk_bytes) : chunk_bytes;
        uint8_t* dst = out + chunk * out_stride;
        if (chunk != slot) t.clear();
        t.last_hash = 0;
        Guard guard;
        uint64_t opos = 0;
        for (uint64_t pos = 0; pos < len; pos += G::kBlock) {
            const uint32_t blen = (len - pos) < G::kBlock ? (uint32_t)(len - pos) : G::kBlock;
            const uint8_t* blk = src + pos;
            if (guard.block_is_copy()) {                                      // codec.rs:35-37
                for (uint32_t i = 0; i < blen; ++i) dst[opos + i] = blk[i];
                opos += blen;
                guard.decay();
                continue;
            }
            uint8_t* rec = dst + opos;
            uint64_t o = G::kSig, sig = 0;
            const uint32_t nq = blen >> 2;
            for (uint32_t k = 0; k < nq; ++k) {                               // codec.rs:42-57
                uint32_t item = 0, ilen = 0;
                const uint32_t flag = enc_quad(t, ld32u(blk + 4u * k), item, ilen);
                sig |= (uint64_t)flag << (G::kFlagBits * k);                  // io/write_signature.rs:14-17
                if (ilen == 2) st16u(rec + o, item); else if (ilen == 4) st32u(rec + o, item);
                o += ilen;
            }
            for (uint32_t i = 4u * nq; i < blen; ++i) rec[o++] = blk[i];     // codec.rs:58-61
            store_sig<ALGO>(rec, sig);
            guard.update(o >= G::kBlock);                                     // codec.rs:68
            opos += o;
        }
        sizes[chunk] = opos;
    }
}

// ---- Codec::decode, codec.rs:82-126 (fast and tail loops unified: per-unit checks are exact for both) ----
template <int ALGO>
__global__ __launch_bounds__(64) void serial_decode_chunks(const uint8_t* __restrict__ in, const uint64_t* __restrict__ offsets,
                                                           const uint64_t* __restrict__ sizes, uint32_t n_chunks,
                                                           uint8_t* __restrict__ out, uint64_t out_stride, uint64_t out_total,
                                                           uint32_t exact, uint64_t* __restrict__ produced, uint32_t* __restrict__ err,
                                                           uint8_t* __restrict__ tables, uint32_t n_slots) {
    using G = Geo<ALGO>;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    constexpr uint64_t kTableBytes = 65536ull * (sizeof(Pair) + 4ull * G::kPredWords);
    constexpr uint64_t kMask = (1ull << G::kFlagBits) - 1;
    Tables<ALGO> t;
    t.dict = reinterpret_cast<Pair*>(tables + slot * kTableBytes);
    t.pred = reinterpret_cast<uint32_t*>(tables + slot * kTableBytes + 65536ull * sizeof(Pair));
    for (uint64_t chunk = slot; chunk < n_chunks; chunk += n_slots) {
        const uint8_t* src = in + offsets[chunk];
        const uint64_t elen = sizes[chunk];
        uint8_t* dst = out + chunk * out_stride;
        const uint64_t room_all = out_total - chunk * out_stride;
        const uint64_t cap = room_all < out_stride ? room_all : out_stride;
        if (chunk != slot) t.clear();
        t.last_hash = 0;
        Guard guard;
        uint64_t ipos = 0, opos = 0;
        bool bad = false, done = false;
        while (ipos < elen && !bad && !done) {
            const uint64_t rem = elen - ipos;
            if (guard.block_is_copy()) {                                      // codec.rs:89-91,103-110
                const uint32_t take = rem > G::kBlock ? G::kBlock : (uint32_t)rem;
                if (opos + take > cap) { bad = true; break; }
                for (uint32_t i = 0; i < take; ++i) dst[opos + i] = src[ipos + i];
                ipos += take; opos += take;
                if (rem <= G::kBlock) break;
                guard.decay();
                continue;
            }
            if (rem < G::kSig) { bad = true; break; }                         // read_signature would panic
            const uint64_t mark = ipos;
            uint64_t sig = load_sig<ALGO>(src + ipos);
            ipos += G::kSig;
            for (uint32_t k = 0; k < G::kBlock / 4; ++k) {
                const uint32_t flag = (uint32_t)(sig & kMask);
                sig >>= G::kFlagBits;
                const uint64_t left = elen - ipos;
                if (flag == 0 && left < 4) {                                  // cheetah.rs:168-176, lion.rs:295-303: end of data
                    if (opos + left > cap) { bad = true; break; }
                    for (uint32_t i = 0; i < left; ++i) dst[opos + i] = src[ipos + i];
                    opos += left; ipos += left; done = true;
                    break;
                }
                const uint32_t need = item_bytes(t, flag);
                if (left < need || opos + 4 > cap) { bad = true; break; }
                const uint32_t q = dec_quad(t, flag, src + ipos);
                ipos += need;
                st32u(dst + opos, q);
                opos += 4;
            }
            if (!done && !bad) guard.update(ipos - mark >= G::kBlock);        // codec.rs:98,122
        }
        if (exact && !bad && opos != cap) bad = true;
        produced[chunk] = opos;
        if (bad) atomicOr(err, 1u);
    }
}


// =================================================================================================================
// Cheetah, one WAVE per chunk stream: a record (32 quads, cheetah.rs:17-23) per step, one quad per lane.
//
// What is sequential in the reference is the state of the two tables.  On encode every table address is known from the
// input alone (dictionary slot = hash of the quad, predictor slot = hash of the PREVIOUS quad, cheetah.rs:123-149), so a step
//   1. gathers the predictor word and the dictionary pair of all 32 quads at once,
//   2. finds, per lane, the nearest earlier lane of the record on the same predictor slot and on the same dictionary slot
//      (exact 16-bit key match from 16 ballots per key),
//   3. resolves the record in dependency order: a lane is ready when the lanes it follows are done, and takes the state of
//      its slots from their registers (ds_bpermute) instead of from memory — usually one round, a chain of equal quads takes
//      one round per link,
//   4. stores flags / items through a prefix sum of the item lengths, and writes each touched slot back once (last lane on it).
// The tables stay in global memory (768 KiB per stream): a step costs one gather and one scatter round trip instead of 32
// dependent ones.  The blow-up protection FSM (codec.rs:35-37,68) runs per record on uniform values; a raw-copy block touches
// nothing.  The ragged last block goes through the scalar code above on lane 0.
//
// On decode the predictor slot of a predicted quad (flag 3) is the hash of a quad that is itself looked up, so runs of
// predicted quads are a chain of dependent reads: each round of step 1 advances every run of the record by one quad,
// speculating that no earlier quad of the same record wrote the slot it reads; step 3 then knows the truth, and a record with a
// mis-speculated prediction (a repeated pair of quads inside 128 bytes) is decoded again by the scalar code.
// =================================================================================================================
// Table accesses: ONE wave ever touches a stream's tables, so nothing wider than its own CU has to agree on them — work-group scope:
// the accesses go through this XCD's L2 instead of past it (agent scope on a chip of eight XCDs means "coherent across their
// L2s": every gather paid the trip to memory, ~1900 cycles a step).
typedef unsigned long long u64a;
__device__ __forceinline__ uint32_t tbl_load32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void tbl_store32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ Pair tbl_load_pair(const Pair* p) {
    const u64a v = __hip_atomic_load(reinterpret_cast<const u64a*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return Pair{(uint32_t)v, (uint32_t)(v >> 32)};
}
__device__ __forceinline__ void tbl_store_pair(Pair* p, Pair v) {
    __hip_atomic_store(reinterpret_cast<u64a*>(p), (u64a)v.a | ((u64a)v.b << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the table stores of a step are complete (in L2, where the next step's gathers read) before anything else happens
__device__ __forceinline__ void tbl_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// lanes 0..31: the lanes whose 16-bit key equals this lane's (bit-plane ballots: exact) — for two keys at once: lanes 0..31 match `k0`
// among themselves while lanes 32..63 match copies of `k1`, one set of ballots
__device__ __forceinline__ void same_key_masks2(uint32_t k0, bool on0, uint32_t k1, bool on1, uint32_t lane, uint32_t& eq0, uint32_t& eq1) {
    const uint32_t k1up = bperm(lane & 31u, k1);                                  // lane 32+i: lane i's second key
    const uint32_t on_up = bperm(lane & 31u, on1 ? 1u : 0u);
    const bool upper = lane >= 32;
    const uint32_t key = upper ? k1up : k0;
    const bool on = upper ? on_up != 0 : on0;
    uint32_t eq = 0xffffffffu;
#pragma unroll
    for (uint32_t b = 0; b < 16; ++b) {
        const bool bit = (key >> b) & 1u;
        const uint64_t plane = ballot64(bit && on);
        const uint32_t half = upper ? (uint32_t)(plane >> 32) : (uint32_t)plane;
        eq &= bit ? half : ~half;
    }
    const uint64_t ons = ballot64(on);
    eq &= upper ? (uint32_t)(ons >> 32) : (uint32_t)ons;
    eq0 = eq;                                                                     // (lanes 0..31)
    eq1 = bperm(lane | 32u, eq);                                                  // lanes 0..31 fetch their second mask from above
}
// the lanes whose 16-bit key equals this lane's, over all 64 lanes (bit-plane ballots: exact)
__device__ __forceinline__ uint64_t same_key_mask64(uint32_t key, bool on) {
    uint64_t eq = ~0ull;
#pragma unroll
    for (uint32_t b = 0; b < 16; ++b) {
        const bool bit = (key >> b) & 1u;
        const uint64_t plane = ballot64(bit && on);
        eq &= bit ? plane : ~plane;
    }
    return eq & ballot64(on);
}
// 2-bit flags of lanes 0..31 -> the 64-bit signature (io/write_signature.rs:14-17: quad k at bits 2k, 2k+1)
__device__ __forceinline__ uint64_t spread32(uint32_t x) {
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
// exclusive prefix sum over lanes 0..31 (values of the other lanes ignored), and the total: DPP row shifts within the rows of 16,
// then lane 15's sum broadcast into row 1 (a shuffle through LDS per step would cost ten times the latency)
__device__ __forceinline__ uint32_t scan32(uint32_t v, uint32_t lane, uint32_t& total) {
    const uint32_t mine = lane < 32 ? v : 0u;
    uint32_t incl = mine;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
    total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31);
    return incl - mine;
}

// One record's worth of sequential table semantics, resolved across lanes.  In: per lane its predictor slot `ps`, dictionary
// slot `ds` and the memory state of both; `op` says what the lane does once it knows its state.  Out: post-state per lane.
struct CheetahLane {
    uint32_t pv, da, db;        // state of my predictor word / dictionary pair when my turn comes (then: after my turn)
    uint32_t pdirty, ddirty;    // ... differs from memory
};

__global__ __launch_bounds__(64) void cheetah_encode_wave(const uint8_t* __restrict__ in, uint64_t total, uint64_t chunk_bytes,
                                                          uint32_t n_chunks, uint8_t* __restrict__ out, uint64_t out_stride,
                                                          uint64_t* __restrict__ sizes, uint8_t* __restrict__ tables, uint32_t n_slots,
                                                          const uint32_t* __restrict__ only, uint32_t* __restrict__ head_state, uint32_t head_bytes,
                                                          uint32_t head_calm, const uint32_t* __restrict__ tail_state) {
    // `only` (nullable): encode just the chunks it marks — the ones exchange_stages.hip hands back (raw copies, a ragged end).
    // `head_state` (nullable): encode just the head of every chunk — at least head_bytes, and on until the blow-up protection has been
    // quiet for head_calm bytes: a cold dictionary makes records incompressible and blocks get copied — and leave the tables (one slot per
    // chunk) and eight words per chunk behind: stream bytes so far, last_hash, FSM (bit 0 penalty running, bit 1 last record
    // incompressible), 0 = the passes take over / 2 = the chunk was finished here (short, or never calm), the FSM's penalty_start and
    // block counter, the input offset where the passes take over.
    // `tail_state` (nullable): encode just the ragged end of the chunks it marks, from where the passes stopped — eight words per chunk:
    // input offset, stream bytes so far, last_hash, last record incompressible, penalty_start, counter, 1 = there is an end to do; the
    // tables (one slot per chunk) are as the passes left them
    using G = Geo<DENSITY_HIP_CHEETAH>;
    const uint32_t slot = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (slot >= n_slots) return;
    constexpr uint64_t kTableBytes = 65536ull * (sizeof(Pair) + 4ull * G::kPredWords);
    Tables<DENSITY_HIP_CHEETAH> t;
    t.dict = reinterpret_cast<Pair*>(tables + slot * kTableBytes);
    t.pred = reinterpret_cast<uint32_t*>(tables + slot * kTableBytes + 65536ull * sizeof(Pair));
    for (uint64_t chunk = slot; chunk < n_chunks; chunk += n_slots) {
        const uint8_t* src = in + chunk * chunk_bytes;
        const uint64_t len = (total - chunk * chunk_bytes) < chunk_bytes ? (total - chunk * chunk_bytes) : chunk_bytes;
        uint8_t* dst = out + chunk * out_stride;
        if (only && !only[chunk]) continue;
        if (tail_state && !tail_state[8 * chunk + 6]) continue;
        if ((chunk != slot || only || head_state) && !tail_state) {                                          // (the launcher zeroed the tables for the first chunk of a slot — not when it filters)
            uint4* p = reinterpret_cast<uint4*>(tables + slot * kTableBytes);
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (uint64_t i = lane; i < kTableBytes / 16; i += 64) p[i] = z;
            __threadfence();
        }
        uint32_t last_hash = 0;
        Guard guard;
        uint64_t opos = 0, pos = 0;
        if (tail_state) {
            const uint32_t* ts = tail_state + 8 * chunk;
            pos = ts[0]; opos = ts[1]; last_hash = ts[2];
            guard.prev = ts[3]; guard.start = ts[4]; guard.counter = ts[5];
        }
        // lanes 0..31 hold the quads of the block at `p`, lanes 32..63 those of the block behind it (where that one is whole)
        auto load_at = [&](uint64_t p) -> uint32_t { return p + (uint64_t)G::kBlock * (1u + (lane >> 5)) <= len ? ld32u(src + p + 4u * lane) : 0u; };
        uint32_t qnext = load_at(pos);
        // head mode: hand over to the exchange passes at the first 4 KiB boundary behind head_bytes where the blow-up protection has been quiet
        // for head_calm bytes; a chunk that is too short for that, or has not calmed down within four times head_bytes (or by its middle), is simply finished here
        bool may_hand_over = head_state && len >= 4ull * head_bytes, handed_over = false;
        uint64_t last_copy_end = 0;
        while (pos + G::kBlock <= len) {                                      // whole blocks
            if (may_hand_over && pos >= head_bytes && (pos & 4095u) == 0) {
                if (pos >= last_copy_end + head_calm && guard.penalty == 0) { handed_over = true; break; }
                if (pos >= 4ull * head_bytes || pos >= len / 2) may_hand_over = false;   // raw copies this far in are not the cold start's: no hand-over
            }
            if (guard.block_is_copy()) {                                      // codec.rs:35-37
                last_copy_end = pos + G::kBlock;
                if (lane < 32) st32u(dst + opos + 4u * lane, qnext);
                qnext = load_at(pos + G::kBlock);
                opos += G::kBlock;
                pos += G::kBlock;
                guard.decay();
                continue;
            }
            // Round 6: TWO records per step where the second one is certain to be coded — a step costs its two memory round trips (gather, stores) whether
            // 32 or 64 lanes take part, and the heads of the exchange passes' chunks are a chain of such steps on one wave per CU.  The block behind this one
            // is coded whatever this one's size if the last record was not incompressible (then this one cannot start a penalty: protection_state.rs:38-47);
            // it stays out where the loop's hand-over test belongs in front of it (a 4 KiB boundary).
            const bool two = guard.prev == 0 && pos + 2ull * G::kBlock <= len && ((pos + G::kBlock) & 4095u) != 0;
            const uint32_t nact = two ? 64u : 32u;
            const bool act = lane < nact;
            // ---- 1. quads, slots, gather ----
            const uint32_t q = qnext;
            const uint32_t h = hash16(q);
            const uint32_t hprev = bperm(lane ? lane - 1u : 0u, h);
            const uint32_t ps = lane == 0 ? last_hash : hprev;               // predictor slot: hash of the previous quad (cheetah.rs:125,146)
            CheetahLane st;
            tbl_drain();                                                      // the previous step's table stores are in L2 (and its record stores, alas)
            st.pv = act ? tbl_load32(t.pred + ps) : 0u;
            const Pair e0 = act ? tbl_load_pair(t.dict + h) : Pair{0u, 0u};
            qnext = load_at(pos + (uint64_t)G::kBlock * (two ? 2u : 1u));     // the next step's quads: in flight across this step
            st.da = e0.a; st.db = e0.b; st.pdirty = 0; st.ddirty = 0;
            // ---- 2. who follows whom ----
            const uint64_t below = (1ull << lane) - 1ull;
            const uint64_t peq = same_key_mask64(ps, act), deq = same_key_mask64(h, act);
            const uint64_t pbefore = peq & below, dbefore = deq & below;
            const uint32_t pprev = pbefore ? 63u - (uint32_t)__builtin_clzll(pbefore) : 64u;   // 64: nobody
            const uint32_t dprev = dbefore ? 63u - (uint32_t)__builtin_clzll(dbefore) : 64u;
            const bool plast = act && ((peq >> lane) >> 1) == 0, dlast = act && ((deq >> lane) >> 1) == 0;
            // ---- 3. resolve in dependency order (cheetah.rs:123-149 per lane) ----
            uint32_t flag = 0;
            bool done = !act;
            for (uint32_t round = 0; round < 64; ++round) {                   // (a chain has at most 64 links)
                const uint64_t done_mask = ballot64(done && act);
                const bool pok = pprev == 64u || ((done_mask >> (pprev & 63u)) & 1ull), dok = dprev == 64u || ((done_mask >> (dprev & 63u)) & 1ull);
                const bool ready = !done && pok && dok;
                // the state my predecessors left (read from their registers; garbage where there is no predecessor)
                const uint32_t fpv = bperm(pprev & 63u, st.pv), fpd = bperm(pprev & 63u, st.pdirty);
                const uint32_t fda = bperm(dprev & 63u, st.da), fdb = bperm(dprev & 63u, st.db), fdd = bperm(dprev & 63u, st.ddirty);
                if (ready) {
                    if (pprev != 64u) { st.pv = fpv; st.pdirty = fpd; }
                    if (dprev != 64u) { st.da = fda; st.db = fdb; st.ddirty = fdd; }
                    if (st.pv == q) flag = 3;
                    else {
                        if (st.da == q) flag = 1;
                        else { flag = st.db == q ? 2u : 0u; st.db = st.da; st.da = q; st.ddirty = 1; }
                        st.pv = q; st.pdirty = 1;
                    }
                    done = true;
                }
                if (ballot64(!done) == 0) break;
            }
            // ---- 4. the records: signatures, items (io/write_buffer.rs), tables ----
            const uint32_t ilen = !act ? 0u : (flag == 0 ? 4u : (flag == 3 ? 0u : 2u));
            uint32_t incl = ilen;                                             // prefix sums within each half of the wave: rows of 16, then row 0's / row 2's total into rows 1 / 3
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xa, 0xf, true);   // row_bcast:15 into rows 1 and 3
            const uint32_t items_a = (uint32_t)__builtin_amdgcn_readlane((int)incl, 31), items_b = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            const uint32_t off = incl - ilen;
            const uint32_t rlen_a = G::kSig + items_a, rlen_b = G::kSig + items_b;
            uint8_t* rec = dst + opos + (lane >= 32 ? rlen_a : 0u);           // my record
            const uint64_t lo = ballot64(act && (flag & 1u)), hi = ballot64(act && (flag & 2u));
            const uint64_t sig_a = spread32((uint32_t)lo) | (spread32((uint32_t)hi) << 1), sig_b = spread32((uint32_t)(lo >> 32)) | (spread32((uint32_t)(hi >> 32)) << 1);
            if (lane < 2) st32u(dst + opos + 4u * lane, lane ? (uint32_t)(sig_a >> 32) : (uint32_t)sig_a);   // codec.rs:24-26
            if (two && (lane & ~1u) == 32u) st32u(dst + opos + rlen_a + 4u * (lane & 1u), (lane & 1u) ? (uint32_t)(sig_b >> 32) : (uint32_t)sig_b);
            if (ilen == 4) st32u(rec + G::kSig + off, q); else if (ilen == 2) st16u(rec + G::kSig + off, h);
            if (plast && st.pdirty) tbl_store32(t.pred + ps, st.pv);
            if (dlast && st.ddirty) tbl_store_pair(t.dict + h, Pair{st.da, st.db});
            last_hash = rfl(bperm(nact - 1u, h));
            guard.update(rlen_a >= G::kBlock);                                // codec.rs:68
            opos += rlen_a;
            pos += G::kBlock;
            if (two) {
                (void)guard.block_is_copy();                                  // (it is not — see `two` — but the call counts the block: protection_state.rs:19-27)
                guard.update(rlen_b >= G::kBlock);
                opos += rlen_b;
                pos += G::kBlock;
            }
        }
        tbl_drain();
        if (head_state) {
            if (lane == 0) {
                head_state[8 * chunk + 0] = (uint32_t)opos;
                head_state[8 * chunk + 1] = last_hash;
                head_state[8 * chunk + 2] = (guard.penalty ? 1u : 0u) | (guard.prev ? 2u : 0u);
                head_state[8 * chunk + 3] = handed_over ? 0u : 2u;             // 2: the chunk is finished, nothing for the passes
                head_state[8 * chunk + 4] = guard.start;
                head_state[8 * chunk + 5] = guard.counter;
                head_state[8 * chunk + 6] = (uint32_t)pos;                    // where the passes take over
            }
            if (handed_over) continue;
        }
        // ---- the ragged last block: the scalar code on lane 0 (same tables; codec.rs:51-63) ----
        if (pos < len) {
            __threadfence();                                                  // (the scalar code reads the tables with plain loads)
            if (lane == 0) {
                t.last_hash = last_hash;
                const uint32_t blen = (uint32_t)(len - pos);
                const uint8_t* blk = src + pos;
                if (guard.block_is_copy()) {
                    for (uint32_t i = 0; i < blen; ++i) dst[opos + i] = blk[i];
                    opos += blen;
                } else {
                    uint8_t* rec = dst + opos;
                    uint64_t o = G::kSig, sig = 0;
                    const uint32_t nq = blen >> 2;
                    for (uint32_t k = 0; k < nq; ++k) {
                        uint32_t item = 0, il = 0;
                        const uint32_t flag = enc_quad(t, ld32u(blk + 4u * k), item, il);
                        sig |= (uint64_t)flag << (G::kFlagBits * k);
                        if (il == 2) st16u(rec + o, item); else if (il == 4) st32u(rec + o, item);
                        o += il;
                    }
                    for (uint32_t i = 4u * nq; i < blen; ++i) rec[o++] = blk[i];
                    store_sig<DENSITY_HIP_CHEETAH>(rec, sig);
                    opos += o;
                }
            }
            opos = ((uint64_t)rfl((uint32_t)(opos >> 32)) << 32) | rfl((uint32_t)opos);
            __threadfence();
        }
        if (lane == 0) sizes[chunk] = opos;
    }
}


// one coded record by the scalar code (lane 0; codec.rs:92-99,111-123 with decode_partial_unit cheetah.rs:165-185): the decoder's
// ragged end and its answer to a mis-speculated record.  Returns "bad".
__device__ __forceinline__ bool cheetah_record_scalar(Tables<DENSITY_HIP_CHEETAH>& t, const uint8_t* src, uint64_t elen, uint64_t& ipos,
                                                      uint8_t* dst, uint64_t cap, uint64_t& opos, bool& done, Guard& guard) {
    using G = Geo<DENSITY_HIP_CHEETAH>;
    if (elen - ipos < G::kSig) return true;                                   // read_signature would panic
    const uint64_t mark = ipos;
    uint64_t sig = load_sig<DENSITY_HIP_CHEETAH>(src + ipos);
    ipos += G::kSig;
    for (uint32_t k = 0; k < G::kBlock / 4; ++k) {
        const uint32_t flag = (uint32_t)(sig & 3u);
        sig >>= 2;
        const uint64_t left = elen - ipos;
        if (flag == 0 && left < 4) {                                          // cheetah.rs:168-176: end of data
            if (opos + left > cap) return true;
            for (uint32_t i = 0; i < left; ++i) dst[opos + i] = src[ipos + i];
            opos += left; ipos += left; done = true;
            return false;
        }
        const uint32_t need = item_bytes(t, flag);
        if (left < need || opos + 4 > cap) return true;
        const uint32_t q = dec_quad(t, flag, src + ipos);
        ipos += need;
        st32u(dst + opos, q);
        opos += 4;
    }
    guard.update(ipos - mark >= G::kBlock);                                   // codec.rs:98,122
    return false;
}
__device__ __forceinline__ uint64_t bcast64(uint64_t v) { return ((uint64_t)rfl((uint32_t)(v >> 32)) << 32) | rfl((uint32_t)v); }

__global__ __launch_bounds__(64) void cheetah_decode_wave(const uint8_t* __restrict__ in, const uint64_t* __restrict__ offsets,
                                                          const uint64_t* __restrict__ sizes, uint32_t n_chunks,
                                                          uint8_t* __restrict__ out, uint64_t out_stride, uint64_t out_total,
                                                          uint32_t exact, uint64_t* __restrict__ produced, uint32_t* __restrict__ err,
                                                          uint8_t* __restrict__ tables, uint32_t n_slots) {
    using G = Geo<DENSITY_HIP_CHEETAH>;
    const uint32_t slot = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (slot >= n_slots) return;
    constexpr uint64_t kTableBytes = 65536ull * (sizeof(Pair) + 4ull * G::kPredWords);
    constexpr uint32_t kMaxRecord = G::kSig + G::kBlock;                      // 8 + 32 x 4
    Tables<DENSITY_HIP_CHEETAH> t;
    t.dict = reinterpret_cast<Pair*>(tables + slot * kTableBytes);
    t.pred = reinterpret_cast<uint32_t*>(tables + slot * kTableBytes + 65536ull * sizeof(Pair));
    const bool act = lane < 32;
    const uint32_t below = (1u << (lane & 31u)) - 1u;
    for (uint64_t chunk = slot; chunk < n_chunks; chunk += n_slots) {
        const uint8_t* src = in + offsets[chunk];
        const uint64_t elen = sizes[chunk];
        uint8_t* dst = out + chunk * out_stride;
        const uint64_t room_all = out_total - chunk * out_stride;
        const uint64_t cap = room_all < out_stride ? room_all : out_stride;
        if (chunk != slot) {
            uint4* p = reinterpret_cast<uint4*>(tables + slot * kTableBytes);
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (uint64_t i = lane; i < kTableBytes / 16; i += 64) p[i] = z;
            __threadfence();
        }
        uint32_t last_hash = 0;
        Guard guard;
        uint64_t ipos = 0, opos = 0;
        bool bad = false, done = false;
        // ---- whole records / whole raw blocks with room to spare: one per step across the wave ----
        while (elen - ipos >= kMaxRecord && cap - opos >= G::kBlock) {
            if (guard.block_is_copy()) {                                      // codec.rs:89-91
                if (act) st32u(dst + opos + 4u * lane, ld32u(src + ipos + 4u * lane));
                ipos += G::kBlock; opos += G::kBlock;
                guard.decay();                                                // (more than a block left: kMaxRecord > kBlock)
                continue;
            }
            const uint8_t* rec = src + ipos;
            const uint64_t sig = (uint64_t)ld32u(rec) | ((uint64_t)ld32u(rec + 4) << 32);   // codec.rs:28-31
            const uint32_t flag = act ? (uint32_t)(sig >> (2u * (lane & 31u))) & 3u : 3u;
            const uint32_t ilen = !act ? 0u : (flag == 0 ? 4u : (flag == 3 ? 0u : 2u));
            uint32_t items;
            const uint32_t off = scan32(ilen, lane, items);
            uint32_t q = 0, h = 0;
            if (ilen == 4) { q = ld32u(rec + G::kSig + off); h = hash16(q); } else if (ilen == 2) h = ld16u(rec + G::kSig + off);
            const bool toucher = act && flag != 3;                            // touches the dictionary and writes the predictor (cheetah.rs:68-103)
            tbl_drain();                                                      // the previous step's table stores are in L2
            const Pair e0 = toucher ? tbl_load_pair(t.dict + h) : Pair{0u, 0u};
            // (the stream ahead, touched early: the next steps' signature and item loads then come from L2)
            const uint32_t ahead = (act && elen - ipos >= 3 * kMaxRecord) ? ld32u(rec + 2 * kMaxRecord - 8 + 4u * lane) : 0u;
            // ---- runs of predicted quads: one dependent predictor read per round, every run of the record at once ----
            bool known = !act || flag != 3;
            for (uint32_t round = 0; round < 32; ++round) {                   // (a run has at most 32 quads)
                const uint32_t hp = bperm(lane ? lane - 1u : 0u, h);
                const uint32_t kpv = bperm(lane ? lane - 1u : 0u, known ? 1u : 0u);   // (unconditional: a lane masked off while the others
                const bool kp = lane == 0 || kpv != 0;                                 //  permute reads as 0 to them)
                if (!known && kp) {
                    q = tbl_load32(t.pred + (lane == 0 ? last_hash : hp));    // speculation: nobody earlier in this record wrote that slot
                    h = hash16(q);
                    known = true;
                }
                if (ballot64(!known) == 0) break;
            }
            const uint32_t hprev = bperm(lane ? lane - 1u : 0u, h);
            const uint32_t ps = lane == 0 ? last_hash : hprev;
            // ---- dictionary: in dependency order among the lanes that touch it ----
            const uint32_t touchers = (uint32_t)ballot64(toucher);
            uint32_t peq, deq;
            same_key_masks2(ps, act, h, toucher, lane, peq, deq);
            const uint32_t dbefore = deq & below;
            const uint32_t dprev = dbefore ? 31u - (uint32_t)__builtin_clz(dbefore) : 64u;
            const bool dlast = toucher && (deq >> (lane & 31u) >> 1) == 0;
            uint32_t da = e0.a, db = e0.b, ddirty = 0;
            bool ddone = !toucher;
            for (uint32_t round = 0; round < 32; ++round) {
                const uint32_t done_mask = (uint32_t)ballot64(ddone && toucher);
                const bool ready = !ddone && (dprev == 64u || ((done_mask >> dprev) & 1u));
                const uint32_t fda = bperm(dprev & 31u, da), fdb = bperm(dprev & 31u, db), fdd = bperm(dprev & 31u, ddirty);
                if (ready) {
                    if (dprev != 64u) { da = fda; db = fdb; ddirty = fdd; }
                    if (flag == 0) { db = da; da = q; ddirty = 1; }
                    else if (flag == 1) q = da;
                    else { q = db; db = da; da = q; ddirty = 1; }
                    ddone = true;
                }
                if (ballot64(!ddone) == 0) break;
            }
            // ---- predictor: a predicted quad must see what the nearest earlier writer of its slot wrote ----
            const uint32_t wbefore = peq & touchers & below;
            const uint32_t wprev = wbefore ? 31u - (uint32_t)__builtin_clz(wbefore) : 64u;
            const uint32_t wq = bperm(wprev & 31u, q);
            const bool wrong = act && flag == 3 && wprev != 64u && wq != q;
            if (ballot64(wrong) != 0) {
                // mis-speculated: nothing has been written yet; the scalar code decodes this record from the tables as they stand
                __threadfence();                                              // (it reads them with plain loads)
                if (lane == 0) { t.last_hash = last_hash; bad = cheetah_record_scalar(t, src, elen, ipos, dst, cap, opos, done, guard); last_hash = t.last_hash; }
                __threadfence();
                ipos = bcast64(ipos); opos = bcast64(opos); last_hash = rfl(last_hash);
                bad = rfl(bad ? 1u : 0u) != 0; done = rfl(done ? 1u : 0u) != 0;
                guard.penalty = rfl(guard.penalty); guard.start = rfl(guard.start); guard.prev = rfl(guard.prev); guard.counter = rfl(guard.counter);
                if (bad || done) break;
                continue;
            }
            const bool plast = toucher && ((peq & touchers) >> (lane & 31u) >> 1) == 0;
            if (act) st32u(dst + opos + 4u * lane, q);
            if (plast) tbl_store32(t.pred + ps, q);
            if (dlast && ddirty) tbl_store_pair(t.dict + h, Pair{da, db});
            last_hash = rfl(bperm(31u, h));
            const uint32_t rlen = G::kSig + items;
            guard.update(rlen >= G::kBlock);                                  // codec.rs:98
            ipos += rlen; opos += G::kBlock;
            asm volatile("" : : "v"(ahead));
        }
        tbl_drain();
        // ---- the rest (short of a whole record with room to spare): the scalar code on lane 0, codec.rs:102-123 ----
        __threadfence();
        if (lane == 0) {
            t.last_hash = last_hash;
            while (ipos < elen && !bad && !done) {
                const uint64_t rem = elen - ipos;
                if (guard.block_is_copy()) {
                    const uint32_t take = rem > G::kBlock ? G::kBlock : (uint32_t)rem;
                    if (opos + take > cap) { bad = true; break; }
                    for (uint32_t i = 0; i < take; ++i) dst[opos + i] = src[ipos + i];
                    ipos += take; opos += take;
                    if (rem <= G::kBlock) break;
                    guard.decay();
                    continue;
                }
                bad = cheetah_record_scalar(t, src, elen, ipos, dst, cap, opos, done, guard);
            }
            if (exact && !bad && opos != cap) bad = true;
            produced[chunk] = opos;
            if (bad) atomicOr(err, 1u);
        }
        __threadfence();
    }
}


// =================================================================================================================
// Lion, one wave per chunk stream: the same scheme on records of 16 quads (lion.rs:17-27) with a 5-entry move-to-front row per
// predictor slot (lion.rs:50-57,211-270).  Every quad reads and (unless it hits the front entry) rewrites its row, so the row is
// what travels along a chain of lanes on the same predictor slot; the dictionary is touched only by quads no entry predicted.
// =================================================================================================================
struct Row5 { uint32_t n[5]; };
__device__ __forceinline__ uint32_t rlane32(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ __forceinline__ Row5 row_load(const uint32_t* p) {
    Row5 r;
#pragma unroll
    for (int i = 0; i < 5; ++i) r.n[i] = tbl_load32(p + i);
    return r;
}
// The same as TWO memory instructions (16 + 4 bytes; a row is 4-byte aligned): for a wave that is the only one to touch its tables.  The loads are the caller's to
// wait for (s_waitcnt vmcnt(0)): the compiler does not count an asm statement's
__device__ __forceinline__ void row_load_wide(const uint32_t* p, bool on, u32x4& lo, uint32_t& hi) {   // (the lanes that are not `on` keep what they hold)
    if (on) asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dword %1, %2, off offset:16" : "+v"(lo), "+v"(hi) : "v"(p) : "memory");
}
__device__ __forceinline__ void row_store_wide(uint32_t* p, const Row5& r) {
    const u32x4 lo = {r.n[0], r.n[1], r.n[2], r.n[3]};
    asm volatile("global_store_dwordx4 %0, %1, off\n\tglobal_store_dword %0, %2, off offset:16" ::"v"(p), "v"(lo), "v"(r.n[4]) : "memory");
}
__device__ __forceinline__ void row_store(uint32_t* p, const Row5& r) {
#pragma unroll
    for (int i = 0; i < 5; ++i) tbl_store32(p + i, r.n[i]);
}
__device__ __forceinline__ Row5 row_from_lane(uint32_t src, const Row5& r) {
    Row5 o;
#pragma unroll
    for (int i = 0; i < 5; ++i) o.n[i] = bperm(src, r.n[i]);
    return o;
}
// entry k (0..4) moves to the front, or q enters at the front and the last entry leaves (k == 5): lion.rs:240-262, :50-57
__device__ __forceinline__ void row_promote(Row5& r, uint32_t k, uint32_t q) {
#pragma unroll
    for (int i = 4; i > 0; --i) r.n[i] = (uint32_t)i <= k ? r.n[i - 1] : r.n[i];
    r.n[0] = q;
}
// 3-bit flags of lanes 0..15, one bit plane: bit i -> bit 3i
__device__ __forceinline__ uint64_t spread16by3(uint32_t x16) {
    uint64_t x = x16 & 0xffffu;
    x = (x | (x << 16)) & 0x00ff0000ff0000ffull;
    x = (x | (x << 8)) & 0xf00f00f00f00f00full;
    x = (x | (x << 4)) & 0x30c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x9249249249249249ull;
    return x;
}

__global__ __launch_bounds__(64) void lion_encode_wave(const uint8_t* __restrict__ in, uint64_t total, uint64_t chunk_bytes,
                                                       uint32_t n_chunks, uint8_t* __restrict__ out, uint64_t out_stride,
                                                       uint64_t* __restrict__ sizes, uint8_t* __restrict__ tables, uint32_t n_slots,
                                                       const uint32_t* __restrict__ only, uint32_t* __restrict__ head_state, uint32_t head_bytes,
                                                       uint32_t head_calm, const uint32_t* __restrict__ tail_state) {
    // only / head_state / tail_state: as cheetah_encode_wave (the chunks handed back by, the heads before and the ragged ends behind the
    // exchange passes of exchange_stages.hip)
    using G = Geo<DENSITY_HIP_LION>;
    const uint32_t slot = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (slot >= n_slots) return;
    constexpr uint64_t kTableBytes = 65536ull * (sizeof(Pair) + 4ull * G::kPredWords);
    Tables<DENSITY_HIP_LION> t;
    t.dict = reinterpret_cast<Pair*>(tables + slot * kTableBytes);
    t.pred = reinterpret_cast<uint32_t*>(tables + slot * kTableBytes + 65536ull * sizeof(Pair));
    const bool act16 = lane < 16;
    const uint32_t myrec = lane >> 4;
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint64_t chunk = slot; chunk < n_chunks; chunk += n_slots) {
        const uint8_t* src = in + chunk * chunk_bytes;
        const uint64_t len = (total - chunk * chunk_bytes) < chunk_bytes ? (total - chunk * chunk_bytes) : chunk_bytes;
        uint8_t* dst = out + chunk * out_stride;
        if (only && !only[chunk]) continue;
        if (tail_state && !tail_state[8 * chunk + 6]) continue;
        if ((chunk != slot || only || head_state) && !tail_state) {
            uint4* p = reinterpret_cast<uint4*>(tables + slot * kTableBytes);
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (uint64_t i = lane; i < kTableBytes / 16; i += 64) p[i] = z;
            __threadfence();
        }
        uint32_t last_hash = 0;
        Guard guard;
        uint64_t opos = 0, pos = 0;
        if (tail_state) {
            const uint32_t* ts = tail_state + 8 * chunk;
            pos = ts[0]; opos = ts[1]; last_hash = ts[2];
            guard.prev = ts[3]; guard.start = ts[4]; guard.counter = ts[5];
        }
        // (Round 4.)  FOUR blocks per step — the whole wave, lanes 16r .. 16r+15 take block r — where they are there and no hand-over boundary lies between
        // them: a step is one gather of rows and pairs, their resolution across lanes and the stores, and costs the same latency for 64 quads as for 16 (the
        // key matches and the chains of one slot work on 64 lanes as they did on 16).  Whether a block behind the first is coded at all depends on the length
        // of the record in front of it (codec.rs:35-37,68), which only the resolution gives: it is resolved on the assumption that it is, and the blocks from
        // the first one the blow-up protection wants copied on are dropped — their lanes store nothing, the slots they share keep the last writers of the
        // blocks that stand (`live`) — and the next step starts with the dropped block.
        auto window = [&](uint64_t at) -> uint32_t { return (at + 4u * lane + 4u <= len) ? ld32u(src + at + 4u * lane) : 0u; };
        uint32_t qwin = window(pos);                                             // the quads of the next four blocks
        bool may_hand_over = head_state && len >= 4ull * head_bytes, handed_over = false;   // (as cheetah_encode_wave)
        uint64_t last_copy_end = 0;
        while (pos + G::kBlock <= len) {
            if (may_hand_over && pos >= head_bytes && (pos & 4095u) == 0) {
                if (pos >= last_copy_end + head_calm && guard.penalty == 0) { handed_over = true; break; }
                if (pos >= 4ull * head_bytes || pos >= len / 2) may_hand_over = false;   // raw copies this far in are not the cold start's: no hand-over
            }
            if (guard.block_is_copy()) {                                      // codec.rs:35-37
                last_copy_end = pos + G::kBlock;
                if (act16) st32u(dst + opos + 4u * lane, qwin);
                pos += G::kBlock; opos += G::kBlock;
                qwin = window(pos);
                guard.decay();
                continue;
            }
            uint32_t nblk = 1;                                                    // blocks the step takes on: whole ones, up to a hand-over boundary
            while (nblk < 4u && pos + (nblk + 1u) * G::kBlock <= len && !(may_hand_over && ((pos + nblk * G::kBlock) & 4095u) == 0)) ++nblk;
            const uint32_t nact = 16u * nblk;
            const bool act = lane < nact;
            const uint32_t q = qwin;
            const uint32_t h = hash16(q);
            const uint32_t hprev = bperm(lane ? lane - 1u : 0u, h);
            const uint32_t ps = lane == 0 ? last_hash : hprev;               // lion.rs:213,268
            tbl_drain();
            // (the row as 16 + 4 bytes, two memory instructions instead of five: at three streams a CU the kernel is bound by the number of requests the memory system
            // takes.  The pair is asked for BEHIND the row and loads return in order: once the compiler has waited for the pair — it must, before the statement that
            // names it — the row is in as well)
            u32x4 row_lo = {0u, 0u, 0u, 0u};
            uint32_t row_hi = 0u;
            row_load_wide(t.pred + 5u * ps, act, row_lo, row_hi);
            Pair e0 = act ? tbl_load_pair(t.dict + h) : Pair{0u, 0u};
            const uint32_t qwin_next = window(pos + 4u * nact);                // (on the assumption that the step takes all its blocks)
            const uint64_t peq = same_key_mask64(ps, act), deq = same_key_mask64(h, act);
            asm volatile("" : "+v"(row_lo), "+v"(row_hi), "+v"(e0.a), "+v"(e0.b));
            Row5 row = Row5{{row_lo.x, row_lo.y, row_lo.z, row_lo.w, row_hi}};
            uint32_t da = e0.a, db = e0.b, pdirty = 0, ddirty = 0;
            const uint64_t pbefore = peq & below, dbefore = deq & below;
            const uint32_t pprev = pbefore ? 63u - (uint32_t)__builtin_clzll(pbefore) : 64u;
            const uint32_t dprev = dbefore ? 63u - (uint32_t)__builtin_clzll(dbefore) : 64u;
            uint32_t flag = 0;
            bool done = !act;
            for (uint32_t round = 0; round < 64; ++round) {                   // (a chain has at most 64 links)
                const uint64_t done_mask = ballot64(done && act);
                const bool pok = pprev == 64u || ((done_mask >> (pprev & 63u)) & 1ull), dok = dprev == 64u || ((done_mask >> (dprev & 63u)) & 1ull);
                const bool ready = !done && pok && dok;
                const Row5 frow = row_from_lane(pprev & 63u, row);
                const uint32_t fpd = bperm(pprev & 63u, pdirty);
                const uint32_t fda = bperm(dprev & 63u, da), fdb = bperm(dprev & 63u, db), fdd = bperm(dprev & 63u, ddirty);
                if (ready) {
                    if (pprev != 64u) { row = frow; pdirty = fpd; }
                    if (dprev != 64u) { da = fda; db = fdb; ddirty = fdd; }
                    if (row.n[0] == q) flag = 1;                              // lion.rs:211-270
                    else if (row.n[1] == q) { flag = 2; row_promote(row, 1, q); pdirty = 1; }
                    else if (row.n[2] == q) { flag = 3; row_promote(row, 2, q); pdirty = 1; }
                    else if (row.n[3] == q) { flag = 4; row_promote(row, 3, q); pdirty = 1; }
                    else {
                        if (row.n[4] == q) flag = 5;
                        else if (da == q) flag = 6;
                        else { flag = db == q ? 7u : 0u; db = da; da = q; ddirty = 1; }
                        row_promote(row, 4, q); pdirty = 1;                   // shift_predictions (a hit on the last entry included)
                    }
                    done = true;
                }
                if (ballot64(!done) == 0) break;
            }
            const uint32_t ilen = !act ? 0u : (flag == 0 ? 4u : (flag >= 6 ? 2u : 0u));
            uint32_t incl = ilen;                                                 // item bytes of my record up to and with mine: a sum over the lanes of my row of 16
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
            const uint32_t rl0 = G::kSig + rlane32(incl, 15), rl1 = G::kSig + rlane32(incl, 31), rl2 = G::kSig + rlane32(incl, 47), rl3 = G::kSig + rlane32(incl, 63);
            // which blocks stand: codec.rs:68 for a record, :35 for the block behind it
            Guard g = guard;
            g.update(rl0 >= G::kBlock);
            uint32_t nlive = 1;
            if (nblk > 1) { Guard gc = g; if (!gc.block_is_copy()) { g = gc; g.update(rl1 >= G::kBlock); nlive = 2; } }
            if (nblk > 2 && nlive == 2) { Guard gc = g; if (!gc.block_is_copy()) { g = gc; g.update(rl2 >= G::kBlock); nlive = 3; } }
            if (nblk > 3 && nlive == 3) { Guard gc = g; if (!gc.block_is_copy()) { g = gc; g.update(rl3 >= G::kBlock); nlive = 4; } }
            const uint64_t live = nlive == 4 ? ~0ull : (1ull << (16u * nlive)) - 1ull;   // the lanes whose work stands
            const bool mine = act && myrec < nlive;
            const bool plast = mine && (((peq & live) >> lane) >> 1) == 0, dlast = mine && (((deq & live) >> lane) >> 1) == 0;
            uint8_t* rec = dst + opos;
            const uint32_t at1 = rl0, at2 = rl0 + rl1, at3 = rl0 + rl1 + rl2;      // where records 1..3 start behind `rec`
            const uint64_t f0 = ballot64(act && (flag & 1u)), f1 = ballot64(act && (flag & 2u)), f2 = ballot64(act && (flag & 4u));
            {   // lion.rs:334-337: 6 bytes per signature; lanes 3r .. 3r+2 store record r's
                const uint32_t r = lane / 3u, part = lane - 3u * r;
                const uint32_t sh = 16u * (r & 3u);
                const uint64_t sig = spread16by3((uint32_t)(f0 >> sh)) | (spread16by3((uint32_t)(f1 >> sh)) << 1) | (spread16by3((uint32_t)(f2 >> sh)) << 2);
                const uint32_t at = r == 0 ? 0u : r == 1 ? at1 : r == 2 ? at2 : at3;
                if (lane < 3u * nlive) st16u(rec + at + 2u * part, (uint32_t)(sig >> (16u * part)) & 0xffffu);
            }
            const uint32_t myat = myrec == 0 ? 0u : myrec == 1 ? at1 : myrec == 2 ? at2 : at3;
            uint8_t* ip = rec + myat + G::kSig + (incl - ilen);
            if (mine) { if (ilen == 4) st32u(ip, q); else if (ilen == 2) st16u(ip, h); }
            if (plast && pdirty) row_store_wide(t.pred + 5u * ps, row);
            if (dlast && ddirty) tbl_store_pair(t.dict + h, Pair{da, db});
            last_hash = rlane32(h, 16u * nlive - 1u);
            guard = g;
            opos += nlive == 1 ? rl0 : nlive == 2 ? at2 : nlive == 3 ? at3 : at3 + rl3;
            pos += (uint64_t)nlive * G::kBlock;
            qwin = nlive == nblk ? qwin_next : window(pos);                       // (blocks resolved in vain: their quads again, from the next step's first on)
        }
        tbl_drain();
        if (head_state) {
            if (lane == 0) {
                head_state[8 * chunk + 0] = (uint32_t)opos;
                head_state[8 * chunk + 1] = last_hash;
                head_state[8 * chunk + 2] = (guard.penalty ? 1u : 0u) | (guard.prev ? 2u : 0u);
                head_state[8 * chunk + 3] = handed_over ? 0u : 2u;             // 2: the chunk is finished, nothing for the passes
                head_state[8 * chunk + 4] = guard.start;
                head_state[8 * chunk + 5] = guard.counter;
                head_state[8 * chunk + 6] = (uint32_t)pos;                    // where the passes take over
            }
            if (handed_over) continue;
        }
        if (pos < len) {                                                      // the ragged last block: scalar code, lane 0
            __threadfence();
            if (lane == 0) {
                t.last_hash = last_hash;
                const uint32_t blen = (uint32_t)(len - pos);
                const uint8_t* blk = src + pos;
                if (guard.block_is_copy()) {
                    for (uint32_t i = 0; i < blen; ++i) dst[opos + i] = blk[i];
                    opos += blen;
                } else {
                    uint8_t* rec = dst + opos;
                    uint64_t o = G::kSig, sig = 0;
                    const uint32_t nq = blen >> 2;
                    for (uint32_t k = 0; k < nq; ++k) {
                        uint32_t item = 0, il = 0;
                        const uint32_t flag = enc_quad(t, ld32u(blk + 4u * k), item, il);
                        sig |= (uint64_t)flag << (G::kFlagBits * k);
                        if (il == 2) st16u(rec + o, item); else if (il == 4) st32u(rec + o, item);
                        o += il;
                    }
                    for (uint32_t i = 4u * nq; i < blen; ++i) rec[o++] = blk[i];
                    store_sig<DENSITY_HIP_LION>(rec, sig);
                    opos += o;
                }
            }
            opos = bcast64(opos);
            __threadfence();
        }
        if (lane == 0) sizes[chunk] = opos;
    }
}


// one coded Lion record by the scalar code (lane 0; codec.rs:92-99,111-123, lion.rs:292-314)
__device__ __forceinline__ bool lion_record_scalar(Tables<DENSITY_HIP_LION>& t, const uint8_t* src, uint64_t elen, uint64_t& ipos,
                                                   uint8_t* dst, uint64_t cap, uint64_t& opos, bool& done, Guard& guard) {
    using G = Geo<DENSITY_HIP_LION>;
    if (elen - ipos < G::kSig) return true;
    const uint64_t mark = ipos;
    uint64_t sig = load_sig<DENSITY_HIP_LION>(src + ipos);
    ipos += G::kSig;
    for (uint32_t k = 0; k < G::kBlock / 4; ++k) {
        const uint32_t flag = (uint32_t)(sig & 7u);
        sig >>= 3;
        const uint64_t left = elen - ipos;
        if (flag == 0 && left < 4) {                                          // lion.rs:295-303: end of data
            if (opos + left > cap) return true;
            for (uint32_t i = 0; i < left; ++i) dst[opos + i] = src[ipos + i];
            opos += left; ipos += left; done = true;
            return false;
        }
        const uint32_t need = item_bytes(t, flag);
        if (left < need || opos + 4 > cap) return true;
        const uint32_t q = dec_quad(t, flag, src + ipos);
        ipos += need;
        st32u(dst + opos, q);
        opos += 4;
    }
    guard.update(ipos - mark >= G::kBlock);
    return false;
}

// ---- round 4: FOUR records per step (the whole wave: lanes 16r .. 16r+15 take record r) ----
// What a step costs is its chain of dependent memory reads — signature, items, one table read per link of a run of predicted quads, the rows — not its
// lanes; where a record's successor starts follows from its signature alone (4 bytes per PLAIN flag, 2 per dictionary flag: lion.rs:317-325), so the four
// signatures are a chain of four reads of lines the touch-ahead has brought in, and everything behind them — items, dictionary pairs, runs of predicted
// quads, rows, the key matches, the repair walk — is done once for 64 quads.
__device__ __forceinline__ uint32_t lion_item_bytes(uint64_t sig) {                  // of a 48-bit signature: 4 per flag 0, 2 per flag 6 / 7, none per predicted one
    constexpr uint64_t kLow = 0x0000249249249249ull;                                // bit 0 of each of the 16 three-bit flags
    const uint64_t b0 = sig & kLow, b1 = (sig >> 1) & kLow, b2 = (sig >> 2) & kLow;
    return 4u * (uint32_t)__builtin_popcountll(~(b0 | b1 | b2) & kLow) + 2u * (uint32_t)__builtin_popcountll(b2 & b1);
}
__device__ __forceinline__ uint64_t lion_sig_at(const uint8_t* p) {                  // lion.rs:340-351, as a scalar
    return (((uint64_t)rfl(ld32u(p + 4)) << 32) | rfl(ld32u(p))) & 0xffffffffffffull;
}
__global__ __launch_bounds__(64) void lion_decode_wave(const uint8_t* __restrict__ in, const uint64_t* __restrict__ offsets,
                                                        const uint64_t* __restrict__ sizes, uint32_t n_chunks,
                                                        uint8_t* __restrict__ out, uint64_t out_stride, uint64_t out_total,
                                                        uint32_t exact, uint64_t* __restrict__ produced, uint32_t* __restrict__ err,
                                                        uint8_t* __restrict__ tables, uint32_t n_slots) {
    using G = Geo<DENSITY_HIP_LION>;
    const uint32_t slot = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (slot >= n_slots) return;
    constexpr uint64_t kTableBytes = 65536ull * (sizeof(Pair) + 4ull * G::kPredWords);
    constexpr uint32_t kMaxRecord = 8 + G::kBlock;                            // 6 + 16 x 4, and the signature is fetched as 8 bytes
    Tables<DENSITY_HIP_LION> t;
    t.dict = reinterpret_cast<Pair*>(tables + slot * kTableBytes);
    t.pred = reinterpret_cast<uint32_t*>(tables + slot * kTableBytes + 65536ull * sizeof(Pair));
    const uint32_t myrec = lane >> 4, k16 = lane & 15u;
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint64_t chunk = slot; chunk < n_chunks; chunk += n_slots) {
        const uint8_t* src = in + offsets[chunk];
        const uint64_t elen = sizes[chunk];
        uint8_t* dst = out + chunk * out_stride;
        const uint64_t room_all = out_total - chunk * out_stride;
        const uint64_t cap = room_all < out_stride ? room_all : out_stride;
        if (chunk != slot) {
            uint4* p = reinterpret_cast<uint4*>(tables + slot * kTableBytes);
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (uint64_t i = lane; i < kTableBytes / 16; i += 64) p[i] = z;
            __threadfence();
        }
        uint32_t last_hash = 0;
        Guard guard;
        uint64_t ipos = 0, opos = 0;
        bool bad = false, done = false;
        uint32_t ahead = 0;                                                       // (a touch-ahead load's value: never looked at)
        while (elen - ipos >= kMaxRecord && cap - opos >= G::kBlock) {
            if (guard.block_is_copy()) {                                      // codec.rs:89-91
                if (lane < 16) st32u(dst + opos + 4u * lane, ld32u(src + ipos + 4u * lane));
                ipos += G::kBlock; opos += G::kBlock;
                guard.decay();
                continue;
            }
            // the step's records: the first, and up to three more while each is whole, has room and the FSM lets it be coded (codec.rs:88-99)
            uint64_t sg0 = lion_sig_at(src + ipos), sg1 = 0, sg2 = 0, sg3 = 0;
            uint32_t at1 = 0, at2 = 0, at3 = 0;                                    // where records 1..3 start, from ipos
            uint32_t nrec = 1;
            Guard g = guard;
            uint32_t len = G::kSig + lion_item_bytes(sg0);                        // stream bytes of the step so far
            g.update(len >= G::kBlock);                                           // codec.rs:98
            {
                Guard gc = g;
                if (elen - ipos - len >= kMaxRecord && cap - opos >= 2u * G::kBlock && !gc.block_is_copy()) {
                    g = gc; at1 = len; sg1 = lion_sig_at(src + ipos + len);
                    const uint32_t rl = G::kSig + lion_item_bytes(sg1);
                    g.update(rl >= G::kBlock); len += rl; nrec = 2;
                }
            }
            if (nrec == 2) {
                Guard gc = g;
                if (elen - ipos - len >= kMaxRecord && cap - opos >= 3u * G::kBlock && !gc.block_is_copy()) {
                    g = gc; at2 = len; sg2 = lion_sig_at(src + ipos + len);
                    const uint32_t rl = G::kSig + lion_item_bytes(sg2);
                    g.update(rl >= G::kBlock); len += rl; nrec = 3;
                }
            }
            if (nrec == 3) {
                Guard gc = g;
                if (elen - ipos - len >= kMaxRecord && cap - opos >= 4u * G::kBlock && !gc.block_is_copy()) {
                    g = gc; at3 = len; sg3 = lion_sig_at(src + ipos + len);
                    const uint32_t rl = G::kSig + lion_item_bytes(sg3);
                    g.update(rl >= G::kBlock); len += rl; nrec = 4;
                }
            }
            const uint32_t nact = 16u * nrec;
            const bool act = lane < nact;
            const uint64_t sig = myrec == 0 ? sg0 : myrec == 1 ? sg1 : myrec == 2 ? sg2 : sg3;
            const uint32_t at = myrec == 0 ? 0u : myrec == 1 ? at1 : myrec == 2 ? at2 : at3;
            const uint32_t flag = act ? (uint32_t)(sig >> (3u * k16)) & 7u : 1u;
            const uint32_t ilen = !act ? 0u : (flag == 0 ? 4u : (flag >= 6 ? 2u : 0u));
            uint32_t incl = ilen;                                                 // where my item lies behind my record's signature: a sum over the lanes of my row of 16
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
            incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
            const uint8_t* ibase = src + ipos + at + G::kSig + (incl - ilen);
            uint32_t q = 0, h = 0;
            if (ilen == 4) { q = ld32u(ibase); h = hash16(q); } else if (ilen == 2) h = ld16u(ibase);
            const bool dtouch = act && (flag == 0 || flag >= 6);              // touches the dictionary (lion.rs:85-186)
            const bool predicted = act && flag >= 1 && flag <= 5;
            // The stream is read 50-70 bytes at a time, each record's place known only from the one before: nothing fetches it ahead, and the signature and
            // the items were two dependent misses per record.  Four lanes touch the lines half a KiB on, in the shadow of the table reads below (loads
            // return in order: issued here, not in front of the signature loads).  The touch's register is held — as an operand of the drain of the NEXT
            // step, by which time it has long landed — so that nothing else lives where the load lands.  (The drain: the previous step's table stores are in L2.)
            // (Round 5: the touch is a load the COMPILER sees — until now a hand-issued one into a C variable, which nothing kept the register allocator
            // from copying or re-using between this statement and the next step's wait.  Its only use is the next step's empty statement below, so
            // the compiler waits for it there, by itself, and nowhere earlier.)
            asm volatile("s_waitcnt vmcnt(0)" : : "v"(ahead) : "memory");
            {
                const uint64_t far = ipos + 512u + 128u * (lane & 3u);
                const uint8_t* pa = src + (far + 4 <= elen ? far : ipos);
                ahead = lane < 4 ? ld32u(pa) : 0u;
            }
            const Pair e0 = dtouch ? tbl_load_pair(t.dict + h) : Pair{0u, 0u};
            // ---- runs of predicted quads: one dependent read per round; speculation: nobody earlier in this step rewrote that row ----
            bool known = !predicted;
            for (uint32_t round = 0; round < 64; ++round) {
                const uint32_t hp = bperm(lane ? lane - 1u : 0u, h);
                const uint32_t kpv = bperm(lane ? lane - 1u : 0u, known ? 1u : 0u);
                const bool kp = lane == 0 || kpv != 0;
                if (!known && kp) {
                    q = tbl_load32(t.pred + 5u * (lane == 0 ? last_hash : hp) + (flag - 1u));
                    h = hash16(q);
                    known = true;
                }
                if (ballot64(!known) == 0) break;
            }
            const uint32_t hprev = bperm(lane ? lane - 1u : 0u, h);
            const uint32_t ps = lane == 0 ? last_hash : hprev;
            Row5 row = act ? row_load(t.pred + 5u * ps) : Row5{{0u, 0u, 0u, 0u, 0u}};
            const Row5 row_mem = row;                                             // as memory holds it (the repair below starts over from it)
            // ---- dictionary, in dependency order among the lanes that touch it ----
            const uint64_t peq = same_key_mask64(ps, act), deq = same_key_mask64(h, dtouch);
            const uint64_t dbefore = deq & below;
            const uint32_t dprev = dbefore ? 63u - (uint32_t)__builtin_clzll(dbefore) : 64u;
            const bool dlast = dtouch && ((deq >> lane) >> 1) == 0;
            uint32_t da = e0.a, db = e0.b, ddirty = 0;
            bool ddone = !dtouch;
            for (uint32_t round = 0; round < 64; ++round) {
                const uint64_t done_mask = ballot64(ddone && dtouch);
                const bool ready = !ddone && (dprev == 64u || ((done_mask >> (dprev & 63u)) & 1ull));
                const uint32_t fda = bperm(dprev & 63u, da), fdb = bperm(dprev & 63u, db), fdd = bperm(dprev & 63u, ddirty);
                if (ready) {
                    if (dprev != 64u) { da = fda; db = fdb; ddirty = fdd; }
                    if (flag == 0) { db = da; da = q; ddirty = 1; }
                    else if (flag == 6) q = da;
                    else { q = db; db = da; da = q; ddirty = 1; }
                    ddone = true;
                }
                if (ballot64(!ddone) == 0) break;
            }
            // ---- predictor rows, in dependency order (every quad rewrites its row unless it hit the front entry) ----
            const uint64_t pbefore = peq & below;
            const uint32_t pprev = pbefore ? 63u - (uint32_t)__builtin_clzll(pbefore) : 64u;
            const bool plast = act && ((peq >> lane) >> 1) == 0;
            uint32_t pdirty = 0;
            bool pdone = !act, wrong = false;
            for (uint32_t round = 0; round < 64; ++round) {
                const uint64_t done_mask = ballot64(pdone && act);
                const bool ready = !pdone && (pprev == 64u || ((done_mask >> (pprev & 63u)) & 1ull));
                const Row5 frow = row_from_lane(pprev & 63u, row);
                const uint32_t fpd = bperm(pprev & 63u, pdirty);
                if (ready) {
                    if (pprev != 64u) { row = frow; pdirty = fpd; }
                    if (predicted) {
                        uint32_t cur = row.n[0];
#pragma unroll
                        for (uint32_t k = 1; k < 5; ++k) cur = flag == k + 1u ? row.n[k] : cur;
                        wrong = cur != q;                                     // the row as it really stands does not hold what the speculation read
                        if (flag > 1) { row_promote(row, flag - 1u, q); pdirty = 1; }
                    } else {
                        row_promote(row, 4, q); pdirty = 1;                   // lion.rs:50-57
                    }
                    pdone = true;
                }
                if (ballot64(!pdone) == 0) break;
            }
            uint32_t psf = ps;                                                    // the predictor slot my row is stored to
            bool plastf = plast;
            if (ballot64(wrong) != 0) {
                // A speculation failed: a predicted quad read an entry that an earlier quad of this step has since moved.  Its real value — the entry of the
                // row as forwarded — has another hash, so the quad behind it sits in another context than assumed, and so on.  Up to round 3 the whole record
                // was decoded again by the scalar code on lane 0 (two dependent memory reads per quad: a fifth of the kernel's time on prose).  Now: ONE
                // exact walk over the step's quads in stream order, in registers — every value wave-uniform — with rows forwarded between quads of one
                // context, taken from the speculative gather where it was made at the right context, and read from memory only where the context turned
                // out to be another one (lion.rs:85-186).
                uint32_t ctx = last_hash;
                uint32_t cxv = 0xffffffffu, dirtyv = 0;                           // per lane, once walked: my true context; my row differs from memory
                Row5 rf = row_mem;
#pragma nounroll
                for (uint32_t i = 0; i < nact; ++i) {
                    const uint64_t m = ballot64(lane < i && cxv == ctx);
                    Row5 r;
                    uint32_t dirty = 0;
                    if (m) {                                                      // the latest earlier quad of this context hands its row on
                        const uint32_t j = 63u - (uint32_t)__builtin_clzll(m);
#pragma unroll
                        for (int k = 0; k < 5; ++k) r.n[k] = rlane32(rf.n[k], j);
                        dirty = rlane32(dirtyv, j);
                    } else if (rlane32(ps, i) == ctx) {                           // nobody before it in this step: memory's row, gathered at the right place
#pragma unroll
                        for (int k = 0; k < 5; ++k) r.n[k] = rlane32(row_mem.n[k], i);
                    } else {
                        r = row_load(t.pred + 5u * ctx);
                    }
                    const uint32_t f = rlane32(flag, i);
                    uint32_t qi, hi;
                    if (f >= 1u && f <= 5u) {
                        qi = r.n[0];
#pragma unroll
                        for (uint32_t k = 1; k < 5; ++k) qi = f == k + 1u ? r.n[k] : qi;
                        hi = hash16(qi);
                        if (f > 1u) { row_promote(r, f - 1u, qi); dirty = 1; }
                    } else {
                        qi = rlane32(q, i); hi = rlane32(h, i);                   // what the dictionary gave: no context in it
                        row_promote(r, 4, qi); dirty = 1;
                    }
                    if (lane == i) { q = qi; h = hi; rf = r; cxv = ctx; dirtyv = dirty; }
                    ctx = hi;
                }
                row = rf; pdirty = dirtyv; psf = cxv;
                const uint64_t peq2 = same_key_mask64(cxv, act);
                plastf = act && ((peq2 >> lane) >> 1) == 0;
            }
            if (act) st32u(dst + opos + 4u * lane, q);
            if (plastf && pdirty) row_store(t.pred + 5u * psf, row);
            if (dlast && ddirty) tbl_store_pair(t.dict + h, Pair{da, db});
            last_hash = rlane32(h, nact - 1u);
            guard = g;
            ipos += len; opos += (uint64_t)nrec * G::kBlock;
        }
        asm volatile("s_waitcnt vmcnt(0)" : : "v"(ahead) : "memory");                // tbl_drain() (the last touch-ahead is consumed here)
        __threadfence();
        if (lane == 0) {                                                      // the rest: scalar code, codec.rs:102-123
            t.last_hash = last_hash;
            while (ipos < elen && !bad && !done) {
                const uint64_t rem = elen - ipos;
                if (guard.block_is_copy()) {
                    const uint32_t take = rem > G::kBlock ? G::kBlock : (uint32_t)rem;
                    if (opos + take > cap) { bad = true; break; }
                    for (uint32_t i = 0; i < take; ++i) dst[opos + i] = src[ipos + i];
                    ipos += take; opos += take;
                    if (rem <= G::kBlock) break;
                    guard.decay();
                    continue;
                }
                bad = lion_record_scalar(t, src, elen, ipos, dst, cap, opos, done, guard);
            }
            if (exact && !bad && opos != cap) bad = true;
            produced[chunk] = opos;
            if (bad) atomicOr(err, 1u);
        }
        __threadfence();
    }
}

#ifdef LION_PHASES
__device__ unsigned long long g_lp[32];
#define LP_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define LP_ADD(i, v) do { if (slot == 7 && lane == 0) atomicAdd(&g_lp[i], (unsigned long long)(v)); } while (0)
#else
#define LP_T(x)
#define LP_ADD(i, v)
#endif
__global__ __launch_bounds__(128) void lion_decode_pair(const uint8_t* __restrict__ in, const uint64_t* __restrict__ offsets,
                                                        const uint64_t* __restrict__ sizes, uint32_t n_chunks,
                                                        uint8_t* __restrict__ out, uint64_t out_stride, uint64_t out_total,
                                                        uint32_t exact, uint64_t* __restrict__ produced, uint32_t* __restrict__ err,
                                                        uint8_t* __restrict__ tables, uint32_t n_slots) {
    using G = Geo<DENSITY_HIP_LION>;
    const uint32_t slot = blockIdx.x;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    if (slot >= n_slots) return;
    // Two waves, two ROLES (later in round 6; before: the waves took the steps in alternation, each parsing its step and then holding the tables — a turn then ended
    // with the acknowledgement of its table stores, 1.45 k cycles of a step's 11.6 k, and a hand-on, because the OTHER wave read the tables next).  Wave 0 PARSES:
    // signatures, the FSM, flags, items, hashes, the dictionary's key match — nothing of which needs a table — and the raw-copy blocks, step after step, and leaves
    // every step in a ring in LDS.  Wave 1 holds the TABLES, all the time: its loads follow its own stores in program order, so nothing is drained and nothing handed on.
    //   ring slot: 16 bytes per lane {flag | act << 3 | h << 16, q, the dictionary's key match (64 bits)} and a header {kind, quads, output position; at the end: stream
    //   position and FSM}; sy[0] = steps published, sy[1] = steps taken (kExit in either: the watchdog fired)
    constexpr uint32_t kRing = 4, kCoded = 0, kRawBlock = 1, kEnd = 2;
    __shared__ __attribute__((aligned(16))) uint32_t sy[4];
    __shared__ __attribute__((aligned(16))) uint32_t hdr[kRing][12];            // kind, quads, opos (2) | ipos (2), - , - | guard (4)
    __shared__ __attribute__((aligned(16))) u32x4 ring[kRing][64];
    constexpr uint32_t kExit = 0xffffffffu;
    const uint32_t sya = lds_addr(sy);
    auto peek = [&](uint32_t word) -> uint32_t {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(sya + 4u * word) : "memory");
        return rfl(v);
    };
    auto poke = [&](uint32_t word, uint32_t v) { asm volatile("ds_write_b32 %0, %1" ::"v"(sya + 4u * word), "v"(v) : "memory"); };
    // waits while word `w` is below `least` (a step count); false: leave (the watchdog fired, here or in the other wave).  A nap between looks: a waiting wave that
    // looks all the time takes issue slots from the working one
    auto await_count = [&](uint32_t w, uint32_t least) -> bool {
        for (uint32_t spins = 0;; ++spins) {
            const uint32_t v = peek(w);
            if (v == kExit) return false;
            if (v >= least) return true;
            if (spins > (1u << 24)) { if (lane == 0) atomicOr(err, 16u); poke(0, kExit); poke(1, kExit); return false; }
            __builtin_amdgcn_s_sleep(1);
        }
    };
    constexpr uint64_t kTableBytes = 65536ull * (sizeof(Pair) + 4ull * G::kPredWords);
    constexpr uint32_t kMaxRecord = 8 + G::kBlock;                            // 6 + 16 x 4, and the signature is fetched as 8 bytes
    Tables<DENSITY_HIP_LION> t;
    t.dict = reinterpret_cast<Pair*>(tables + slot * kTableBytes);
    t.pred = reinterpret_cast<uint32_t*>(tables + slot * kTableBytes + 65536ull * sizeof(Pair));
    const uint32_t myrec = lane >> 4, k16 = lane & 15u;
    const uint64_t below = (1ull << lane) - 1ull;
    for (uint64_t chunk = slot; chunk < n_chunks; chunk += n_slots) {
        const uint8_t* src = in + offsets[chunk];
        const uint64_t elen = sizes[chunk];
        uint8_t* dst = out + chunk * out_stride;
        const uint64_t room_all = out_total - chunk * out_stride;
        const uint64_t cap = room_all < out_stride ? room_all : out_stride;
        if (chunk != slot) {
            __syncthreads();                                                      // (the table wave is through with the chunk before: the parser gets here long before it)
            uint4* p = reinterpret_cast<uint4*>(tables + slot * kTableBytes);
            const uint4 z = make_uint4(0, 0, 0, 0);
            for (uint64_t i = threadIdx.x; i < kTableBytes / 16; i += 128) p[i] = z;
            __threadfence();
        }
        __syncthreads();                                                          // (both waves are through with the chunk before; the tables are clear)
        if (threadIdx.x < 4) sy[threadIdx.x] = 0u;
        __syncthreads();
        if (wave == 0) {
            // ================= the PARSER =================
            Guard guard;
            uint64_t ipos = 0, opos = 0;
            uint32_t ahead = 0;                                                   // (a touch-ahead load's value: never looked at)
            for (uint32_t s = 0;; ++s) {
                const uint32_t at_slot = s % kRing;
                const uint32_t ha = lds_addr(&hdr[at_slot][0]);
                if (!(elen - ipos >= kMaxRecord && cap - opos >= G::kBlock)) {    // the hot loop ends here: the rest is the table wave's (scalar code, codec.rs:102-123)
                    if (!await_count(1, s + 1u >= kRing ? s + 1u - kRing : 0u)) break;
                    const u32x4 h0 = {kEnd, 0u, (uint32_t)opos, (uint32_t)(opos >> 32)}, h1 = {(uint32_t)ipos, (uint32_t)(ipos >> 32), 0u, 0u},
                                h2 = {guard.penalty, guard.start, guard.prev, guard.counter};
                    asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32" ::"v"(ha), "v"(h0), "v"(h1), "v"(h2) : "memory");
                    poke(0, s + 1u);
                    break;
                }
                if (guard.block_is_copy()) {                                      // codec.rs:89-91: a raw block touches no table — copied here, the table wave skips it
                    if (lane < 16) st32u(dst + opos + 4u * lane, ld32u(src + ipos + 4u * lane));
                    if (!await_count(1, s + 1u >= kRing ? s + 1u - kRing : 0u)) break;
                    const u32x4 h0 = {kRawBlock, 0u, 0u, 0u};
                    asm volatile("ds_write_b128 %0, %1" ::"v"(ha), "v"(h0) : "memory");
                    poke(0, s + 1u);
                    guard.decay();
                    ipos += G::kBlock; opos += G::kBlock;
                    continue;
                }
                LP_T(c1);
                // the step's records: the first, and up to three more while each is whole, has room and the FSM lets it be coded (codec.rs:88-99)
                uint64_t sg0 = lion_sig_at(src + ipos), sg1 = 0, sg2 = 0, sg3 = 0;
                uint32_t at1 = 0, at2 = 0, at3 = 0;                                    // where records 1..3 start, from ipos
                uint32_t nrec = 1;
                Guard g = guard;
                uint32_t len = G::kSig + lion_item_bytes(sg0);                        // stream bytes of the step so far
                g.update(len >= G::kBlock);                                           // codec.rs:98
                {
                    Guard gc = g;
                    if (elen - ipos - len >= kMaxRecord && cap - opos >= 2u * G::kBlock && !gc.block_is_copy()) {
                        g = gc; at1 = len; sg1 = lion_sig_at(src + ipos + len);
                        const uint32_t rl = G::kSig + lion_item_bytes(sg1);
                        g.update(rl >= G::kBlock); len += rl; nrec = 2;
                    }
                }
                if (nrec == 2) {
                    Guard gc = g;
                    if (elen - ipos - len >= kMaxRecord && cap - opos >= 3u * G::kBlock && !gc.block_is_copy()) {
                        g = gc; at2 = len; sg2 = lion_sig_at(src + ipos + len);
                        const uint32_t rl = G::kSig + lion_item_bytes(sg2);
                        g.update(rl >= G::kBlock); len += rl; nrec = 3;
                    }
                }
                if (nrec == 3) {
                    Guard gc = g;
                    if (elen - ipos - len >= kMaxRecord && cap - opos >= 4u * G::kBlock && !gc.block_is_copy()) {
                        g = gc; at3 = len; sg3 = lion_sig_at(src + ipos + len);
                        const uint32_t rl = G::kSig + lion_item_bytes(sg3);
                        g.update(rl >= G::kBlock); len += rl; nrec = 4;
                    }
                }
                const uint32_t nact = 16u * nrec;
                const bool act = lane < nact;
                const uint64_t sig = myrec == 0 ? sg0 : myrec == 1 ? sg1 : myrec == 2 ? sg2 : sg3;
                const uint32_t at = myrec == 0 ? 0u : myrec == 1 ? at1 : myrec == 2 ? at2 : at3;
                const uint32_t flag = act ? (uint32_t)(sig >> (3u * k16)) & 7u : 1u;
                const uint32_t ilen = !act ? 0u : (flag == 0 ? 4u : (flag >= 6 ? 2u : 0u));
                uint32_t incl = ilen;                                                 // where my item lies behind my record's signature: a sum over the lanes of my row of 16
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xf, 0xf, true);   // row_shr:1
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xf, 0xf, true);   // row_shr:2
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xf, 0xf, true);   // row_shr:4
                incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xf, 0xf, true);   // row_shr:8
                const uint8_t* ibase = src + ipos + at + G::kSig + (incl - ilen);
                uint32_t q = 0, h = 0;
                if (ilen == 4) { q = ld32u(ibase); h = hash16(q); } else if (ilen == 2) h = ld16u(ibase);
                const bool dtouch = act && (flag == 0 || flag >= 6);              // touches the dictionary (lion.rs:85-186)
                const bool predicted = act && flag >= 1 && flag <= 5;
                // ---- AHEAD of my table turn: what needs no table.  (A first version also READ the step's dictionary pairs and the rows of the lanes whose context
                // is in the stream here, to warm their lines for the turn: measured against a build without those reads, 3.147 against 3.172 ms — nothing; what the
                // second wave buys is the parse and the items off the turn.)  The stream is touched half a KiB on, as in the one-wave kernel. ----
                {
                    const uint64_t far = ipos + 512u + 128u * (lane & 3u);
                    const uint8_t* pa = src + (far + 4 <= elen ? far : ipos);
                    ahead = lane < 4 ? ld32u(pa) : 0u;
                    asm volatile("" : : "v"(ahead));
                }
                const uint64_t deq = same_key_mask64(h, dtouch);                      // (who follows whom in the dictionary is in the stream: matched ahead of the turn)
                LP_T(c2);
                // the step into the ring (the count behind its payload: a wave's LDS operations execute as issued)
                if (!await_count(1, s + 1u >= kRing ? s + 1u - kRing : 0u)) break;
                LP_T(c2b);
                {
                    const u32x4 h0 = {kCoded, nact, (uint32_t)opos, (uint32_t)(opos >> 32)};
                    const u32x4 mine = {flag | (act ? 8u : 0u) | (h << 16), q, (uint32_t)deq, (uint32_t)(deq >> 32)};
                    asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %2, %3" ::"v"(ha), "v"(h0), "v"(lds_addr(&ring[at_slot][lane])), "v"(mine) : "memory");
                    poke(0, s + 1u);
                }
                LP_ADD(0, c2b - c2); LP_ADD(1, c2 - c1);
                guard = g;
                ipos += len; opos += (uint64_t)nrec * G::kBlock;
            }
            asm volatile("s_waitcnt vmcnt(0)" : : "v"(ahead) : "memory");
        } else {
            // ================= the TABLES =================
            uint32_t last_hash = 0;
            Guard guard;
            uint64_t ipos = 0, opos = 0;
            bool bad = false, done = false, mine_to_finish = false;
            for (uint32_t s = 0;; ++s) {
                LP_T(c2);
                if (!await_count(0, s + 1u)) break;
                LP_T(c3);
                const uint32_t at_slot = s % kRing;
                u32x4 h0, mine;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(h0), "=v"(mine) : "v"(lds_addr(&hdr[at_slot][0])), "v"(lds_addr(&ring[at_slot][lane])) : "memory");
                const uint32_t kind = rfl(h0.x);
                if (kind == kEnd) {
                    u32x4 h1, h2;
                    asm volatile("ds_read_b128 %0, %2 offset:16\n\tds_read_b128 %1, %2 offset:32\n\ts_waitcnt lgkmcnt(0)" : "=v"(h1), "=v"(h2) : "v"(lds_addr(&hdr[at_slot][0])) : "memory");
                    opos = (uint64_t)rfl(h0.z) | ((uint64_t)rfl(h0.w) << 32); ipos = (uint64_t)rfl(h1.x) | ((uint64_t)rfl(h1.y) << 32);
                    guard.penalty = rfl(h2.x); guard.start = rfl(h2.y); guard.prev = rfl(h2.z); guard.counter = rfl(h2.w);
                    mine_to_finish = true;
                    break;
                }
                poke(1, s + 1u);                                                  // (the slot is in registers: the parser may have it back)
                if (kind == kRawBlock) continue;
                const uint32_t nact = rfl(h0.y);
                opos = (uint64_t)rfl(h0.z) | ((uint64_t)rfl(h0.w) << 32);
                const uint32_t flag = mine.x & 7u;
                const bool act = (mine.x & 8u) != 0;
                uint32_t q = mine.y, h = mine.x >> 16;
                const uint64_t deq = (uint64_t)mine.z | ((uint64_t)mine.w << 32);
                const bool dtouch = act && (flag == 0 || flag >= 6);              // touches the dictionary (lion.rs:85-186)
                const bool predicted = act && flag >= 1 && flag <= 5;
                const Pair e0 = dtouch ? tbl_load_pair(t.dict + h) : Pair{0u, 0u};
                // ---- runs of predicted quads: one dependent read per round; speculation: nobody earlier in this step rewrote that row ----
                // (a link reads its ONE entry; whole rows here — 16 + 4 bytes for every predicted lane, the row load behind the chain then for the others only — measured
                // slower, 2.78 against 2.68 ms: a link's round trip grows with what it asks for)
                bool known = !predicted;
                for (uint32_t round = 0; round < 64; ++round) {
                    const uint32_t hp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x138, 0xf, 0xf, false);                   // wave_shr:1 (no LDS round trip in the link)
                    const uint32_t kpv = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)(known ? 1u : 0u), 0x138, 0xf, 0xf, false);
                    const bool kp = kpv != 0;                                         // (lane 0 keeps the old operand: 1)
                    if (!known && kp) {
                        q = tbl_load32(t.pred + 5u * (lane == 0 ? last_hash : hp) + (flag - 1u));
                        h = hash16(q);
                        known = true;
                    }
                    LP_ADD(10, 1);
                    if (ballot64(!known) == 0) break;
                }
                LP_T(c4);
                const uint32_t hprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x138, 0xf, 0xf, false);   // wave_shr:1
                const uint32_t ps = lane == 0 ? last_hash : hprev;
                // the rows: 16 + 4 bytes, two memory instructions (up to here five dwords in five: at three streams a CU the decoder is bound by the number of requests the
                // memory system takes, not by their size — 2.94 -> 2.68 ms), waited for behind the dictionary rounds, which run under the load
                u32x4 row_lo = {0u, 0u, 0u, 0u};
                uint32_t row_hi = 0u;
                row_load_wide(t.pred + 5u * ps, act, row_lo, row_hi);
#ifdef LION_PHASES
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the account's stamp wants the rows in; the shipped kernel lets the dictionary rounds run under their load)
#endif
                LP_T(c5);
                const uint64_t peq = same_key_mask64(ps, act);
                const uint64_t dbefore = deq & below;
                const uint32_t dprev = dbefore ? 63u - (uint32_t)__builtin_clzll(dbefore) : 64u;
                const bool dlast = dtouch && ((deq >> lane) >> 1) == 0;
                uint32_t da = e0.a, db = e0.b, ddirty = 0;
                bool ddone = !dtouch;
                if (dtouch && dprev == 64u) {                                         // the first of its slot in this step: memory's pair, nothing to be handed on
                    if (flag == 0) { db = da; da = q; ddirty = 1; }
                    else if (flag == 6) q = da;
                    else { q = db; db = da; da = q; ddirty = 1; }
                    ddone = true;
                }
                for (uint32_t round = 0; round < 64 && ballot64(!ddone) != 0; ++round) {
                    const uint64_t done_mask = ballot64(ddone && dtouch);
                    const bool ready = !ddone && (dprev == 64u || ((done_mask >> (dprev & 63u)) & 1ull));
                    const uint32_t fda = bperm(dprev & 63u, da), fdb = bperm(dprev & 63u, db), fdd = bperm(dprev & 63u, ddirty);
                    if (ready) {
                        if (dprev != 64u) { da = fda; db = fdb; ddirty = fdd; }
                        if (flag == 0) { db = da; da = q; ddirty = 1; }
                        else if (flag == 6) q = da;
                        else { q = db; db = da; da = q; ddirty = 1; }
                        ddone = true;
                    }
                    if (ballot64(!ddone) == 0) break;
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(row_lo), "+v"(row_hi) : : "memory");
                Row5 row = Row5{{row_lo.x, row_lo.y, row_lo.z, row_lo.w, row_hi}};
                const Row5 row_mem = row;                                             // as memory holds it (the repair below starts over from it)
                LP_T(c6);
                const uint64_t pbefore = peq & below;
                const uint32_t pprev = pbefore ? 63u - (uint32_t)__builtin_clzll(pbefore) : 64u;
                const bool plast = act && ((peq >> lane) >> 1) == 0;
                uint32_t pdirty = 0;
                bool pdone = !act, wrong = false;
                if (act && pprev == 64u) {                                            // the first of its context in this step: memory's row
                    if (predicted) {
                        if (flag > 1) { row_promote(row, flag - 1u, q); pdirty = 1; }   // (it read that row itself: never wrong)
                    } else {
                        row_promote(row, 4, q); pdirty = 1;
                    }
                    pdone = true;
                }
                for (uint32_t round = 0; round < 64 && ballot64(!pdone) != 0; ++round) {
                    const uint64_t done_mask = ballot64(pdone && act);
                    const bool ready = !pdone && (pprev == 64u || ((done_mask >> (pprev & 63u)) & 1ull));
                    const Row5 frow = row_from_lane(pprev & 63u, row);
                    const uint32_t fpd = bperm(pprev & 63u, pdirty);
                    if (ready) {
                        if (pprev != 64u) { row = frow; pdirty = fpd; }
                        if (predicted) {
                            uint32_t cur = row.n[0];
#pragma unroll
                            for (uint32_t k = 1; k < 5; ++k) cur = flag == k + 1u ? row.n[k] : cur;
                            wrong = cur != q;                                     // the row as it really stands does not hold what the speculation read
                            if (flag > 1) { row_promote(row, flag - 1u, q); pdirty = 1; }
                        } else {
                            row_promote(row, 4, q); pdirty = 1;                   // lion.rs:50-57
                        }
                        pdone = true;
                    }
                    LP_ADD(11, 1);
                    if (ballot64(!pdone) == 0) break;
                }
                LP_T(c7);
                uint32_t psf = ps;                                                    // the predictor slot my row is stored to
                bool plastf = plast;
                if (ballot64(wrong) != 0) {
                    // A speculation failed: a predicted quad read an entry that an earlier quad of this step has since moved.  Its real value — the entry of the
                    // row as forwarded — has another hash, so the quad behind it sits in another context than assumed, and so on.  Up to round 3 the whole record
                    // was decoded again by the scalar code on lane 0 (two dependent memory reads per quad: a fifth of the kernel's time on prose).  Now: ONE
                    // exact walk over the step's quads in stream order, in registers — every value wave-uniform — with rows forwarded between quads of one
                    // context, taken from the speculative gather where it was made at the right context, and read from memory only where the context turned
                    // out to be another one (lion.rs:85-186).
                    // (Round 6: the walk starts at the FIRST wrong lane — the lanes in front of it read what they should have, so their quads, contexts and
                    // resolved rows stand as they are and take part only as "the latest earlier quad of this context"; walked from lane 0 the repair was a
                    // fifth of the decoder's time.)
                    const uint32_t i0 = (uint32_t)__builtin_ctzll(ballot64(wrong));
                    LP_ADD(12, 1);
                    uint32_t ctx = i0 == 0 ? last_hash : rlane32(h, i0 - 1u);
                    uint32_t cxv = lane < i0 ? ps : 0xffffffffu, dirtyv = lane < i0 ? pdirty : 0u;   // per lane, once walked (or standing): my true context; my row differs from memory
                    Row5 rf = row_mem;
                    if (lane < i0) rf = row;
                    // (Later in round 6: the walk SKIPS the stretches that stand.  Where the walk arrives at a quad in the context the vector pass assumed for it, the
                    // quads from there on are as the vector pass left them up to the next one that read wrong or whose context — assumed — is one that a walked quad
                    // has been in, assumed or truly: their rows were forwarded among themselves and from quads that stand.)
                    bool tainted = false;
                    uint32_t i = i0;
#pragma nounroll
                    while (i < nact) {
                        if (ctx == rlane32(ps, i)) {
                            const uint64_t stop = ballot64(act && lane >= i && (tainted || wrong));
                            const uint32_t j = stop ? (uint32_t)__builtin_ctzll(stop) : nact;
                            if (j > i) {
                                if (lane >= i && lane < j) { cxv = ps; rf = row; dirtyv = pdirty; }
                                ctx = rlane32(h, j - 1u);
                                i = j;
                                continue;
                            }
                        }
                        LP_ADD(13, 1);
                        const uint64_t m = ballot64(lane < i && cxv == ctx);
                        Row5 r;
                        uint32_t dirty = 0;
                        if (m) {                                                      // the latest earlier quad of this context hands its row on
                            const uint32_t j = 63u - (uint32_t)__builtin_clzll(m);
#pragma unroll
                            for (int k = 0; k < 5; ++k) r.n[k] = rlane32(rf.n[k], j);
                            dirty = rlane32(dirtyv, j);
                        } else if (rlane32(ps, i) == ctx) {                           // nobody before it in this step: memory's row, gathered at the right place
#pragma unroll
                            for (int k = 0; k < 5; ++k) r.n[k] = rlane32(row_mem.n[k], i);
                        } else {
                            { u32x4 lo = {0u, 0u, 0u, 0u}; uint32_t hi = 0u; row_load_wide(t.pred + 5u * ctx, true, lo, hi); asm volatile("s_waitcnt vmcnt(0)" : "+v"(lo), "+v"(hi) : : "memory"); r = Row5{{lo.x, lo.y, lo.z, lo.w, hi}}; }
                        }
                        const uint32_t f = rlane32(flag, i);
                        uint32_t qi, hi;
                        if (f >= 1u && f <= 5u) {
                            qi = r.n[0];
#pragma unroll
                            for (uint32_t k = 1; k < 5; ++k) qi = f == k + 1u ? r.n[k] : qi;
                            hi = hash16(qi);
                            if (f > 1u) { row_promote(r, f - 1u, qi); dirty = 1; }
                        } else {
                            qi = rlane32(q, i); hi = rlane32(h, i);                   // what the dictionary gave: no context in it
                            row_promote(r, 4, qi); dirty = 1;
                        }
                        if (lane == i) { q = qi; h = hi; rf = r; cxv = ctx; dirtyv = dirty; }
                        tainted = tainted || ps == rlane32(ps, i) || ps == ctx;       // quad i was walked: what the vector pass made of its two contexts — the assumed and the true one — does not stand
                        ctx = hi;
                        ++i;
                    }
                    row = rf; pdirty = dirtyv; psf = cxv;
                    const uint64_t peq2 = same_key_mask64(cxv, act);
                    plastf = act && ((peq2 >> lane) >> 1) == 0;
                }
                LP_T(c8);
                if (plastf && pdirty) row_store_wide(t.pred + 5u * psf, row);
                if (dlast && ddirty) tbl_store_pair(t.dict + h, Pair{da, db});
                last_hash = rlane32(h, nact - 1u);
                LP_T(c9);
                LP_ADD(2, c3 - c2); LP_ADD(3, c4 - c3); LP_ADD(4, c5 - c4); LP_ADD(5, c6 - c5); LP_ADD(6, c7 - c6); LP_ADD(7, c8 - c7); LP_ADD(8, c9 - c8); LP_ADD(9, 1);
                if (act) st32u(dst + opos + 4u * lane, q);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __threadfence();
            if (mine_to_finish && lane == 0) {                                    // the rest: scalar code, codec.rs:102-123
                t.last_hash = last_hash;
                while (ipos < elen && !bad && !done) {
                    const uint64_t rem = elen - ipos;
                    if (guard.block_is_copy()) {
                        const uint32_t take = rem > G::kBlock ? G::kBlock : (uint32_t)rem;
                        if (opos + take > cap) { bad = true; break; }
                        for (uint32_t i = 0; i < take; ++i) dst[opos + i] = src[ipos + i];
                        ipos += take; opos += take;
                        if (rem <= G::kBlock) break;
                        guard.decay();
                        continue;
                    }
                    bad = lion_record_scalar(t, src, elen, ipos, dst, cap, opos, done, guard);
                }
                if (exact && !bad && opos != cap) bad = true;
                produced[chunk] = opos;
                if (bad) atomicOr(err, 1u);
            }
        }
        __threadfence();
    }
}

}  // namespace

uint64_t serial_table_bytes(int algo) { return 65536ull * (sizeof(Pair) + 4ull * (algo == DENSITY_HIP_LION ? 5 : 1)); }

hipError_t launch_serial_encode(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out,
                                uint64_t out_stride, uint64_t* d_sizes, uint8_t* d_tables, uint32_t n_slots, hipStream_t stream) {
    if (n_chunks == 0) return hipSuccess;
    const uint32_t blocks = (n_slots + 63) / 64;
    hipError_t e = hipMemsetAsync(d_tables, 0, (size_t)n_slots * serial_table_bytes(algo), stream);
    if (e != hipSuccess) return e;
    if (algo == DENSITY_HIP_CHEETAH && !g_force_lane_codec)
        hipLaunchKernelGGL(cheetah_encode_wave, dim3(n_slots), dim3(64), 0, stream, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots, (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 0u, (const uint32_t*)nullptr);
    else if (algo == DENSITY_HIP_CHEETAH)
        hipLaunchKernelGGL(serial_encode_chunks<DENSITY_HIP_CHEETAH>, dim3(blocks), dim3(64), 0, stream, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots);
    else if (!g_force_lane_codec)
        hipLaunchKernelGGL(lion_encode_wave, dim3(n_slots), dim3(64), 0, stream, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots,
                           (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 0u, (const uint32_t*)nullptr);
    else
        hipLaunchKernelGGL(serial_encode_chunks<DENSITY_HIP_LION>, dim3(blocks), dim3(64), 0, stream, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots);
    return hipGetLastError();
}

// the one-wave kernels in the service of exchange_stages.hip: the chunks d_only marks, whole (the wave clears its tables itself)
hipError_t launch_wave_encode_only(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                                   uint64_t* d_sizes, uint8_t* d_tables, uint32_t n_slots, const uint32_t* d_only, hipStream_t stream) {
    if (n_chunks == 0) return hipSuccess;
    auto kernel = algo == DENSITY_HIP_CHEETAH ? cheetah_encode_wave : lion_encode_wave;
    hipLaunchKernelGGL(kernel, dim3(n_slots), dim3(64), 0, stream, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots, d_only,
                       (uint32_t*)nullptr, 0u, 0u, (const uint32_t*)nullptr);
    return hipGetLastError();
}
hipError_t launch_wave_encode_heads(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                                    uint64_t* d_sizes, uint8_t* d_tables, uint32_t* d_head_state, uint32_t head_bytes, uint32_t head_calm, hipStream_t stream) {
    if (n_chunks == 0) return hipSuccess;
    auto kernel = algo == DENSITY_HIP_CHEETAH ? cheetah_encode_wave : lion_encode_wave;
    hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(64), 0, stream, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_chunks,
                       (const uint32_t*)nullptr, d_head_state, head_bytes, head_calm, (const uint32_t*)nullptr);
    return hipGetLastError();
}
hipError_t launch_wave_encode_tails(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                                    uint64_t* d_sizes, uint8_t* d_tables, const uint32_t* d_tail_state, hipStream_t stream) {
    if (n_chunks == 0) return hipSuccess;
    auto kernel = algo == DENSITY_HIP_CHEETAH ? cheetah_encode_wave : lion_encode_wave;
    hipLaunchKernelGGL(kernel, dim3(n_chunks), dim3(64), 0, stream, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_chunks,
                       (const uint32_t*)nullptr, (uint32_t*)nullptr, 0u, 0u, d_tail_state);
    return hipGetLastError();
}

hipError_t launch_serial_decode(int algo, const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks,
                                uint8_t* d_out, uint64_t out_stride, uint64_t out_total, bool exact, uint64_t* d_produced, uint32_t* d_err,
                                uint8_t* d_tables, uint32_t n_slots, hipStream_t stream) {
    if (n_chunks == 0) return hipSuccess;
    const uint32_t blocks = (n_slots + 63) / 64;
    hipError_t e = hipMemsetAsync(d_tables, 0, (size_t)n_slots * serial_table_bytes(algo), stream);
    if (e != hipSuccess) return e;
    if (algo == DENSITY_HIP_CHEETAH && !g_force_lane_codec)
        hipLaunchKernelGGL(cheetah_decode_wave, dim3(n_slots), dim3(64), 0, stream, d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact ? 1u : 0u, d_produced, d_err, d_tables, n_slots);
    else if (algo == DENSITY_HIP_CHEETAH)
        hipLaunchKernelGGL(serial_decode_chunks<DENSITY_HIP_CHEETAH>, dim3(blocks), dim3(64), 0, stream, d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact ? 1u : 0u, d_produced, d_err, d_tables, n_slots);
    else if (!g_force_lane_codec && !g_lion_one_wave)
        hipLaunchKernelGGL(lion_decode_pair, dim3(n_slots), dim3(128), 0, stream, d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact ? 1u : 0u, d_produced, d_err, d_tables, n_slots);
    else if (!g_force_lane_codec)
        hipLaunchKernelGGL(lion_decode_wave, dim3(n_slots), dim3(64), 0, stream, d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact ? 1u : 0u, d_produced, d_err, d_tables, n_slots);
    else
        hipLaunchKernelGGL(serial_decode_chunks<DENSITY_HIP_LION>, dim3(blocks), dim3(64), 0, stream, d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact ? 1u : 0u, d_produced, d_err, d_tables, n_slots);
    return hipGetLastError();
}

}  // namespace density

#ifdef LION_PHASES
extern "C" void density_debug_lion_phases(unsigned long long* out, int reset) {
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(density::g_lp), sizeof(unsigned long long) * 32);
    if (reset) { unsigned long long z[32] = {}; hipMemcpyToSymbol(HIP_SYMBOL(density::g_lp), z, sizeof z); }
}
#endif
