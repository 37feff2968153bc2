// decode_passes.hip — Cheetah container DECODING in passes (gfx950): everything that is parallel inside a chunk runs as ordered LDS
// exchange passes over the whole chunk, and what is not — one chain of dependent 16-bit look-ups per chunk — runs alone, in LDS.
//
// The reference decodes a chunk quad by quad through three tables (cheetah.rs:68-103,154-163).  Taken apart:
//
//   records    A stream has no framing: a record's length follows from its signature, raw-copy blocks from the blow-up protection over
//              the lengths before them (codec.rs:88-123, protection_state.rs).  `parse`: one wave per chunk chases the records through
//              an LDS window of the stream (one LDS round trip per record) and leaves every record's position and kind.
//   flags      `prepare`: one lane per quad reads its flag and item (io/read_signature.rs:9-15): PLAIN quads go straight to the output,
//              every quad gets a descriptor {slot, flag}.
//   dictionary chunk_map[h] = {a, b} is touched by the non-predicted quads only, at slots the STREAM names (the hash of a PLAIN quad, the
//              item of a MAP quad): nothing in it depends on a predicted quad.  PLAIN: (a, b) <- (q, a); MAP_A reads a; MAP_B reads b and
//              swaps (cheetah.rs:76-93).  With two cells X, Y per slot and an order bit o (a = o ? Y : X, b = o ? X : Y) this is:
//              PLAIN writes the cell b sits in and toggles o; MAP_B toggles o; MAP_A changes nothing.  So an ordered XOR per slot gives
//              every quad the o it meets, and then X and Y are plain last-writer cells: an ordered exchange per cell (mask 0 = read),
//              exactly what the encoder's stages do (exchange_stages.hip).  `dictionary` (round 6: one kernel, a quarter of the slots per
//              work-group, the taking-part quads of a trip packed into whole blocks first) does both; every MAP quad is then known,
//              without a single dependent look-up.
//   contexts   prediction_map[last_hash] is read by predicted quads and written by the others (:72,81,90,98).  Its SLOT is the hash of the
//              quad before — for a quad behind a predicted one the hash of a value that has to be looked up first: a chain of dependent
//              reads through the data being produced.  This is the only sequential part, and it needs hashes, not quads: H[c] = hash of
//              what follows context c is a table of 64 Ki x 16 bits = 128 KiB of LDS.  `walk`: per chunk a team of four waves takes 128
//              quads a turn — speculative reads of H ahead of the turn, ONE ordered pass over H and its verification under a token
//              (round 6; rounds 3-5: one wave, a dependent LDS round trip per predicted quad) — and leaves every quad's context.
//   values     With the contexts known the prediction table is one more ordered exchange pass: predicted quads read T[c], the others write
//              their quad (`values`); the answers are the predicted quads.
//
// The output buffer itself holds the quads between the passes (PLAIN from `prepare`, MAP from `cells`, predicted from `values`).
// Bit-exact for ANY input: tables start as the reference's zeroed tables and hold full 32-bit quads; a stream the reference would panic
// on (truncated, output too small) raises the error word.  Lion keeps its one-wave decoder: its five-deep prediction rows would need
// 640 KiB of hashes for the walk (lion.rs:29-48), four CUs' worth of LDS.
#include "common.hpp"
#include "kernels.hpp"

#include <cstdlib>
#include <type_traits>

namespace density {

extern __shared__ __attribute__((aligned(16))) uint8_t pass_lds[];
bool g_force_serial_decode = false;   // density_hip_set_kernel_variant(128): Cheetah containers on the one-wave decoder instead
bool g_serial_parse = false;          // density_hip_set_kernel_variant(1024): the records of a chunk found by the one-wave walk alone (no window kernels)
bool g_chain_walk = false;            // density_hip_set_kernel_variant(4096): the contexts walked run by run (round 5's walk) instead of 64 quads at a time
int g_walk_blocks = 2;                // 2: the walk by a team of four waves (default); 1 / 4: by ONE wave, 64 / 128 quads at a time (density_hip_set_kernel_variant bits 13-14: cross-checks)

namespace {

constexpr uint32_t kRecBytes = 128, kRecQuads = 32, kSigBytes = 8;        // cheetah.rs:188-196
constexpr uint32_t kRaw = 0x80000000u;                                    // rec[]: the block is a raw copy (codec.rs:89-91)
constexpr uint32_t kFlagPlain = 0, kFlagMapA = 1, kFlagPred = 3;                      // (2: MAP_B)   // cheetah.rs:17-23
// descriptor of a quad: slot [0,16) | flag [16,18) | takes no part (raw block, beyond the end) [19] | a MAP quad that read a never-written 0 [20]
constexpr uint32_t kDescNone = 1u << 19, kDescZero = 1u << 20;   // kDescZero: a MAP quad that read 0 (see the walk)
constexpr uint32_t kErrFormat = 1u, kErrWatchdog = 16u;
constexpr uint32_t kSpinLimit = 1u << 22, kPoison = 0xfffffffeu;

__device__ __forceinline__ uint32_t hash16(uint32_t q) { return (q * kHashMul) >> 16; }
// vec[lane] = val (both wave-uniform scalars): a compare and a select — v_writelane_b32 would want its lane select in M0 (an SGPR value and an SGPR
// lane select together break gfx9's one-scalar-operand rule), and M0 is the compiler's, not an asm statement's, to write
__device__ __forceinline__ uint32_t writelane(uint32_t vec, uint32_t val, uint32_t lane) {
    return lane_id() == lane ? val : vec;
}

// per chunk, left by `parse`: blocks, decoded bytes, quads and raw tail bytes of a ragged last record, where those bytes sit in the stream
struct ChunkInfo { uint32_t blocks, produced, last_quads, tail_bytes, tail_at, bad, ragged, pad1; };   // ragged: the last block is a partial record of last_quads quads + tail_bytes raw bytes

struct PassArgs {
    const uint8_t* in; const uint64_t* offsets; const uint64_t* sizes; uint32_t n_chunks;
    uint8_t* out; uint64_t out_stride, out_total;
    uint32_t* rec; uint32_t* desc; uint16_t* ctx; ChunkInfo* info; uint32_t* err;
};
__device__ __forceinline__ uint64_t chunk_cap(const PassArgs& a, uint64_t chunk) {
    const uint64_t room = a.out_total - chunk * a.out_stride;
    return room < a.out_stride ? room : a.out_stride;
}

// ---------------------------------------------------------------------------------------------------------------
// parse, in parallel (round 4): where the records of a CALM chunk stream start, without walking the stream from its first byte.
// A record's length is in its signature (8 + 128 - 2 per MAP flag - 4 per predicted one: cheetah.rs:17-23), so every even offset p is a candidate
// record start with a known successor, whether or not a record starts there; and as long as never two records in a row are incompressible
// (protection_state.rs:38-47) there are no raw copies and the FSM stays at rest.  The chain of real record starts enters each window of 2 KiB at one
// of 68 even offsets (a record is 8 .. 136 bytes long):
//   head      the one-wave walk below, with the real FSM, over the chunk's first records — a fresh dictionary makes them incompressible, raw copies
//             follow — until the FSM is at rest behind a record that was not incompressible: the windows count from there;
//   windows   per window, a lane per entry: from entry e the chain leaves the window at offset x of the next one after n records (or stops: fewer
//             than 136 bytes are left, the one-wave walk below takes over there), and whether it saw two incompressible records in a row;
//   walk      per chunk, the one-wave walk below with the windows' tables in LDS: wherever it stands at a window's entry with the FSM at rest, the
//             table says no pair lies ahead in this window and the record in front does not make one with its first, the window is taken in one step
//             (its entry and first block number noted); everything else — the windows with raw copies in them, the last records, the ragged end,
//             the checks — record by record as before;
//   emit      per window that was taken in one step, from its entry: rec[] of its blocks.
// (A chunk whose stream the tables have no room for, or that has no calm head, is walked whole.)  SURVEY.md §8: codec/codec.rs:82-126.
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kPW = 2048, kPE = 68, kPStop = 127, kPNone = 255, kHeadMark = 0x5eed0002u, kHeadMaxBlocks = 512;
constexpr uint32_t kPWinLds = kPW + 144;                                  // a window and what a record that starts in it may reach into the next
__device__ __forceinline__ uint32_t record_bytes(const uint8_t* lds_at) {  // item bytes behind the signature at a 2-byte aligned LDS address
    const uint16_t* h = reinterpret_cast<const uint16_t*>(lds_at);
    const uint32_t s0 = (uint32_t)h[0] | ((uint32_t)h[1] << 16), s1 = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
    const uint32_t lo0 = s0 & 0x55555555u, hi0 = (s0 >> 1) & 0x55555555u, lo1 = s1 & 0x55555555u, hi1 = (s1 >> 1) & 0x55555555u;
    return 4u * kRecQuads - 2u * (uint32_t)(__builtin_popcount(lo0 | hi0) + __builtin_popcount(lo1 | hi1)) - 2u * (uint32_t)(__builtin_popcount(lo0 & hi0) + __builtin_popcount(lo1 & hi1));
}
__device__ __forceinline__ void stage_window(uint8_t* win, const uint8_t* src, uint32_t wstart, uint32_t elen, uint32_t tid, uint32_t threads) {
    typedef uint4 uint4_u __attribute__((aligned(1)));
    for (uint32_t v = tid; v < kPWinLds / 16; v += threads) {
        const uint32_t p = wstart + 16u * v;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (p + 16u <= elen) { const uint4_u* sp = reinterpret_cast<const uint4_u*>(src + p); x = make_uint4(sp->x, sp->y, sp->z, sp->w); }
        else if (p < elen) {
            uint8_t t[16] = {};
            for (uint32_t i = 0; i < 16 && p + i < elen; ++i) t[i] = src[p + i];
            x = *reinterpret_cast<const uint4*>(t);
        }
        *reinterpret_cast<uint4*>(win + 16u * v) = x;
    }
    __syncthreads();
}
// T[(chunk * wpc + w) * kPE + e]: exit [0,7) (kPStop: the chain stopped inside the window) | records [7,17) | where it stopped, from the window's start [17,29)
// | first record incompressible [29] | last one [30] | two in a row [31]
__global__ __launch_bounds__(128) void cheetah_parse_windows(PassArgs a, uint32_t wpc, uint32_t* __restrict__ T, uint8_t* __restrict__ went) {
    __shared__ __attribute__((aligned(16))) uint8_t win[kPWinLds];
    const uint32_t e = threadIdx.x, w = blockIdx.x;                           // (two waves: the second one's four lanes take entries 64 .. 67)
    const uint64_t chunk = blockIdx.y;
    if (e == 0) went[chunk * wpc + w] = (uint8_t)kPNone;
    const uint64_t elen64 = a.sizes[chunk];
    if (elen64 >= 0x7fffff00ull) return;
    const ChunkInfo head = a.info[chunk];
    if (head.pad1 != kHeadMark) return;                                       // (no calm head: the one-wave walk takes the whole chunk)
    const uint32_t elen = (uint32_t)elen64;
    const uint64_t wstart64 = (uint64_t)head.produced + (uint64_t)w * kPW;    // windows count from where the head walk stopped
    if (wstart64 >= elen) return;
    const uint32_t wstart = (uint32_t)wstart64;
    const uint32_t hot_end = elen >= kSigBytes + kRecBytes ? elen - (kSigBytes + kRecBytes) + 1u : 0u;   // record starts below this have a whole record's room behind them (codec.rs:88)
    stage_window(win, a.in + a.offsets[chunk], wstart, elen, e, 128);
    if (e >= kPE) return;
    uint32_t pos = 2u * e, cnt = 0, first = 0, last = 0, pair = 0, stop = 0;
    while (pos < kPW) {
        if (wstart + pos >= hot_end) { stop = 1; break; }
        const uint32_t bytes = record_bytes(win + pos);
        const uint32_t inc = kSigBytes + bytes >= kRecBytes ? 1u : 0u;           // codec.rs:98
        if (cnt == 0) first = inc; else pair |= inc & last;
        last = inc; ++cnt;
        pos += kSigBytes + bytes;
    }
    T[(chunk * wpc + w) * kPE + e] = (stop ? kPStop : (pos - kPW) / 2u) | (cnt << 7) | ((stop ? pos : 0u) << 17) | (first << 29) | (last << 30) | (pair << 31);
}
__global__ __launch_bounds__(64) void cheetah_parse_emit(PassArgs a, uint32_t wpc, const uint8_t* __restrict__ went, const uint32_t* __restrict__ wbase) {
    __shared__ __attribute__((aligned(16))) uint8_t win[kPWinLds];
    const uint32_t lane = threadIdx.x, w = blockIdx.x;
    const uint64_t chunk = blockIdx.y;
    const uint32_t entry = went[chunk * wpc + w];
    if (entry == kPNone) return;                                              // (walked record by record, or not at all)
    const uint32_t elen = (uint32_t)a.sizes[chunk], wstart = a.info[chunk].pad1 + w * kPW;   // (pad1: where the chunk's windows count from, left by the walk)
    stage_window(win, a.in + a.offsets[chunk], wstart, elen, lane, 64);
    if (lane != 0) return;
    uint32_t* rec = a.rec + chunk * (a.out_stride / kRecBytes);
    uint32_t b = wbase[chunk * wpc + w], pos = 2u * entry;
    while (pos < kPW) {                                                       // (a window that is taken in one step holds whole records only, and its chain leaves it)
        rec[b++] = wstart + pos;
        pos += kSigBytes + record_bytes(win + pos);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// parse: Codec::decode's walk over the records (codec/codec.rs:82-126) without decoding them
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kWin = 8192, kWinStride = kWin - 64;                   // bytes of stream per LDS window; windows overlap by 64 bytes (a signature + slack)
// mode 0: the whole chunk | 1: from the head on with the windows' tables (the whole chunk where there are none) | 2: the head alone
__global__ __launch_bounds__(64) void cheetah_parse(PassArgs a, uint32_t mode, uint32_t wpc, uint32_t table_lds, const uint32_t* __restrict__ T, uint8_t* __restrict__ went,
                                                    uint32_t* __restrict__ wbase) {
    __shared__ __attribute__((aligned(16))) uint8_t win[2][kWin];
    const uint32_t lane = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const uint8_t* src = a.in + a.offsets[chunk];
    const uint64_t elen64 = a.sizes[chunk];
    const uint64_t cap = chunk_cap(a, chunk);
    uint32_t* rec = a.rec + chunk * (a.out_stride / kRecBytes);
    const uint32_t max_blocks = (uint32_t)((cap + kRecBytes - 1) / kRecBytes);
    ChunkInfo ci{};
    if (elen64 >= 0x7fffff00ull) { ci.bad = 1; }                           // 31-bit stream positions (a chunk stream: the launcher bounds it)
    const uint32_t elen = (uint32_t)elen64;
    Guard g;
    uint32_t ip = 0, op = 0, b = 0;
    const uint32_t misalign = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u);   // (windows are loaded from 16-byte aligned addresses)
    const uint8_t* base0 = src - misalign;
    const uint32_t have = elen + misalign;                                // bytes from base0 to the stream's end
    // window k = bytes [k * kWinStride, k * kWinStride + kWin) from base0, in win[k & 1]; window k + 1 is in flight (registers) while k is walked
    uint4 pf[kWin / 1024];
    auto fetch = [&](uint32_t k) {
#pragma unroll
        for (uint32_t j = 0; j < kWin / 1024; ++j) {
            const uint32_t i = k * kWinStride + (j * 64u + lane) * 16u;
            pf[j] = i < have ? *reinterpret_cast<const uint4*>(base0 + i) : make_uint4(0, 0, 0, 0);   // (at most 15 bytes past the end, inside its 16-byte line)
        }
    };
    auto land = [&](uint32_t k) {
#pragma unroll
        for (uint32_t j = 0; j < kWin / 1024; ++j) *reinterpret_cast<uint4*>(&win[k & 1u][(j * 64u + lane) * 16u]) = pf[j];
        __syncthreads();
    };
    uint32_t wk = 0;                                                      // the window being walked
    fetch(0); land(0); fetch(1);
    // record positions are collected 64 at a time in a register (lane b % 64) and stored together
    uint32_t recv = 0;
    bool tables = false;
    uint32_t p0 = 0, nwin = 0;
    const uint32_t* Ts = reinterpret_cast<const uint32_t*>(pass_lds);
    if (mode == 1 && !ci.bad) {                                           // the head walk and the window kernels have been over this chunk
        const ChunkInfo head = a.info[chunk];
        if (head.pad1 == kHeadMark && head.produced < elen) {
            p0 = head.produced;
            nwin = (elen - p0 + kPW - 1) / kPW;
            tables = nwin <= wpc && (uint64_t)nwin * kPE * 4 <= table_lds;
        }
        if (tables) {
            uint32_t* Tw = reinterpret_cast<uint32_t*>(pass_lds);
            for (uint32_t i = lane; i < nwin * kPE; i += 64) Tw[i] = T[chunk * wpc * kPE + i];
            __syncthreads();
            b = head.blocks; ip = p0; op = b * kRecBytes;
            g.counter = b;                                                // at rest (no penalty, start 1) behind a record that was not incompressible
            if ((b & 63u) != 0 && lane < (b & 63u)) recv = rec[(b & ~63u) + lane];   // the positions of the group that is being filled
        }
    }
    auto put = [&](uint32_t blk, uint32_t value) {
        recv = writelane(recv, value, blk & 63u);
        if ((blk & 63u) == 63u && (blk & ~63u) + lane < max_blocks) rec[(blk & ~63u) + lane] = recv;   // (never into the next chunk's positions)
    };
    auto window_for = [&](uint32_t pos) {                                  // the window that holds the 12 bytes at stream offset pos
        const uint32_t want = (pos + misalign) / kWinStride;
        if (want != wk) {
            if (want != wk + 1u) fetch(want);                             // (never: a record is far shorter than a window)
            land(want);
            wk = want;
            fetch(want + 1u);
        }
    };
    auto signature_at = [&](uint32_t pos) -> uint64_t {                   // 8 bytes at a 2-byte aligned stream offset, as a scalar
        const uint32_t at = pos + misalign - wk * kWinStride;
        const uint32_t* wd = reinterpret_cast<const uint32_t*>(&win[wk & 1u][at & ~3u]);
        const uint32_t d0 = rfl(wd[0]), d1 = rfl(wd[1]), d2 = rfl(wd[2]);
        return (at & 2u) ? (((uint64_t)((d1 >> 16) | (d2 << 16)) << 32) | ((d0 >> 16) | (d1 << 16))) : (((uint64_t)d1 << 32) | d0);
    };
    const uint32_t full_blocks = (uint32_t)(cap / kRecBytes);             // blocks of 128 decoded bytes the output has room for
    bool head_ok = false;
    while (!ci.bad && ip < elen) {
        if (mode == 2) {                                                  // the head: until the FSM is at rest behind a record that was not incompressible
            if (b >= 2 && g.penalty == 0 && g.start == 1 && g.prev == 0) { head_ok = true; break; }
            if (b >= kHeadMaxBlocks) break;
        }
        if (tables && g.penalty == 0 && g.start == 1 && ip >= p0) {       // at a window's entry with the FSM at rest: the whole window in one step?
            const uint32_t rel = ip - p0, w = rel / kPW, off = rel - w * kPW;
            if (off < 2u * kPE && w < nwin) {
                const uint32_t t = rfl(Ts[w * kPE + off / 2u]), cnt = (t >> 7) & 0x3ffu, ex = t & 127u;
                if (cnt != 0 && ex != kPStop && (t >> 31) == 0 && (g.prev & (t >> 29) & 1u) == 0 && b + cnt <= full_blocks) {
                    if ((b & 63u) != 0 && lane < (b & 63u)) rec[(b & ~63u) + lane] = recv;   // what has been collected of the group so far (`emit` fills in behind it)
                    if (lane == 0) { went[chunk * wpc + w] = (uint8_t)(off / 2u); wbase[chunk * wpc + w] = b; }
                    b += cnt; op += cnt * kRecBytes; g.counter += cnt;
                    ip = p0 + (w + 1u) * kPW + 2u * ex;
                    g.prev = (t >> 30) & 1u;
                    continue;
                }
            }
        }
        window_for(ip);
        if (mode == 0 || (mode == 1 && !tables)) {   // The common case in as few (scalar) instructions as it takes — a lone wave issues one instruction every 4-5 cycles: records that
            // are whole (codec.rs:88: 136 bytes are left), whose signature lies in this window, with the blow-up protection at rest (no
            // penalty, penalty start 1: protection_state.rs:19-27 then only counts) and room in the output.
            const uint32_t wend = (wk + 1u) * kWinStride - misalign;       // stream offsets below this have their 12 bytes in the window
            const uint32_t whole_end = elen >= kSigBytes + kRecBytes ? elen - (kSigBytes + kRecBytes) + 1u : 0u;
            const uint32_t hot_end = wend < whole_end ? wend : whole_end;
            while (ip < hot_end && g.penalty == 0 && g.start == 1 && b < full_blocks) {
                const uint64_t sig = signature_at(ip);
                const uint64_t lo = sig & 0x5555555555555555ull, hi = (sig >> 1) & 0x5555555555555555ull;
                // 4 bytes per PLAIN quad, 2 per MAP quad, none per predicted one (cheetah.rs:17-23): 128 - 2 * (flags with a bit set) - 2 * (flags with both)
                const uint32_t bytes = 4u * kRecQuads - 2u * (uint32_t)__builtin_popcountll(lo | hi) - 2u * (uint32_t)__builtin_popcountll(lo & hi);
                put(b, ip);
                ++b; ip += kSigBytes + bytes; op += kRecBytes;
                ++g.counter;
                const uint32_t inc = kSigBytes + bytes >= kRecBytes ? 1u : 0u;   // codec.rs:98
                if (inc & g.prev) g.penalty = g.start;                    // protection_state.rs:38-47
                g.prev = inc;
            }
            if (ip >= elen) break;
            window_for(ip);
        }
        const uint32_t left = elen - ip;
        const bool fast = left >= kSigBytes + kRecBytes;                  // codec.rs:88: a whole record is certainly there
        if (g.block_is_copy()) {                                         // codec.rs:89-91,103-110
            const uint32_t take = left > kRecBytes ? kRecBytes : left;
            if (b >= max_blocks || (uint64_t)op + take > cap) { ci.bad = 1; break; }
            put(b, ip | kRaw);
            ++b; ip += take; op += take;
            if (!fast && ip == elen) break;                               // :107-109: no decay behind the last raw block
            g.decay();
            continue;
        }
        if (left < kSigBytes) { ci.bad = 1; break; }                      // reference: read_u64_le panics
        const uint64_t sig = signature_at(ip);
        const uint64_t lo = sig & 0x5555555555555555ull, hi = (sig >> 1) & 0x5555555555555555ull;
        if (fast) {
            const uint32_t nplain = kRecQuads - (uint32_t)__builtin_popcountll(lo | hi), npred = (uint32_t)__builtin_popcountll(lo & hi);
            const uint32_t bytes = 4u * nplain + 2u * (kRecQuads - nplain - npred);
            if (b >= max_blocks || (uint64_t)op + kRecBytes > cap) { ci.bad = 1; break; }
            put(b, ip);
            ++b; ip += kSigBytes + bytes; op += kRecBytes;
            g.update(kSigBytes + bytes >= kRecBytes);                     // codec.rs:98
            continue;
        }
        // tail loop (codec.rs:102-123 with cheetah.rs:165-185): lane k looks at quad k of the record
        const uint32_t k = lane & 31u;
        const uint32_t f = (uint32_t)(sig >> (2u * k)) & 3u;
        const uint64_t below = (1ull << (2u * k)) - 1ull;
        const uint32_t pl_b = k - (uint32_t)__builtin_popcountll((lo | hi) & below), pr_b = (uint32_t)__builtin_popcountll(lo & hi & below);
        const uint32_t before = 4u * pl_b + 2u * (k - pl_b - pr_b);       // item bytes in front of mine
        const uint32_t rem = left - kSigBytes;
        const uint32_t mine = f == kFlagPlain ? 4u : f == kFlagPred ? 0u : 2u;
        const bool gone = before > rem;                                   // (cannot be reached without an earlier stop or error)
        const bool stop = !gone && f == kFlagPlain && rem - before < 4u;  // :169-176: implicit PLAIN at the end of the data
        const bool fail = !gone && !stop && rem - before < mine;          // reference: slice panic
        const uint64_t stops = ballot64(lane < 32 && (stop || gone)), fails = ballot64(lane < 32 && fail);
        const uint32_t ks = stops ? (uint32_t)__builtin_ctzll(stops) : 32u;
        if (fails && (uint32_t)__builtin_ctzll(fails) < ks) { ci.bad = 1; break; }
        if (ks < 32u) {
            const uint32_t at_stop = rfl(bperm(ks, before));
            const uint32_t tail = rem - at_stop;                          // 0..3 raw bytes
            if (ks == 0 && tail == 0) { ip = elen; break; }               // a signature with nothing behind it that yields a byte: the reference stops here, having written nothing (:169-170)
            if (b >= max_blocks || (uint64_t)op + 4u * ks + tail > cap) { ci.bad = 1; break; }
            put(b, ip);
            ++b;
            ci.ragged = 1; ci.last_quads = ks; ci.tail_bytes = tail; ci.tail_at = ip + kSigBytes + at_stop;
            op += 4u * ks + tail; ip = elen;
            break;
        }
        // a whole (short) record after all
        const uint32_t nplain = kRecQuads - (uint32_t)__builtin_popcountll(lo | hi), npred = (uint32_t)__builtin_popcountll(lo & hi);
        const uint32_t bytes = 4u * nplain + 2u * (kRecQuads - nplain - npred);
        if (b >= max_blocks || (uint64_t)op + kRecBytes > cap) { ci.bad = 1; break; }
        put(b, ip);
        ++b; ip += kSigBytes + bytes; op += kRecBytes;
        g.update(kSigBytes + bytes >= kRecBytes);
    }
    if ((b & 63u) != 0 && lane < (b & 63u)) rec[(b & ~63u) + lane] = recv;   // the positions not yet stored
    if (mode == 2) {                                                      // (what is wrong with a chunk is the last walk's to say)
        ChunkInfo h{};
        h.blocks = b; h.produced = ip; h.pad1 = head_ok ? kHeadMark : 0u;
        if (lane == 0) a.info[chunk] = h;
        return;
    }
    ci.blocks = b; ci.produced = op; ci.pad1 = p0;
    if (lane == 0) {
        a.info[chunk] = ci;
        if (ci.bad) atomicOr(a.err, kErrFormat);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// prepare: flags and items of every quad (one lane per quad, two records per wave)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cheetah_prepare(PassArgs a, uint32_t blocks_per_chunk) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t pair = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);   // 64 quads = two records
    const uint64_t pairs_per_chunk = blocks_per_chunk / 2;
    const uint64_t chunk = pair / pairs_per_chunk;
    if (chunk >= a.n_chunks) return;
    const uint32_t pb = (uint32_t)(pair % pairs_per_chunk) * 2u + (lane >> 5);   // my record's block in the chunk
    const uint32_t k = lane & 31u;
    const ChunkInfo ci = a.info[chunk];
    const uint64_t step = chunk * (a.out_stride / 4) + (uint64_t)pb * kRecQuads + k;
    uint32_t d = kDescNone;
    if (!ci.bad && pb < ci.blocks) {
        const uint8_t* src = a.in + a.offsets[chunk];
        uint8_t* dst = a.out + chunk * a.out_stride + (uint64_t)pb * kRecBytes;
        const uint32_t r = a.rec[chunk * (a.out_stride / kRecBytes) + pb];
        const bool last = pb + 1 == ci.blocks;
        const uint32_t block_len = last ? ci.produced - pb * kRecBytes : kRecBytes;   // decoded bytes of this block
        if (r & kRaw) {
            const uint8_t* p = src + (r & ~kRaw);
            if (4u * k + 4u <= block_len) st32u(dst + 4u * k, ld32u(p + 4u * k));
            else for (uint32_t i = 4u * k; i < block_len; ++i) dst[i] = p[i];
        } else {
            const uint8_t* p = src + r;
            const uint64_t sig = (uint64_t)ld32u(p) | ((uint64_t)ld32u(p + 4) << 32);
            const uint32_t nq = (last && ci.ragged) ? ci.last_quads : kRecQuads;
            if (k < nq) {
                const uint32_t f = (uint32_t)(sig >> (2u * k)) & 3u;
                const uint64_t lo = sig & 0x5555555555555555ull, hi = (sig >> 1) & 0x5555555555555555ull, below = (1ull << (2u * k)) - 1ull;
                const uint32_t pl_b = k - (uint32_t)__builtin_popcountll((lo | hi) & below), pr_b = (uint32_t)__builtin_popcountll(lo & hi & below);
                const uint8_t* item = p + kSigBytes + 4u * pl_b + 2u * (k - pl_b - pr_b);
                uint32_t slot = 0;
                if (f == kFlagPlain) { const uint32_t q = ld32u(item); slot = hash16(q); *reinterpret_cast<uint32_t*>(dst + 4u * k) = q; }   // cheetah.rs:69-70
                else if (f != kFlagPred) slot = ld16u(item);                                                                           // :79,86
                d = slot | (f << 16);
            }
            if (last && ci.ragged && k < ci.tail_bytes) dst[4u * nq + k] = src[ci.tail_at + k];   // cheetah.rs:171-174
        }
    }
    a.desc[step] = d;
}

// ---------------------------------------------------------------------------------------------------------------
// ordered passes: one table (a part of its slots) per work-group, the quads of the chunk in stream order behind LDS tokens
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kHalfSlots = 32768, kQuarterSlots = 16384, kTable = kHalfSlots * 4, kAhead = 16, kPassWaves = 8;
constexpr uint32_t kOrderBytes = kQuarterSlots / 8;                         // the order bits of a quarter of the slots
constexpr uint32_t kDenseCap = 384;                                         // quads of a trip (of 1024) that the dictionary pass packs into whole blocks: six blocks' worth
constexpr uint32_t kPassBase = kTable + kPassWaves * 64 * 4 + 16 + kOrderBytes;
constexpr uint32_t pass_lds_bytes(int pass) { return kPassBase + (pass == 1 ? kPassWaves * kDenseCap * 8u : 0u); }

#define DENSITY_PASS_X16(OP, ra, m, v, tokaddr, tokval)                                                                            \
    asm volatile(                                                                                                                 \
        OP " %0, %0, %16, %32\n\t" OP " %1, %1, %17, %33\n\t" OP " %2, %2, %18, %34\n\t" OP " %3, %3, %19, %35\n\t"                   \
        OP " %4, %4, %20, %36\n\t" OP " %5, %5, %21, %37\n\t" OP " %6, %6, %22, %38\n\t" OP " %7, %7, %23, %39\n\t"                   \
        OP " %8, %8, %24, %40\n\t" OP " %9, %9, %25, %41\n\t" OP " %10, %10, %26, %42\n\t" OP " %11, %11, %27, %43\n\t"               \
        OP " %12, %12, %28, %44\n\t" OP " %13, %13, %29, %45\n\t" OP " %14, %14, %30, %46\n\t" OP " %15, %15, %31, %47\n\t"           \
        "ds_write_b32 %48, %49\n\t"                                                                                               \
        "s_waitcnt lgkmcnt(0)"                                                                                                    \
        : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),                 \
          "+v"(ra[8]), "+v"(ra[9]), "+v"(ra[10]), "+v"(ra[11]), "+v"(ra[12]), "+v"(ra[13]), "+v"(ra[14]), "+v"(ra[15])            \
        : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]),                                 \
          "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11]), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]),                           \
          "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),                                 \
          "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]),                           \
          "v"(tokaddr), "v"(tokval)                                                                                               \
        : "memory")
// (the XOR form takes no mask: ds_xor_rtn_b32 vdst, addr, data)
#define DENSITY_PASS_XOR16(ra, v, tokaddr, tokval)                                                                                 \
    asm volatile(                                                                                                                 \
        "ds_xor_rtn_b32 %0, %0, %16\n\t" "ds_xor_rtn_b32 %1, %1, %17\n\t" "ds_xor_rtn_b32 %2, %2, %18\n\t" "ds_xor_rtn_b32 %3, %3, %19\n\t"         \
        "ds_xor_rtn_b32 %4, %4, %20\n\t" "ds_xor_rtn_b32 %5, %5, %21\n\t" "ds_xor_rtn_b32 %6, %6, %22\n\t" "ds_xor_rtn_b32 %7, %7, %23\n\t"         \
        "ds_xor_rtn_b32 %8, %8, %24\n\t" "ds_xor_rtn_b32 %9, %9, %25\n\t" "ds_xor_rtn_b32 %10, %10, %26\n\t" "ds_xor_rtn_b32 %11, %11, %27\n\t"     \
        "ds_xor_rtn_b32 %12, %12, %28\n\t" "ds_xor_rtn_b32 %13, %13, %29\n\t" "ds_xor_rtn_b32 %14, %14, %30\n\t" "ds_xor_rtn_b32 %15, %15, %31\n\t" \
        "ds_write_b32 %32, %33\n\t"                                                                                               \
        "s_waitcnt lgkmcnt(0)"                                                                                                    \
        : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),                 \
          "+v"(ra[8]), "+v"(ra[9]), "+v"(ra[10]), "+v"(ra[11]), "+v"(ra[12]), "+v"(ra[13]), "+v"(ra[14]), "+v"(ra[15])            \
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),                                 \
          "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]),                           \
          "v"(tokaddr), "v"(tokval)                                                                                               \
        : "memory")

// K ordered operations and the token behind them in ONE statement (the dense form of the dictionary pass: K = 3 or 6 whole blocks per trip)
template <uint32_t K>
__device__ __forceinline__ void lds_xor_token(uint32_t (&ra)[K], const uint32_t (&x)[K], uint32_t tokaddr, uint32_t tokval) {
    static_assert(K == 3 || K == 6, "three or six blocks");
    if constexpr (K == 3)
        asm volatile("ds_xor_rtn_b32 %0, %0, %3\n\tds_xor_rtn_b32 %1, %1, %4\n\tds_xor_rtn_b32 %2, %2, %5\n\tds_write_b32 %6, %7\n\ts_waitcnt lgkmcnt(0)"
                     : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]) : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(tokaddr), "v"(tokval) : "memory");
    else
        asm volatile("ds_xor_rtn_b32 %0, %0, %6\n\tds_xor_rtn_b32 %1, %1, %7\n\tds_xor_rtn_b32 %2, %2, %8\n\tds_xor_rtn_b32 %3, %3, %9\n\tds_xor_rtn_b32 %4, %4, %10\n\tds_xor_rtn_b32 %5, %5, %11\n\t"
                     "ds_write_b32 %12, %13\n\ts_waitcnt lgkmcnt(0)"
                     : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5])
                     : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(tokaddr), "v"(tokval) : "memory");
}
template <uint32_t K>
__device__ __forceinline__ void lds_mskor_token(uint32_t (&ra)[K], const uint32_t (&mk)[K], const uint32_t (&vl)[K], uint32_t tokaddr, uint32_t tokval) {
    static_assert(K == 3 || K == 6, "three or six blocks");
    if constexpr (K == 3)
        asm volatile("ds_mskor_rtn_b32 %0, %0, %3, %6\n\tds_mskor_rtn_b32 %1, %1, %4, %7\n\tds_mskor_rtn_b32 %2, %2, %5, %8\n\tds_write_b32 %9, %10\n\ts_waitcnt lgkmcnt(0)"
                     : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2])
                     : "v"(mk[0]), "v"(mk[1]), "v"(mk[2]), "v"(vl[0]), "v"(vl[1]), "v"(vl[2]), "v"(tokaddr), "v"(tokval) : "memory");
    else
        asm volatile("ds_mskor_rtn_b32 %0, %0, %6, %12\n\tds_mskor_rtn_b32 %1, %1, %7, %13\n\tds_mskor_rtn_b32 %2, %2, %8, %14\n\tds_mskor_rtn_b32 %3, %3, %9, %15\n\t"
                     "ds_mskor_rtn_b32 %4, %4, %10, %16\n\tds_mskor_rtn_b32 %5, %5, %11, %17\n\tds_write_b32 %18, %19\n\ts_waitcnt lgkmcnt(0)"
                     : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5])
                     : "v"(mk[0]), "v"(mk[1]), "v"(mk[2]), "v"(mk[3]), "v"(mk[4]), "v"(mk[5]), "v"(vl[0]), "v"(vl[1]), "v"(vl[2]), "v"(vl[3]), "v"(vl[4]), "v"(vl[5]),
                       "v"(tokaddr), "v"(tokval) : "memory");
}

// PASS 1 `dictionary` (round 6: `order` and `cells` in one kernel): a work-group owns a QUARTER of the slots, both cells X, Y of each (16 Ki x 2 dwords) and
//        their order bits (16 Ki bits).  Per trip two ordered groups of LDS operations, each behind a token of its own:
//        A  an ordered XOR on the slot's order bit — PLAIN and MAP_B toggle it, MAP_A reads it (cheetah.rs:71-72,87-89); what comes back is the o the quad meets;
//        B  an ordered exchange on the cell the quad touches — a sits in cell o, b in cell 1 - o: PLAIN writes its quad to b's cell, MAP_A / MAP_B read a's / b's.
//        (Up to round 5 two kernels: `order` — two work-groups per chunk on dword-wide order bits, o written back into the descriptors — and `cells` — four
//        per chunk, (cell, half of the slots) — every one of them streaming the chunk's descriptors: 0.17 + 0.42 ms per 100 MB.)
// PASS 2 `values`: slot = the quad's context (a half of them per work-group); predicted quads read, the others write their quad (cheetah.rs:72,81,90,98)
// The 8 waves take the trips of 16 blocks of 64 quads in rotation (exchange_stages.hip); a wave asks for its NEXT trip's descriptors, quads and contexts
// before it waits for its turn (round 6: until then a trip began with its loads, and a wave's iteration was two memory round trips long — 2,600 cycles per
// trip and work-group where the exchanges and the hand-off take 700).
template <int PASS>
__global__ __launch_bounds__(kPassWaves * 64) void cheetah_pass(PassArgs a) {
    static_assert(PASS == 1 || PASS == 2, "dictionary or values");
    constexpr uint32_t W = kPassWaves;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    constexpr uint32_t parts = PASS == 1 ? 4u : 2u;
    const uint64_t chunk = blockIdx.x / parts;
    const uint32_t mypart = blockIdx.x % parts;
    const ChunkInfo ci = a.info[chunk];
    if (ci.bad) return;
    const uint32_t nsteps = ci.blocks * kRecQuads;                                 // (quads of a ragged last record beyond its end carry kDescNone)
    const uint32_t trips = (nsteps + kAhead * 64u - 1u) / (kAhead * 64u);
    const uint64_t s0 = chunk * (a.out_stride / 4);
    uint32_t* w = reinterpret_cast<uint32_t*>(pass_lds);
    for (uint32_t k = threadIdx.x; k < kPassBase / 4u; k += W * 64u) w[k] = 0u;   // the reference's zeroed tables; sinks; the tokens; the order bits
    __syncthreads();
    uint32_t* __restrict__ desc = a.desc + s0;
    const uint16_t* __restrict__ ctx = a.ctx + s0;
    uint32_t* __restrict__ val = reinterpret_cast<uint32_t*>(a.out + chunk * a.out_stride);
    const uint32_t lds0 = lds_addr(pass_lds);
    const uint32_t sink = lds0 + kTable + threadIdx.x * 4u;
    const uint32_t token_a = lds0 + kTable + W * 256u, token_b = token_a + 4u;
    const uint32_t obits = token_a + 16u;
    const uint64_t cap = chunk_cap(a, chunk);
    const uint32_t limit = (uint32_t)((cap + 3) / 4), whole = (uint32_t)(cap / 4);   // dwords of this chunk's output that exist / that exist whole
    uint32_t nd[kAhead], nv[kAhead], nk[kAhead];
    auto fetch = [&](uint32_t t) {
#pragma unroll
        for (uint32_t j = 0; j < kAhead; ++j) {
            const uint32_t i = (t * kAhead + j) * 64u + lane;
            nd[j] = i < nsteps ? desc[i] : kDescNone;
            nv[j] = i < whole ? val[i] : 0u;                                       // (a writer's quad, left there by `prepare` / `cells`; anything for the others)
            nk[j] = (PASS == 2 && i < nsteps) ? ctx[i] : 0u;
        }
    };
    // my turn on a token: every earlier trip's operations of that group are queued
    auto await = [&](uint32_t token, uint32_t t) -> bool {
        for (uint32_t spins = 0;; ++spins) {
            uint32_t seen;
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(token) : "memory");
            seen = rfl(seen);
            if (seen == t) return true;
            if (seen == kPoison || spins > kSpinLimit) {
                if (seen != kPoison && lane == 0) { atomicOr(a.err, kErrWatchdog); w[(token_a - lds0) / 4u] = kPoison; w[(token_b - lds0) / 4u] = kPoison; }
                return false;
            }
        }
    };
    if (wave < trips) fetch(wave);
    // The dictionary pass in DENSE form (round 6): a quarter of the slots means one lane in six takes part in a block's operations, and an ordered LDS
    // instruction costs its 23+ cycles whatever its exec mask.  So a wave packs the taking-part quads of its trip, in stream order, into whole blocks first —
    // {slot | flag | place in the trip, quad} through a staging area of its own in LDS (ballot, count of the lanes below, one 8-byte write per block; a wave's LDS
    // operations execute in order: no barrier) — and both ordered groups run on 3 (up to 192 quads) or 6 (up to 384) blocks instead of 16; a trip with more
    // (one slot over and over) keeps the sixteen-block form below.  Results go back to the quads' places by the place each entry carries.
    auto dense_trip = [&](auto kc, uint32_t t, uint32_t cnt, uint32_t stg) -> bool {
        constexpr uint32_t K = decltype(kc)::value, kOff = 1u << 31;
        uint32_t lo[K], ra[K], x[K], m[K], v[K];
#pragma unroll
        for (uint32_t b = 0; b < K; ++b) {
            uint64_t e;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(e) : "v"(stg + (b * 64u + lane) * 8u) : "memory");
            const bool on = b * 64u + lane < cnt;
            lo[b] = on ? (uint32_t)e : kOff;                                  // (an entry's low word: slot [0,16) | flag [16,18) | place in the trip [18,28))
            v[b] = (uint32_t)(e >> 32);
            const uint32_t sq = lo[b] & (kQuarterSlots - 1u), f = (lo[b] >> 16) & 3u;
            ra[b] = on ? obits + (sq >> 5) * 4u : sink;
            x[b] = (on && f != kFlagMapA) ? 1u << (sq & 31u) : 0u;                 // toggles
        }
#pragma unroll
        for (uint32_t b = 0; b < K; ++b) asm volatile("" : "+v"(ra[b]), "+v"(x[b]));   // (made ahead of the wait, not behind it)
        if (!await(token_a, t)) return false;
        const uint32_t tokval = t + 1u;
        lds_xor_token<K>(ra, x, lane == 0 ? token_a : sink, tokval);
#pragma unroll
        for (uint32_t b = 0; b < K; ++b) {
            const bool on = !(lo[b] & kOff);
            const uint32_t sq = lo[b] & (kQuarterSlots - 1u), f = (lo[b] >> 16) & 3u, o = (ra[b] >> (sq & 31u)) & 1u;
            const uint32_t mycell = f == kFlagMapA ? o : 1u - o;
            const bool write = f == kFlagPlain;
            ra[b] = on ? lds0 + ((sq * 2u + mycell) * 4u) : sink;
            m[b] = (on && write) ? 0xffffffffu : 0u;
            v[b] = (on && write) ? v[b] : 0u;
        }
#pragma unroll
        for (uint32_t b = 0; b < K; ++b) asm volatile("" : "+v"(ra[b]), "+v"(m[b]), "+v"(v[b]));
        if (!await(token_b, t)) return false;
        lds_mskor_token<K>(ra, m, v, lane == 0 ? token_b : sink, tokval);
#pragma unroll
        for (uint32_t b = 0; b < K; ++b) {
            const bool rd = !(lo[b] & kOff) && ((lo[b] >> 16) & 3u) != kFlagPlain;
            const uint32_t i = t * (kAhead * 64u) + ((lo[b] >> 18) & 1023u);
            if (rd && i < limit) {
                val[i] = ra[b];
                if (ra[b] == 0u) desc[i] = (lo[b] & 0x3ffffu) | kDescZero;           // (see below)
            }
        }
        return true;
    };
    const uint32_t stage0 = lds0 + kPassBase + wave * (kDenseCap * 8u);
    // the sixteen-block form of a trip, from descriptors / quads / contexts in registers (the values pass; the dictionary pass's trips that do not pack)
    auto wide_trip = [&](uint32_t t, const uint32_t (&xd)[kAhead], const uint32_t (&xv)[kAhead], const uint32_t (&xk)[kAhead], bool fetch_next) -> bool {
        uint32_t ra[kAhead], m[kAhead], v[kAhead];
        uint32_t dd[kAhead], sh[kAhead];
        uint32_t minebits = 0, rdbits = 0;                                         // per lane, bit j: block j's quad is this work-group's / reads (in a vector register: 32 lane masks spill the scalar file)
#pragma unroll
        for (uint32_t j = 0; j < kAhead; ++j) {
            const uint32_t d = xd[j];
            dd[j] = d;
            const uint32_t f = (d >> 16) & 3u;
            const bool none = (d & kDescNone) != 0;
            if (PASS == 1) {
                const uint32_t key = d & 0xffffu, sq = key & (kQuarterSlots - 1u);
                const bool mine = !none && f != kFlagPred && (key >> 14) == mypart;
                minebits |= (mine ? 1u : 0u) << j;
                sh[j] = sq & 31u;
                ra[j] = mine ? obits + (sq >> 5) * 4u : sink;
                m[j] = (mine && f != kFlagMapA) ? 1u << sh[j] : 0u;                // toggles
                v[j] = xv[j];
            } else {
                const uint32_t key = xk[j];
                const bool write = f != kFlagPred;
                const bool mine = !none && (key >> 15) == mypart;
                ra[j] = mine ? lds0 + (key & (kHalfSlots - 1u)) * 4u : sink;
                m[j] = (mine && write) ? 0xffffffffu : 0u;
                v[j] = (mine && write) ? xv[j] : 0u;
                rdbits |= ((mine && !write) ? 1u : 0u) << j;
            }
        }
        if (fetch_next && t + W < trips) fetch(t + W);                             // in flight across the waits below
        // (the operands are made HERE, ahead of the wait: left to itself the compiler sinks their 230 instructions behind the poll loop — into the critical
        // section, where every later trip of the work-group waits for them)
#pragma unroll
        for (uint32_t j = 0; j < kAhead; ++j) asm volatile("" : "+v"(ra[j]), "+v"(m[j]), "+v"(v[j]));
        if (!await(token_a, t)) return false;
        if (PASS == 1) {
            const uint32_t tokaddr = lane == 0 ? token_a : sink, tokval = t + 1u;
            DENSITY_PASS_XOR16(ra, m, tokaddr, tokval);
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) {
                const uint32_t f = (dd[j] >> 16) & 3u, o = (ra[j] >> sh[j]) & 1u;
                // a sits in cell o, b in cell 1 - o (X = 0, Y = 1): PLAIN writes b's cell, MAP_A reads a's, MAP_B reads b's
                const uint32_t mycell = f == kFlagMapA ? o : 1u - o;
                const bool write = f == kFlagPlain, mine = (minebits >> j) & 1u;
                ra[j] = mine ? lds0 + (((dd[j] & (kQuarterSlots - 1u)) * 2u + mycell) * 4u) : sink;
                m[j] = (mine && write) ? 0xffffffffu : 0u;
                v[j] = (mine && write) ? v[j] : 0u;
                rdbits |= ((mine && !write) ? 1u : 0u) << j;
            }
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) asm volatile("" : "+v"(ra[j]), "+v"(m[j]), "+v"(v[j]));
            if (!await(token_b, t)) return false;
            const uint32_t tokaddr_b = lane == 0 ? token_b : sink;
            DENSITY_PASS_X16("ds_mskor_rtn_b32", ra, m, v, tokaddr_b, tokval);
        } else {
            const uint32_t tokaddr = lane == 0 ? token_a : sink, tokval = t + 1u;
            DENSITY_PASS_X16("ds_mskor_rtn_b32", ra, m, v, tokaddr, tokval);
        }
#pragma unroll
        for (uint32_t j = 0; j < kAhead; ++j) {
            const uint32_t i = (t * kAhead + j) * 64u + lane;
            if (((rdbits >> j) & 1u) && i < limit) {
                val[i] = ra[j];                                                    // cheetah.rs:80,87 / :96: the quad
                // A MAP quad's NEXT context is its item (cheetah.rs:78-83,85-92: the hash returned is the one read from the stream), but what a
                // later predicted quad in ITS context hashes to is the hash of the VALUE (:97-102).  The two agree whenever the cell holds a quad
                // some PLAIN quad put there — its slot is its hash — and differ only for a cell nothing has written yet: 0, whose hash is 0.  No
                // encoder produces that (a MAP of a never-written slot), a corrupt stream can: the walk needs to know (never taken otherwise).
                if (PASS == 1 && ra[j] == 0u) desc[i] = dd[j] | kDescZero;
            }
        }
        return true;
    };
    if (PASS == 2) {
        for (uint32_t t = wave; t < trips; t += W)
            if (!wide_trip(t, nd, nv, nk, true)) break;
        return;
    }
    // the dictionary pass: trip t packed in the staging area, trip t + W's descriptors and quads in flight in registers
    auto pack = [&]() -> uint32_t {
        uint32_t cnt = 0;
#pragma unroll
        for (uint32_t j = 0; j < kAhead; ++j) {
            const uint32_t d = nd[j], f = (d >> 16) & 3u;
            const bool mine = !(d & kDescNone) && f != kFlagPred && ((d & 0xffffu) >> 14) == mypart;
            const uint64_t bm = ballot64(mine);
            const uint32_t pos = cnt + mbcnt64(bm);
            if (mine && pos < kDenseCap) {
                const uint64_t e = (uint64_t)((d & 0x3ffffu) | ((j * 64u + lane) << 18)) | ((uint64_t)nv[j] << 32);
                asm volatile("ds_write_b64 %0, %1" ::"v"(stage0 + pos * 8u), "v"(e) : "memory");
            }
            cnt += (uint32_t)__builtin_popcountll(bm);
        }
        return cnt;
    };
    uint32_t cnt = 0;
    if (wave < trips) { cnt = pack(); if (wave + W < trips) fetch(wave + W); }
    for (uint32_t t = wave; t < trips; t += W) {
        bool ok;
        if (cnt <= 192u) ok = dense_trip(std::integral_constant<uint32_t, 3>{}, t, cnt, stage0);
        else if (cnt <= kDenseCap) ok = dense_trip(std::integral_constant<uint32_t, 6>{}, t, cnt, stage0);
        else {                                                                     // (rare: the trip's own descriptors and quads again, nothing in flight for it)
            uint32_t fd[kAhead], fv[kAhead], fk[kAhead];
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) {
                const uint32_t i = (t * kAhead + j) * 64u + lane;
                fd[j] = i < nsteps ? desc[i] : kDescNone;
                fv[j] = i < whole ? val[i] : 0u;
                fk[j] = 0u;
            }
            ok = wide_trip(t, fd, fv, fk, false);
        }
        if (!ok) break;
        if (t + W < trips) { cnt = pack(); if (t + 2u * W < trips) fetch(t + 2u * W); }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// walk: the contexts (cheetah.rs:97-102,161: last_hash) — the one chain of the decoder, on 16-bit hashes in LDS
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t kWalkTable = 65536u * 2u, kTileBlocks = 16, kTileBytes = kTileBlocks * 256u, kWalkLds = kWalkTable + 2u * kTileBytes;
// VEC (round 6): 64 quads at a time.  The chain is only as long as its DEPENDENT links: a predicted quad's context is the hash of the quad before it,
// which is in the descriptor unless that quad was predicted too — in 100 MB of prose four of five predicted quads follow a quad that was not.  So per
// block of 64 quads (a quad per lane):
//   speculate  the contexts of the lanes behind predicted quads by plain READS of H as it stands, level by level (a lane's level = the predicted lanes
//              right in front of it: one LDS round trip per level for all 64 lanes, where the run-by-run walk paid one per run and quad);
//   execute    ALL 64 table operations in one ordered instruction (ds_mskor_rtn_b32 on the 16-bit halves: gfx950 serves the lanes of one LDS
//              instruction in ascending order — §4.2 of DESIGN.md, verified at start-up —, a predicted lane reads, the others write H[context]);
//   verify     a predicted lane must have read in the ordered pass what its successors' contexts were derived from.  If every one did, the contexts
//              ARE the sequential ones (induction over the lanes: lane 0's context is the running one; if lanes 0..i hold the right contexts the
//              ordered pass did to H exactly what cheetah.rs:72,81,90,97-102 do up to quad i, so what lane i read is right, and with it lane i+1's
//              context).  If lane i0 is the first that read something else (a context written earlier in the SAME block: "the " twice within 256
//              bytes with two followers), lanes 0..i0 stand, the lanes behind it take their writes back — old halves, highest lane first: the lane-
//              reversed store of rotor.hip — and go again from what lane i0 really read.
// Blocks with a run of eight and more predicted quads (periodic input, zeros) keep the run-by-run chain below: a level costs what a link does.
// NB > 1: NB blocks (128 / 256 quads, NB registers per lane) go through speculate / execute / verify TOGETHER — the levels' reads of all of them are in
// flight at once, the ordered pass is NB instructions back to back (a wave's LDS instructions execute in issue order: block 0's lanes, then block 1's ...),
// so the LDS round trips, which are what a lone wave waits for, are shared by NB blocks; a wrong speculation costs one more pass over what lies behind it.
// lane-mask select: mask[lane] ? a : b with the mask in a scalar register pair (one VALU instruction; the compiler's own form of "(m >> lane) & 1" is three)
__device__ __forceinline__ uint32_t msel(uint64_t m, uint32_t ifset, uint32_t ifclear) {
    uint32_t r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(ifclear), "v"(ifset), "s"(m));
    return r;
}
// G LDS operations issued back to back and waited for inside ONE statement (an answer in flight lives in a register the compiler believes written:
// nothing but the wait may stand between issue and use)
template <uint32_t G>
__device__ __forceinline__ void lds_read_u16_group(uint32_t (&r)[G], const uint32_t (&addr)[G]) {
    static_assert(G == 1 || G == 2 || G == 4, "group of 1, 2 or 4 blocks");
    if constexpr (G == 1) asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r[0]) : "v"(addr[0]) : "memory");
    else if constexpr (G == 2) asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r[0]), "=&v"(r[1]) : "v"(addr[0]), "v"(addr[1]) : "memory");
    else asm volatile("ds_read_u16 %0, %4\n\tds_read_u16 %1, %5\n\tds_read_u16 %2, %6\n\tds_read_u16 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]) : "memory");
}
template <uint32_t G>
__device__ __forceinline__ void lds_mskor_group(uint32_t (&r)[G], const uint32_t (&addr)[G], const uint32_t (&mk)[G], const uint32_t (&vl)[G]) {
    if constexpr (G == 1) asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r[0]) : "v"(addr[0]), "v"(mk[0]), "v"(vl[0]) : "memory");
    else if constexpr (G == 2) asm volatile("ds_mskor_rtn_b32 %0, %2, %4, %6\n\tds_mskor_rtn_b32 %1, %3, %5, %7\n\ts_waitcnt lgkmcnt(0)"
                                            : "=&v"(r[0]), "=&v"(r[1]) : "v"(addr[0]), "v"(addr[1]), "v"(mk[0]), "v"(mk[1]), "v"(vl[0]), "v"(vl[1]) : "memory");
    else asm volatile("ds_mskor_rtn_b32 %0, %4, %8, %12\n\tds_mskor_rtn_b32 %1, %5, %9, %13\n\tds_mskor_rtn_b32 %2, %6, %10, %14\n\tds_mskor_rtn_b32 %3, %7, %11, %15\n\ts_waitcnt lgkmcnt(0)"
                      : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                      : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "v"(mk[0]), "v"(mk[1]), "v"(mk[2]), "v"(mk[3]), "v"(vl[0]), "v"(vl[1]), "v"(vl[2]), "v"(vl[3]) : "memory");
}
// One block of 64 quads that all take part, some of them predicted: the chain run by run, in as few instructions as it takes (the comments are at its
// use in cheetah_walk).  c: the running context in, out; returns every lane's context.
__device__ __forceinline__ uint32_t walk_chain_block(uint32_t lds0, uint32_t& c, uint32_t hprev, uint32_t h, uint32_t hw, uint64_t Pin) {
    const uint64_t P = ((uint64_t)rfl((uint32_t)(Pin >> 32)) << 32) | rfl((uint32_t)Pin);   // (wave-uniform by construction: said so to the register allocator)
    uint32_t av = lds0 + 2u * hprev;
    const uint32_t h2 = lds0 + 2u * h;                                      // what the running context becomes behind a quad that is not predicted
    uint64_t prem = P;
    uint32_t c2 = rfl(lds0 + 2u * c);
    uint32_t s_pos, s_p, s_r, v_t, v_u, s_m0;
    uint64_t s_m;
    asm volatile(
        "s_mov_b32 %[m0s], m0\n\t"                                            // (M0 is the compiler's: handed back as found)
        "s_mov_b32 %[pos], 0\n"
        "1:\n\t"                                                             // ---- next run of quads that are not predicted: [pos, p)
        "s_ff1_i32_b64 %[p], %[prem]\n\t"
        "s_min_u32 %[p], %[p], 64\n\t"                                       // (no predicted quad left: -1 -> 64)
        "s_sub_u32 %[r], %[p], %[pos]\n\t"
        "s_cmp_eq_u32 %[r], 0\n\t"
        "s_cbranch_scc1 2f\n\t"
        "s_bfm_b64 %[m], %[r], %[pos]\n\t"                                   // r bits from pos on (r < 64: some quad is predicted)
        "s_mov_b32 m0, %[pos]\n\t"
        "s_add_u32 %[r], %[p], -1\n\t"
        "v_writelane_b32 %[av], %[c2], m0\n\t"
        "s_mov_b64 exec, %[m]\n\t"
        "ds_write_b16 %[av], %[h]\n\t"
        "s_mov_b64 exec, -1\n\t"
        "v_readlane_b32 %[c2], %[h2], %[r]\n"
        "2:\n\t"
        "s_cmp_ge_u32 %[p], 64\n\t"
        "s_cbranch_scc1 4f\n\t"
        "v_mov_b32 %[t], %[c2]\n"
        // ---- a predicted quad at lane p: c <- H[c] (cheetah.rs:97-102).  Round 4: the chain is the read, one add and the branch — ~85 cycles
        // instead of ~105.  The address of H[c] (`t`, the same in every lane) goes into lane p's `av` by a select under a one-lane mask, the
        // bookkeeping and the test "is the next quad predicted too" are issued while the read is in flight (its answer lands in `u`, so `t`
        // stays readable), and the scalar copy of the context is taken once per run instead of once per quad.
        "3:\n\t"
        "ds_read_u16 %[u], %[t]\n\t"
        "s_bfm_b64 %[m], 1, %[p]\n\t"
        "s_bitset0_b64 %[prem], %[p]\n\t"
        "s_add_u32 %[p], %[p], 1\n\t"
        "v_cndmask_b32_e64 %[av], %[av], %[t], %[m]\n\t"
        "s_bitcmp1_b64 %[prem], %[p]\n\t"                                    // (p == 64 tests bit 0, which is clear by now: lane 0 was either not predicted or has been taken)
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshl_add_u32 %[t], %[u], 1, %[lds0]\n\t"
        "s_cbranch_scc1 3b\n\t"
        "s_nop 0\n\t"
        "v_readfirstlane_b32 %[c2], %[t]\n\t"
        "s_cmp_ge_u32 %[p], 64\n\t"
        "s_cbranch_scc1 4f\n\t"
        "s_mov_b32 %[pos], %[p]\n\t"
        "s_branch 1b\n"
        "4:\n\t"
        "s_mov_b32 m0, %[m0s]\n\t"
        : [av] "+v"(av), [c2] "+s"(c2), [prem] "+s"(prem), [pos] "=&s"(s_pos), [p] "=&s"(s_p), [r] "=&s"(s_r), [m] "=&s"(s_m), [t] "=&v"(v_t), [u] "=&v"(v_u), [m0s] "=&s"(s_m0)
        : [h] "v"(hw), [h2] "v"(h2), [lds0] "s"(lds0)
        : "memory", "scc");
    c = (c2 - lds0) >> 1;
    return (av - lds0) >> 1;
}

template <int NB>
__global__ __launch_bounds__(64) void cheetah_walk(PassArgs a) {
    constexpr bool VEC = NB >= 1;
    constexpr uint32_t G = NB > 1 ? NB : 1;
    const uint32_t lane = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const ChunkInfo ci = a.info[chunk];
    if (ci.bad) return;
    const uint32_t nsteps = ci.blocks * kRecQuads;
    const uint32_t nblk = (nsteps + 63u) / 64u;
    const uint64_t s0 = chunk * (a.out_stride / 4);
    const uint32_t* __restrict__ desc = a.desc + s0;
    uint16_t* __restrict__ ctx = a.ctx + s0;
    const uint32_t* __restrict__ val = reinterpret_cast<const uint32_t*>(a.out + chunk * a.out_stride);
    {   // H starts as the hash of the reference's zeroed prediction table: hash(0) = 0
        uint4* p = reinterpret_cast<uint4*>(pass_lds);
        for (uint32_t i = lane; i < kWalkTable / 16; i += 64) p[i] = make_uint4(0, 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const uint32_t lds0 = lds_addr(pass_lds);
    uint32_t* stage = reinterpret_cast<uint32_t*>(pass_lds + kWalkTable);           // two tiles of 16 blocks of descriptors
    uint32_t c = 0;                                                                // cheetah.rs:52: last_hash = 0
    // The chain below never waits for memory: the descriptors of tile t + 1 are in flight (registers) while tile t is walked out of LDS
    // (one wave has one load's latency — microseconds — per request: with a block per request the walk ran at the speed of its loads).
    uint32_t tile[kTileBlocks];
    auto fetch = [&](uint32_t t) {
#pragma unroll
        for (uint32_t j = 0; j < kTileBlocks; ++j) {
            const uint32_t i = (t * kTileBlocks + j) * 64u + lane;
            tile[j] = i < nsteps ? desc[i] : kDescNone;
        }
    };
    auto land = [&](uint32_t t) {
#pragma unroll
        for (uint32_t j = 0; j < kTileBlocks; ++j) stage[(t & 1u) * (kTileBytes / 4) + j * 64u + lane] = tile[j];
    };
    fetch(0); land(0);
    // a MAP quad's hash — the next quad's context — is its item, a PLAIN quad's the hash of its value (in the descriptor either way); a predicted
    // quad's comes out of H, which holds per context the hash of the VALUE last left there (`hw`)
    for (uint32_t blk = 0; blk < nblk; ++blk) {
        const uint32_t t = blk / kTileBlocks, j = blk % kTileBlocks;
        if (j == 0) fetch(t + 1u);
        if (NB > 1 && j % G == 0 && blk + G <= nblk) {
            // ---- a group of G blocks, if every one of them is whole and has no long run ----
            uint32_t dv[G], hv[G], hwv[G], cvv[G], rsv[G], rfv[G], r2v[G], shv[G];
            uint64_t Pm[G], Nm[G], K0m[G], known[G], fin[G], rdone[G];
            bool ok = true;
#pragma unroll
            for (uint32_t b = 0; b < G; ++b) {
                dv[b] = stage[(t & 1u) * (kTileBytes / 4) + (j + b) * 64u + lane];
                hv[b] = dv[b] & 0xffffu;
                hwv[b] = (dv[b] & kDescZero) ? 0u : hv[b];
                const bool none = (dv[b] & kDescNone) != 0, pred = ((dv[b] >> 16) & 3u) == kFlagPred;
                Pm[b] = ballot64(!none && pred); Nm[b] = ballot64(!none && !pred);
                uint64_t lr = Pm[b] & (Pm[b] >> 1); lr &= lr >> 2; lr &= lr >> 4;
                ok = ok && (Pm[b] | Nm[b]) == ~0ull && lr == 0;
            }
            if (__builtin_expect(ok, 1)) {
                if (j + G == kTileBlocks || blk + G == nblk) land(t + 1u);
#pragma unroll
                for (uint32_t b = 0; b < G; ++b) {
                    const uint32_t hp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hv[b], 0x138, 0xf, 0xf, false);   // wave_shr:1
                    const uint32_t first = b == 0 ? c : (uint32_t)__builtin_amdgcn_readlane((int)hv[b ? b - 1 : 0], 63);   // (meaningful only where the quad before was not predicted)
                    cvv[b] = lane == 0 ? first : hp;
                    K0m[b] = (Nm[b] << 1) | (b == 0 ? 1ull : (Nm[b ? b - 1 : 0] >> 63));
                    known[b] = K0m[b]; fin[b] = 0; rsv[b] = 0; rfv[b] = 0;
                }
                for (;;) {
                    // speculate: every level's reads of all G blocks in flight together
#pragma unroll
                    for (uint32_t b = 0; b < G; ++b) rdone[b] = fin[b];
                    for (;;) {
                        uint64_t R[G], any = 0;
#pragma unroll
                        for (uint32_t b = 0; b < G; ++b) { R[b] = Pm[b] & known[b] & ~rdone[b]; any |= R[b]; }
                        if (!any) break;
                        uint32_t r[G], ad[G];
#pragma unroll
                        for (uint32_t b = 0; b < G; ++b) ad[b] = lds0 + 2u * cvv[b];
                        lds_read_u16_group<G>(r, ad);                                  // (every block reads, whether or not one of its lanes needs it: a stale context is a valid address)
#pragma unroll
                        for (uint32_t b = 0; b < G; ++b) {
                            const uint64_t in = (R[b] << 1) | (b == 0 ? 0ull : (R[b ? b - 1 : 0] >> 63));   // the lanes that learn their context this round
                            if (R[b]) rsv[b] = msel(R[b], r[b], rsv[b]);
                            if (in) {
                                uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r[b], 0x138, 0xf, 0xf, false);   // wave_shr:1
                                if (b != 0 && (in & 1ull)) { const uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)r[b ? b - 1 : 0], 63); up = lane == 0 ? carry : up; }
                                cvv[b] = msel(in, up, cvv[b]);
                            }
                            known[b] |= in; rdone[b] |= R[b];
                        }
                    }
                    // execute: the lanes that do not stand yet, block after block, each in lane order
                    uint32_t ret[G], xa[G], xm[G], xv[G];
#pragma unroll
                    for (uint32_t b = 0; b < G; ++b) {
                        shv[b] = (cvv[b] & 1u) * 16u;
                        const uint64_t w = Nm[b] & ~fin[b];
                        xa[b] = lds0 + ((2u * cvv[b]) & ~3u);
                        xm[b] = msel(w, 0xffffu << shv[b], 0u); xv[b] = msel(w, hwv[b] << shv[b], 0u);
                    }
                    lds_mskor_group<G>(ret, xa, xm, xv);
                    // verify
                    uint64_t bad[G], anybad = 0;
#pragma unroll
                    for (uint32_t b = 0; b < G; ++b) {
                        r2v[b] = (ret[b] >> shv[b]) & 0xffffu;
                        bad[b] = ballot64(r2v[b] != rsv[b]) & Pm[b] & ~fin[b];
                        anybad |= bad[b];
                    }
                    if (__builtin_expect(anybad == 0, 1)) {
#pragma unroll
                        for (uint32_t b = 0; b < G; ++b) rfv[b] = msel(~fin[b], r2v[b], rfv[b]);
                        break;
                    }
                    uint32_t b0 = 0;
#pragma unroll
                    for (uint32_t b = G; b-- > 0;) if (bad[b]) b0 = b;
                    uint64_t badb = bad[0];
#pragma unroll
                    for (uint32_t b = 1; b < G; ++b) badb = b0 == b ? bad[b] : badb;
                    const uint32_t i0 = (uint32_t)__builtin_ctzll(badb);
                    const uint64_t upto = (2ull << i0) - 1ull;                           // lanes 0 .. i0 of block b0 (i0 == 63: all of them)
                    uint64_t stands[G];
#pragma unroll
                    for (uint32_t b = 0; b < G; ++b) stands[b] = b < b0 ? ~0ull : b == b0 ? upto : 0ull;
                    // the writes behind the first wrong read are taken back: old halves, the latest write first (blocks from the last to b0, lanes reversed)
#pragma unroll
                    for (uint32_t b = G; b-- > 0;) {
                        const uint64_t undo = Nm[b] & ~stands[b] & ~fin[b];
                        if (undo) {
                            const uint32_t ar = bperm(63u - lane, lds0 + 2u * cvv[b]), old = bperm(63u - lane, r2v[b]);
                            if ((undo >> (63u - lane)) & 1ull) asm volatile("ds_write_b16 %0, %1" ::"v"(ar), "v"(old) : "memory");
                        }
                    }
                    const uint32_t truth = (uint32_t)__builtin_amdgcn_readlane((int)(b0 == 0 ? r2v[0] : b0 == 1 ? r2v[G > 1 ? 1 : 0] : b0 == 2 ? r2v[G > 2 ? 2 : 0] : r2v[G > 3 ? 3 : 0]), (int)i0);
                    bool all = true;
#pragma unroll
                    for (uint32_t b = 0; b < G; ++b) {
                        rfv[b] = msel(stands[b] & ~fin[b], r2v[b], rfv[b]);
                        fin[b] = stands[b];
                        all = all && fin[b] == ~0ull;
                        // the lane behind (b0, i0) now knows its context; everything else behind it is as unknown as before the first round
                        const uint64_t next = b == b0 ? (i0 == 63u ? 0ull : (2ull << i0) & ~upto) : (b == b0 + 1u && i0 == 63u ? 1ull : 0ull);
                        if (next) cvv[b] = msel(next, truth, cvv[b]);
                        known[b] = stands[b] | next | K0m[b];
                    }
                    if (all) break;
                }
                const uint32_t last = msel(Pm[G - 1], rfv[G - 1], hv[G - 1]);
                c = (uint32_t)__builtin_amdgcn_readlane((int)last, 63);
#pragma unroll
                for (uint32_t b = 0; b < G; ++b) { const uint32_t i = (blk + b) * 64u + lane; if (i < nsteps) ctx[i] = (uint16_t)cvv[b]; }
                blk += G - 1u;
                continue;
            }
        }
        const uint32_t d = stage[(t & 1u) * (kTileBytes / 4) + j * 64u + lane];
        if (j == kTileBlocks - 1u || blk + 1u == nblk) land(t + 1u);                 // (behind the read of the tile's last block: the other buffer)
        const uint32_t h = d & 0xffffu;
        const uint32_t hw = (d & kDescZero) ? 0u : h;                              // what H takes for this quad: the hash of its VALUE (kDescZero: a MAP quad that read a never-written 0)
        const bool none = (d & kDescNone) != 0, pred = ((d >> 16) & 3u) == kFlagPred;
        const uint64_t P = ballot64(!none && pred), N = ballot64(!none && !pred);
        // what the quad before me hashed to: my context if that quad was not predicted (lane 0: the running context)
        const uint32_t hprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x138, 0xf, 0xf, false);   // wave_shr:1
        uint32_t cv = hprev;                                                       // my context: the hash of the quad before me, unless patched below
        const uint64_t active = P | N;
        uint64_t longrun = P & (P >> 1); longrun &= longrun >> 2; longrun &= longrun >> 4;   // bit i: lanes i .. i+7 are all predicted
        if (VEC && __builtin_expect(active == ~0ull && longrun == 0, 1)) {
            const uint64_t K0 = (N << 1) | 1ull;                                     // lanes whose context is in the descriptors (lane 0: the running context)
            cv = lane == 0 ? c : hprev;
            const bool lp = (P >> lane) & 1ull;
            uint64_t fin = 0, known = K0;
            uint32_t rfin = 0;                                                       // predicted lanes: what they read (their successor's context)
            for (;;) {
                // speculate
                uint32_t rs = 0;
                uint64_t rdone = fin;
                for (;;) {
                    const uint64_t R = P & known & ~rdone;
                    if (!R) break;
                    uint32_t r;
                    asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(lds0 + 2u * cv) : "memory");
                    const uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r, 0x138, 0xf, 0xf, false);   // wave_shr:1
                    if ((R >> lane) & 1ull) rs = r;
                    if (((R << 1) >> lane) & 1ull) cv = up;
                    known |= R << 1; rdone |= R;
                }
                // execute: every lane that does not stand yet, in stream order
                const bool pend = !((fin >> lane) & 1ull), wr = pend && !lp;
                const uint32_t sh = (cv & 1u) * 16u;
                uint32_t ret;
                asm volatile("ds_mskor_rtn_b32 %0, %1, %2, %3\n\ts_waitcnt lgkmcnt(0)" : "=v"(ret) : "v"(lds0 + ((2u * cv) & ~3u)), "v"(wr ? 0xffffu << sh : 0u), "v"(wr ? hw << sh : 0u) : "memory");
                const uint32_t r2 = (ret >> sh) & 0xffffu;
                // verify
                const uint64_t bad = ballot64(pend && lp && r2 != rs);
                if (__builtin_expect(bad == 0, 1)) { if (pend) rfin = r2; break; }
                const uint32_t i0 = (uint32_t)__builtin_ctzll(bad);
                const uint64_t stands = (2ull << i0) - 1ull;                           // lanes 0 .. i0 (i0 == 63: all of them)
                const uint64_t undo = N & ~stands;                                   // the writes behind i0 (every pending lane beyond i0 that is not predicted wrote)
                if (undo) {
                    const uint32_t ar = bperm(63u - lane, lds0 + 2u * cv), old = bperm(63u - lane, r2);
                    if ((undo >> (63u - lane)) & 1ull) asm volatile("ds_write_b16 %0, %1" ::"v"(ar), "v"(old) : "memory");
                }
                if (pend && ((stands >> lane) & 1ull)) rfin = r2;
                fin = stands;
                if (fin == ~0ull) break;
                const uint32_t truth = (uint32_t)__builtin_amdgcn_readlane((int)r2, (int)i0);
                if (lane == i0 + 1u) cv = truth;
                known = stands | (stands << 1) | K0;
            }
            const uint32_t last = lp ? rfin : h;
            c = (uint32_t)__builtin_amdgcn_readlane((int)last, 63);
        } else
        if (__builtin_expect(active == ~0ull && P != 0 && P != ~0ull, 1)) {
            // Every quad of the block takes part (no raw-copy block, not the chunk's end) and some, not all, are predicted: the chain in as
            // few instructions as it takes — a lone wave issues one instruction every 4-5 cycles, so the instruction count IS the walk's
            // time (the compiled form of the loop below spent ~110 instructions per run, this one ~29).  Per run of quads that are not
            // predicted ONE ordered 16-bit store under an exec mask (each writes H[its context] = its hash, cheetah.rs:72,81,90; the first
            // one's context is the running one, patched into its lane), per predicted quad one LDS round trip (:97-102).
            // (v_writelane takes its lane from M0: an SGPR value and an SGPR lane select in one instruction break gfx9's one-scalar rule)
            // State in the loop: `c2` = LDS address of H[running context]; `av` = per lane the LDS address of H[its context] (2 * hash of the quad
            // before it, patched where the context is the running one) — the store's address operand and, shifted back, the context to report.
            uint32_t av = lds0 + 2u * hprev;
            const uint32_t h2 = lds0 + 2u * h;                                      // what the running context becomes behind a quad that is not predicted
            uint64_t prem = P;
            uint32_t c2 = lds0 + 2u * c;
            uint32_t s_pos, s_p, s_r, v_t, v_u, s_m0;
            uint64_t s_m;
            asm volatile(
                "s_mov_b32 %[m0s], m0\n\t"                                            // (M0 is the compiler's: handed back as found)
                "s_mov_b32 %[pos], 0\n"
                "1:\n\t"                                                             // ---- next run of quads that are not predicted: [pos, p)
                "s_ff1_i32_b64 %[p], %[prem]\n\t"
                "s_min_u32 %[p], %[p], 64\n\t"                                       // (no predicted quad left: -1 -> 64)
                "s_sub_u32 %[r], %[p], %[pos]\n\t"
                "s_cmp_eq_u32 %[r], 0\n\t"
                "s_cbranch_scc1 2f\n\t"
                "s_bfm_b64 %[m], %[r], %[pos]\n\t"                                   // r bits from pos on (r < 64: some quad is predicted)
                "s_mov_b32 m0, %[pos]\n\t"
                "s_add_u32 %[r], %[p], -1\n\t"
                "v_writelane_b32 %[av], %[c2], m0\n\t"
                "s_mov_b64 exec, %[m]\n\t"
                "ds_write_b16 %[av], %[h]\n\t"
                "s_mov_b64 exec, -1\n\t"
                "v_readlane_b32 %[c2], %[h2], %[r]\n"
                "2:\n\t"
                "s_cmp_ge_u32 %[p], 64\n\t"
                "s_cbranch_scc1 4f\n\t"
                "v_mov_b32 %[t], %[c2]\n"
                // ---- a predicted quad at lane p: c <- H[c] (cheetah.rs:97-102).  Round 4: the chain is the read, one add and the branch — ~85 cycles
                // instead of ~105.  The address of H[c] (`t`, the same in every lane) goes into lane p's `av` by a select under a one-lane mask, the
                // bookkeeping and the test "is the next quad predicted too" are issued while the read is in flight (its answer lands in `u`, so `t`
                // stays readable), and the scalar copy of the context is taken once per run instead of once per quad.
                "3:\n\t"
                "ds_read_u16 %[u], %[t]\n\t"
                "s_bfm_b64 %[m], 1, %[p]\n\t"
                "s_bitset0_b64 %[prem], %[p]\n\t"
                "s_add_u32 %[p], %[p], 1\n\t"
                "v_cndmask_b32_e64 %[av], %[av], %[t], %[m]\n\t"
                "s_bitcmp1_b64 %[prem], %[p]\n\t"                                    // (p == 64 tests bit 0, which is clear by now: lane 0 was either not predicted or has been taken)
                "s_waitcnt lgkmcnt(0)\n\t"
                "v_lshl_add_u32 %[t], %[u], 1, %[lds0]\n\t"
                "s_cbranch_scc1 3b\n\t"
                "s_nop 0\n\t"
                "v_readfirstlane_b32 %[c2], %[t]\n\t"
                "s_cmp_ge_u32 %[p], 64\n\t"
                "s_cbranch_scc1 4f\n\t"
                "s_mov_b32 %[pos], %[p]\n\t"
                "s_branch 1b\n"
                "4:\n\t"
                "s_mov_b32 m0, %[m0s]\n\t"
                : [av] "+v"(av), [c2] "+s"(c2), [prem] "+s"(prem), [pos] "=&s"(s_pos), [p] "=&s"(s_p), [r] "=&s"(s_r), [m] "=&s"(s_m), [t] "=&v"(v_t), [u] "=&v"(v_u), [m0s] "=&s"(s_m0)
                : [h] "v"(hw), [h2] "v"(h2), [lds0] "s"(lds0)
                : "memory", "scc");
            c = (c2 - lds0) >> 1;
            cv = (av - lds0) >> 1;
        } else {
            // a block with quads that take no part (a raw-copy block's, the chunk's end): the same, run by run, stepping over them — the
            // context passes through (codec.rs:89-91: a raw block touches no state)
            cv = 0;
            uint32_t pos = 0;
            while (pos < 64u) {
                const uint64_t rest = active >> pos;
                if (!rest) break;
                pos += (uint32_t)__builtin_ctzll(rest);
                if ((N >> pos) & 1ull) {
                    const uint64_t inv = ~(N >> pos);
                    const uint32_t r = inv ? (uint32_t)__builtin_ctzll(inv) : 64u - pos;
                    const bool in = lane >= pos && lane < pos + r;
                    const uint32_t mine = lane == pos ? c : hprev;
                    if (in) {
                        cv = mine;
                        asm volatile("ds_write_b16 %0, %1" ::"v"(lds0 + 2u * mine), "v"(hw) : "memory");
                    }
                    c = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)(pos + r - 1u));
                    pos += r;
                } else {
                    const uint64_t inv = ~(P >> pos);
                    const uint32_t r = inv ? (uint32_t)__builtin_ctzll(inv) : 64u - pos;
                    for (uint32_t t = 0; t < r; ++t) {
                        if (lane == pos + t) cv = c;
                        uint32_t nx;
                        asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nx) : "v"(lds0 + 2u * c) : "memory");
                        nx = rfl(nx);
                        if (nx == c) {                                             // a fixed point: the table does not change inside a run
                            if (lane > pos + t && lane < pos + r) cv = c;
                            break;
                        }
                        c = nx;
                    }
                    pos += r;
                }
            }
        }
        const uint32_t i = blk * 64u + lane;
        if (i < nsteps) ctx[i] = (uint16_t)cv;
    }
    (void)val;
}

// ---------------------------------------------------------------------------------------------------------------
// walk, by a TEAM of waves (round 6; the default).  One wave spends its time ISSUING: ~380 instructions per 128 quads at one instruction per five cycles —
// classification, the speculative reads level by level, the bookkeeping — and only ~500 cycles of those 2,500 in what must happen in stream order (the ordered
// pass over H and its verification).  So kTeam waves of one work-group share the chunk's H: wave w takes the groups g = w (mod kTeam) of 128 quads, does
// everything that needs no order AHEAD of its turn — its descriptors come straight from memory into registers two turns ahead, the speculative reads see H
// as it stands, groups of other waves not yet applied: speculation may be as stale as it likes, the verification below does not care how a context was
// guessed — and then, holding the token (an LDS word: the group whose turn it is; the running context travels beside it):
//   patch      lane 0's context, if the quad before the group was predicted (its hash is the predecessor's to tell);
//   execute    the ordered pass (as above), verify, take back and go again from the first wrong read until every lane stands;
//   hand on    the running context and the token; the contexts are stored behind that.
// Groups the 128-at-a-time form does not take (a raw-copy block, the chunk's end, a run of eight predicted quads) are walked block by block under the token
// by the run-by-run code.  A wave's LDS operations execute in issue order and the token is written behind them: whoever sees it sees H after them (§4.2).
// ---------------------------------------------------------------------------------------------------------------
// (geometry, as compile-time switches for same-box A/B builds — tools/build_variant.sh, tools/gpu_walk_ab.py: waves of a team, blocks of 64 quads a turn, how
// many turns ahead of its own a wave starts its speculative reads.  Measured on config 3, decode ms: 2 x 4 waves, one turn ahead 1.207; 1 x 4 1.205; 2 x 3 1.208;
// 2 x 2 1.29; 4 x 4 1.39; 2 x 6 1.23; 2 x 4 reading as early as it can — three turns ahead — 1.33)
#ifndef DENSITY_WALK_TEAM
#define DENSITY_WALK_TEAM 4
#endif
#ifndef DENSITY_WALK_G
#define DENSITY_WALK_G 2
#endif
#ifndef DENSITY_WALK_JIT
#define DENSITY_WALK_JIT 1
#endif
constexpr uint32_t kTeam = DENSITY_WALK_TEAM, kTeamLds = kWalkTable + 64;
template <uint32_t G>
__global__ __launch_bounds__(kTeam * 64) void cheetah_walk_team(PassArgs a) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint64_t chunk = blockIdx.x;
    const ChunkInfo ci = a.info[chunk];
    if (ci.bad) return;
    const uint32_t nsteps = ci.blocks * kRecQuads;
    const uint32_t nblk = (nsteps + 63u) / 64u;
    const uint32_t ngroups = (nblk + G - 1u) / G;
    const uint64_t s0 = chunk * (a.out_stride / 4);
    const uint32_t* __restrict__ desc = a.desc + s0;
    uint16_t* __restrict__ ctx = a.ctx + s0;
    {   // H starts as the hash of the reference's zeroed prediction table: hash(0) = 0; token 0, running context 0 (cheetah.rs:52)
        uint4* p = reinterpret_cast<uint4*>(pass_lds);
        for (uint32_t i = threadIdx.x; i < kTeamLds / 16; i += kTeam * 64) p[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    const uint32_t lds0 = lds_addr(pass_lds);
    const uint32_t token = lds0 + kWalkTable;                                       // {the group whose turn it is, the running context in front of it}: one 8-byte word
    uint32_t nd[G], nprev = 0;
    auto fetch = [&](uint32_t g) {
#pragma unroll
        for (uint32_t b = 0; b < G; ++b) {
            const uint32_t i = (g * G + b) * 64u + lane;
            nd[b] = i < nsteps ? desc[i] : kDescNone;
        }
        nprev = g ? desc[g * G * 64u - 1u] : 0u;                                      // the quad in front of the group
    };
    if (wave < ngroups) fetch(wave);
    for (uint32_t g = wave; g < ngroups; g += kTeam) {
        const uint32_t blk = g * G;
        uint32_t dv[G], hv[G], hwv[G], cvv[G], rsv[G], rfv[G], r2v[G], shv[G];
        uint64_t Pm[G], Nm[G], K0m[G], known[G], fin[G], rdone[G];
        const uint32_t dprev = rfl(nprev);
        bool ok = blk + G <= nblk;
#pragma unroll
        for (uint32_t b = 0; b < G; ++b) {
            dv[b] = nd[b];
            hv[b] = dv[b] & 0xffffu;
            hwv[b] = (dv[b] & kDescZero) ? 0u : hv[b];
            const bool none = (dv[b] & kDescNone) != 0, pred = ((dv[b] >> 16) & 3u) == kFlagPred;
            Pm[b] = ballot64(!none && pred); Nm[b] = ballot64(!none && !pred);
            uint64_t lr = Pm[b] & (Pm[b] >> 1); lr &= lr >> 2; lr &= lr >> 4;
            ok = ok && (Pm[b] | Nm[b]) == ~0ull && lr == 0;
        }
        if (g + kTeam < ngroups) fetch(g + kTeam);                                    // in flight across this turn
        // the context in front of the group, where the descriptors tell it: the hash of a quad that took part and was not predicted
        const bool c_known = g == 0 || (!(dprev & kDescNone) && ((dprev >> 16) & 3u) != kFlagPred);
        const uint32_t c_spec = g == 0 ? 0u : (dprev & 0xffffu);
        // speculate: every level's reads of all G blocks in flight together (lanes whose context is known and who have not read yet)
        auto speculate = [&]() __attribute__((always_inline)) {
            for (;;) {
                uint64_t R[G], any = 0;
#pragma unroll
                for (uint32_t b = 0; b < G; ++b) { R[b] = Pm[b] & known[b] & ~rdone[b]; any |= R[b]; }
                if (!any) break;
                uint32_t r[G], ad[G];
#pragma unroll
                for (uint32_t b = 0; b < G; ++b) ad[b] = lds0 + 2u * cvv[b];
                lds_read_u16_group<G>(r, ad);                                          // (every block reads, whether or not one of its lanes needs it: a stale context is a valid address)
#pragma unroll
                for (uint32_t b = 0; b < G; ++b) {
                    const uint64_t in = (R[b] << 1) | (b == 0 ? 0ull : (R[b ? b - 1 : 0] >> 63));   // the lanes that learn their context this round
                    if (R[b]) rsv[b] = msel(R[b], r[b], rsv[b]);
                    if (in) {
                        uint32_t up = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)r[b], 0x138, 0xf, 0xf, false);   // wave_shr:1
                        if (b != 0 && (in & 1ull)) { const uint32_t carry = (uint32_t)__builtin_amdgcn_readlane((int)r[b ? b - 1 : 0], 63); up = lane == 0 ? carry : up; }
                        cvv[b] = msel(in, up, cvv[b]);
                    }
                    known[b] |= in; rdone[b] |= R[b];
                }
            }
        };
        // the ordered pass's operands: made AHEAD of the turn too (only a patched lane 0 or a wrong read makes them again)
        uint32_t xa[G], xm[G], xv[G];
        auto prepare_exec = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (uint32_t b = 0; b < G; ++b) {
                shv[b] = (cvv[b] & 1u) * 16u;
                const uint64_t w = Nm[b] & ~fin[b];
                xa[b] = lds0 + ((2u * cvv[b]) & ~3u);
                xm[b] = msel(w, 0xffffu << shv[b], 0u); xv[b] = msel(w, hwv[b] << shv[b], 0u);
            }
        };
        if (ok) {
#pragma unroll
            for (uint32_t b = 0; b < G; ++b) {
                const uint32_t hp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hv[b], 0x138, 0xf, 0xf, false);   // wave_shr:1
                const uint32_t first = b == 0 ? c_spec : (uint32_t)__builtin_amdgcn_readlane((int)hv[b ? b - 1 : 0], 63);   // (meaningful only where the quad before was not predicted)
                cvv[b] = lane == 0 ? first : hp;
                K0m[b] = (Nm[b] << 1) | (b == 0 ? (c_known ? 1ull : 0ull) : (Nm[b ? b - 1 : 0] >> 63));
                known[b] = K0m[b]; fin[b] = 0; rsv[b] = 0; rfv[b] = 0; rdone[b] = 0;
            }
            if (DENSITY_WALK_JIT && g >= DENSITY_WALK_JIT) {                              // not before my turn is DENSITY_WALK_JIT turns away: what is read earlier is stale more often than not
                for (uint32_t spins = 0; spins < kSpinLimit; ++spins) {
                    uint32_t seen;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(token) : "memory");
                    seen = rfl(seen);
                    if (seen + DENSITY_WALK_JIT >= g) break;
                }
            }
            speculate();                                                              // AHEAD of my turn: H as it stands
            prepare_exec();
        }
        // ---- my turn ----
        uint32_t c;
        {
            bool poisoned = false;
            for (uint32_t spins = 0;; ++spins) {
                uint64_t tc;
                asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(tc) : "v"(token) : "memory");
                const uint32_t seen = rfl((uint32_t)tc);
                if (seen == g) { c = rfl((uint32_t)(tc >> 32)); break; }
                if (seen == kPoison || spins > kSpinLimit) {
                    if (seen != kPoison && lane == 0) { atomicOr(a.err, kErrWatchdog); *reinterpret_cast<volatile uint32_t*>(pass_lds + kWalkTable) = kPoison; }
                    poisoned = true;
                    break;
                }
            }
            if (poisoned) break;
        }
        if (ok) {
            if (!c_known) {                                                          // the predecessor's last quad was predicted (or took no part): its hash is what the predecessor says
                cvv[0] = lane == 0 ? c : cvv[0];
                K0m[0] |= 1ull; known[0] |= 1ull;
                speculate();                                                          // (what lane 0's context sets free)
                prepare_exec();
            }
            for (;;) {
                // execute: the lanes that do not stand yet, block after block, each in lane order
                uint32_t ret[G];
                lds_mskor_group<G>(ret, xa, xm, xv);
                // verify
                uint64_t bad[G], anybad = 0;
#pragma unroll
                for (uint32_t b = 0; b < G; ++b) {
                    r2v[b] = (ret[b] >> shv[b]) & 0xffffu;
                    // a predicted lane that has not read at all yet (its context never became known: cannot happen once lane 0's is) counts as wrong
                    bad[b] = (ballot64(r2v[b] != rsv[b]) | ~rdone[b]) & Pm[b] & ~fin[b];
                    anybad |= bad[b];
                }
                if (__builtin_expect(anybad == 0, 1)) {
#pragma unroll
                    for (uint32_t b = 0; b < G; ++b) rfv[b] = msel(~fin[b], r2v[b], rfv[b]);
                    break;
                }
                uint32_t b0 = 0;
#pragma unroll
                for (uint32_t b = G; b-- > 0;) if (bad[b]) b0 = b;
                uint64_t badb = bad[0];
#pragma unroll
                for (uint32_t b = 1; b < G; ++b) badb = b0 == b ? bad[b] : badb;
                const uint32_t i0 = (uint32_t)__builtin_ctzll(badb);
                const uint64_t upto = (2ull << i0) - 1ull;                           // lanes 0 .. i0 of block b0 (i0 == 63: all of them)
                uint64_t stands[G];
#pragma unroll
                for (uint32_t b = 0; b < G; ++b) stands[b] = b < b0 ? ~0ull : b == b0 ? upto : 0ull;
                // the writes behind the first wrong read are taken back: old halves, the latest write first (blocks from the last to b0, lanes reversed)
#pragma unroll
                for (uint32_t b = G; b-- > 0;) {
                    const uint64_t undo = Nm[b] & ~stands[b] & ~fin[b];
                    if (undo) {
                        const uint32_t ar = bperm(63u - lane, lds0 + 2u * cvv[b]), old = bperm(63u - lane, r2v[b]);
                        if ((undo >> (63u - lane)) & 1ull) asm volatile("ds_write_b16 %0, %1" ::"v"(ar), "v"(old) : "memory");
                    }
                }
                const uint32_t truth = (uint32_t)__builtin_amdgcn_readlane((int)(b0 == 0 ? r2v[0] : b0 == 1 ? r2v[G > 1 ? 1 : 0] : b0 == 2 ? r2v[G > 2 ? 2 : 0] : r2v[G > 3 ? 3 : 0]), (int)i0);
                bool all = true;
                uint64_t nextm[G];
#pragma unroll
                for (uint32_t b = 0; b < G; ++b) {
                    rfv[b] = msel(stands[b] & ~fin[b], r2v[b], rfv[b]);
                    fin[b] = stands[b];
                    all = all && fin[b] == ~0ull;
                    // the lane behind (b0, i0) now knows its context; everything else behind it is as unknown as before the first round
                    const uint64_t next = b == b0 ? (i0 == 63u ? 0ull : (2ull << i0) & ~upto) : (b == b0 + 1u && i0 == 63u ? 1ull : 0ull);
                    if (next) cvv[b] = msel(next, truth, cvv[b]);
                    nextm[b] = next;
                }
                if (all) break;
                // What has to be guessed again is the CHAIN behind the wrong read — the lane that now knows its context, the run of predicted lanes it starts
                // and the lane behind that run —, not the block: every other lane's context and speculative read are as good a guess as they were, and the
                // next verification holds all of them to the ordered pass again.  (A chain that runs on into the next block: everything behind the wrong read, as before.)
                const uint32_t cb = i0 == 63u ? b0 + 1u : b0, start = i0 == 63u ? 0u : i0 + 1u;
                uint64_t pcb = Pm[0];
#pragma unroll
                for (uint32_t b = 1; b < G; ++b) pcb = cb == b ? Pm[b] : pcb;
                const uint64_t inv = ~(pcb >> start);
                const uint32_t end = start + (inv ? (uint32_t)__builtin_ctzll(inv) : 64u);   // the lane behind the run (the chain's last)
                if (end <= 63u) {
                    const uint64_t A = ((2ull << end) - 1ull) & ~((1ull << start) - 1ull);
#pragma unroll
                    for (uint32_t b = 0; b < G; ++b)
                        if (b == cb) { known[b] = (known[b] & ~A) | nextm[b]; rdone[b] &= ~A; }
                } else {
#pragma unroll
                    for (uint32_t b = 0; b < G; ++b) { known[b] = fin[b] | nextm[b] | K0m[b]; rdone[b] = fin[b]; }
                }
                speculate();
                prepare_exec();
            }
            const uint32_t last = msel(Pm[G - 1], rfv[G - 1], hv[G - 1]);
            c = (uint32_t)__builtin_amdgcn_readlane((int)last, 63);
        } else {
            // block by block, run by run (cheetah.rs:97-102; raw-copy blocks and quads beyond the end take no part: the context passes through)
            auto one_block = [&](auto bc) __attribute__((always_inline)) {                                           // (written out per block: left as a loop over b the compiler keeps it rolled and the masks in vector registers)
                constexpr uint32_t b = decltype(bc)::value;
                if (blk + b >= nblk) { cvv[b] = 0; return; }
                const uint32_t h = hv[b], hw = hwv[b];
                const uint64_t P = Pm[b], N = Nm[b], active = P | N;
                const uint32_t hprev = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)h, 0x138, 0xf, 0xf, false);   // wave_shr:1
                if (active == ~0ull && P != 0 && P != ~0ull) {                        // every quad takes part, some are predicted: the hand-written chain (see cheetah_walk)
                    cvv[b] = walk_chain_block(lds0, c, hprev, h, hw, P);
                    return;
                }
                uint32_t cv = 0, pos = 0;
                while (pos < 64u) {
                    const uint64_t rest = active >> pos;
                    if (!rest) break;
                    pos += (uint32_t)__builtin_ctzll(rest);
                    if ((N >> pos) & 1ull) {
                        const uint64_t inv = ~(N >> pos);
                        const uint32_t r = inv ? (uint32_t)__builtin_ctzll(inv) : 64u - pos;
                        const bool in = lane >= pos && lane < pos + r;
                        const uint32_t mine = lane == pos ? c : hprev;
                        if (in) {
                            cv = mine;
                            asm volatile("ds_write_b16 %0, %1" ::"v"(lds0 + 2u * mine), "v"(hw) : "memory");
                        }
                        c = (uint32_t)__builtin_amdgcn_readlane((int)h, (int)(pos + r - 1u));
                        pos += r;
                    } else {
                        const uint64_t inv = ~(P >> pos);
                        const uint32_t r = inv ? (uint32_t)__builtin_ctzll(inv) : 64u - pos;
                        for (uint32_t t = 0; t < r; ++t) {
                            if (lane == pos + t) cv = c;
                            uint32_t nx;
                            asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(nx) : "v"(lds0 + 2u * c) : "memory");
                            nx = rfl(nx);
                            if (nx == c) {                                             // a fixed point: the table does not change inside a run
                                if (lane > pos + t && lane < pos + r) cv = c;
                                break;
                            }
                            c = nx;
                        }
                        pos += r;
                    }
                }
                cvv[b] = cv;
            };
            one_block(std::integral_constant<uint32_t, 0>{});
            if constexpr (G > 1) one_block(std::integral_constant<uint32_t, 1>{});
            if constexpr (G > 2) { one_block(std::integral_constant<uint32_t, 2>{}); one_block(std::integral_constant<uint32_t, 3>{}); }
        }
        // hand on: the token and the running context in one 8-byte write, behind everything this turn did to H (a wave's LDS operations execute as issued)
        {
            const uint64_t tc = (uint64_t)(g + 1u) | ((uint64_t)c << 32);
            asm volatile("ds_write_b64 %0, %1" ::"v"(token), "v"(tc) : "memory");
        }
#pragma unroll
        for (uint32_t b = 0; b < G; ++b) { const uint32_t i = (blk + b) * 64u + lane; if (i < nsteps) ctx[i] = (uint16_t)cvv[b]; }
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
// scratch: rec (a dword per 128-byte block) | desc (a dword per quad) | ctx (16 bits per quad) | chunk info
uint64_t decode_pass_scratch_bytes(uint64_t out_stride, uint32_t n_chunks) {
    const uint64_t span = (uint64_t)n_chunks * out_stride;                         // (arrays are indexed chunk * stride + ...)
    return ((span / kRecBytes * 4 + 255) & ~255ull) + ((span + 255) & ~255ull) + ((span / 2 + 255) & ~255ull) + (((uint64_t)n_chunks * sizeof(ChunkInfo) + 255) & ~255ull) + 256;
}
bool decode_pass_eligible(int algo, const uint8_t* d_out, uint32_t n_chunks, uint64_t out_stride, uint64_t out_total) {
    if (algo != DENSITY_HIP_CHEETAH || g_force_serial_decode || g_force_lane_codec || g_force_wave_codec || g_rotor_unsafe) return false;
    if (n_chunks == 0 || (uintptr_t)d_out % 4 != 0) return false;
    // one chunk (a stream): the output capacity is the stride; chunks: whole pairs of records
    if (n_chunks > 1 && out_stride % 256 != 0) return false;
    // Where the passes win: from 64 KiB chunks on (measured, 100 MB of prose, decode ms passes / one wave per stream: 64 KiB 3.2 / 3.6,
    // 128 KiB 3.3 / 4.2, 256 KiB 4.4 / 6.2, 512 KiB 5.2 / 10.5, 1 MiB 9.8 / 20.2); shorter chunks have less to parse than the passes have to
    // set up.  DENSITY_HIP_PASS_MIN (bytes; tuning runs) moves the threshold.
    static const uint64_t min_chunk = debug_env("DENSITY_HIP_PASS_MIN") ? (uint64_t)atoll(debug_env("DENSITY_HIP_PASS_MIN")) : 64ull * 1024;
    const uint64_t per_chunk = n_chunks == 1 ? out_total : out_stride;
    return per_chunk >= min_chunk && per_chunk >= 65536 && per_chunk < (1ull << 31);
}
hipError_t launch_decode_passes(int algo, const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out,
                                uint64_t out_stride, uint64_t out_total, bool exact, uint64_t* d_produced, uint32_t* d_err, uint8_t* d_scratch,
                                hipStream_t stream);

namespace {
__global__ __launch_bounds__(256) void cheetah_finish(PassArgs a, uint32_t exact, uint64_t* __restrict__ produced) {
    const uint32_t chunk = blockIdx.x * 256u + threadIdx.x;
    if (chunk >= a.n_chunks) return;
    const ChunkInfo ci = a.info[chunk];
    produced[chunk] = ci.bad ? 0 : ci.produced;
    if (!ci.bad && exact && ci.produced != chunk_cap(a, chunk)) atomicOr(a.err, kErrFormat);
}
}  // namespace

hipError_t launch_decode_passes(int algo, const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out,
                                uint64_t out_stride, uint64_t out_total, bool exact, uint64_t* d_produced, uint32_t* d_err, uint8_t* d_scratch,
                                hipStream_t stream) {
    (void)algo;
    if (n_chunks == 1) out_stride = (out_total + 255) & ~255ull;                    // a single stream: its chunk is the whole output
    PassArgs a{};
    a.in = d_in; a.offsets = d_offsets; a.sizes = d_sizes; a.n_chunks = n_chunks; a.out = d_out; a.out_stride = out_stride; a.out_total = out_total;
    const uint64_t span = (uint64_t)n_chunks * out_stride;
    uint8_t* p = d_scratch;
    a.rec = reinterpret_cast<uint32_t*>(p); p += (span / kRecBytes * 4 + 255) & ~255ull;
    a.desc = reinterpret_cast<uint32_t*>(p); p += (span + 255) & ~255ull;
    a.ctx = reinterpret_cast<uint16_t*>(p); p += (span / 2 + 255) & ~255ull;
    a.info = reinterpret_cast<ChunkInfo*>(p);
    a.err = d_err;
    const uint32_t blocks_per_chunk = (uint32_t)(out_stride / kRecBytes);
    hipError_t e = hipFuncSetAttribute((const void*)cheetah_pass<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pass_lds_bytes(1));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)cheetah_pass<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pass_lds_bytes(2));
    // the walk: a team of four waves, 128 quads a turn (default); g_walk_blocks 1 / 4: ONE wave, 64 / 128 quads at a time; g_chain_walk: one wave, run by run
    const bool team = !g_chain_walk && g_walk_blocks == 2;
    auto walk = g_chain_walk ? cheetah_walk<0> : g_walk_blocks == 1 ? cheetah_walk<1> : cheetah_walk<2>;
    if (e == hipSuccess) e = team ? hipFuncSetAttribute((const void*)cheetah_walk_team<DENSITY_WALK_G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTeamLds)
                                  : hipFuncSetAttribute((const void*)walk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWalkLds);
    if (e != hipSuccess) return e;
    // the records of calm stretches are found by the window kernels (their tables live where the descriptors and contexts will: nothing else is in use yet)
    const uint64_t slot_bound = out_stride + out_stride / kRecBytes * kSigBytes + kSigBytes;
    const uint32_t wpc = (uint32_t)((slot_bound + kPW - 1) / kPW);
    // (the window kernels put the chunk on the grid's y axis: at most 65535 chunks — 4 GiB and more in 64 KiB chunks take the one-wave parse)
    const bool windows = !g_serial_parse && n_chunks <= 65535u && out_stride >= (64u << 10) && (uint64_t)wpc * kPE * 4 <= out_stride && (uint64_t)wpc * 8 <= out_stride / 2;
    if (windows) {
        uint32_t* T = a.desc;
        uint32_t* wbase = reinterpret_cast<uint32_t*>(a.ctx);
        uint8_t* went = reinterpret_cast<uint8_t*>(wbase + (uint64_t)n_chunks * wpc);
        const uint32_t table_lds = (uint32_t)std::min<uint64_t>((uint64_t)wpc * kPE * 4, 140u * 1024);   // (+ the walk's own 16 KiB of stream windows)
        e = hipFuncSetAttribute((const void*)cheetah_parse, hipFuncAttributeMaxDynamicSharedMemorySize, (int)table_lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(cheetah_parse, dim3(n_chunks), dim3(64), 0, stream, a, 2u, 0u, 0u, (const uint32_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr);
        hipLaunchKernelGGL(cheetah_parse_windows, dim3(wpc, n_chunks), dim3(128), 0, stream, a, wpc, T, went);
        if ((e = hipGetLastError()) != hipSuccess) return e;                      // (the walk below reads what these kernels wrote)
        hipLaunchKernelGGL(cheetah_parse, dim3(n_chunks), dim3(64), table_lds, stream, a, 1u, wpc, table_lds, (const uint32_t*)T, went, wbase);
        hipLaunchKernelGGL(cheetah_parse_emit, dim3(wpc, n_chunks), dim3(64), 0, stream, a, wpc, (const uint8_t*)went, (const uint32_t*)wbase);
    } else {
        hipLaunchKernelGGL(cheetah_parse, dim3(n_chunks), dim3(64), 0, stream, a, 0u, 0u, 0u, (const uint32_t*)nullptr, (uint8_t*)nullptr, (uint32_t*)nullptr);
    }
    const uint64_t pairs = (uint64_t)n_chunks * (blocks_per_chunk / 2);
    hipLaunchKernelGGL(cheetah_prepare, dim3((uint32_t)((pairs + 3) / 4)), dim3(256), 0, stream, a, blocks_per_chunk);
    hipLaunchKernelGGL(cheetah_pass<1>, dim3(4 * n_chunks), dim3(kPassWaves * 64), pass_lds_bytes(1), stream, a);
    if (team) hipLaunchKernelGGL(cheetah_walk_team<DENSITY_WALK_G>, dim3(n_chunks), dim3(kTeam * 64), kTeamLds, stream, a);
    else hipLaunchKernelGGL(walk, dim3(n_chunks), dim3(64), kWalkLds, stream, a);
    hipLaunchKernelGGL(cheetah_pass<2>, dim3(2 * n_chunks), dim3(kPassWaves * 64), pass_lds_bytes(2), stream, a);
    hipLaunchKernelGGL(cheetah_finish, dim3((n_chunks + 255) / 256), dim3(256), 0, stream, a, exact ? 1u : 0u, d_produced);
    return hipGetLastError();
}

}  // namespace density
