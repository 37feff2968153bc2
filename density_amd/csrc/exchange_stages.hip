// exchange_stages.hip — Cheetah and Lion container ENCODING as passes of ordered LDS exchanges (gfx950).
//
// The reference walks a chunk quad by quad through three tables (cheetah.rs:123-149) or seven (lion.rs:211-270).  Every one of those
// table steps is an unconditional EXCHANGE once it is known which quads take part:
//
//   Cheetah  P   old = xchg(pred[h(q[i-1])], q[i])   every quad;             predicted <=> old == q[i]   (a hit rewrites what is there)
//            A   old = xchg(a[h(q[i])], q[i])        the quads P missed;     MAP_A     <=> old == q[i]
//            B   old = xchg(b[h(q[i])], old_A)       the quads A missed too; MAP_B     <=> old == q[i]   (cheetah.rs:140-141: b = a, a = quad)
//   Lion     P0..P4: the five-deep move-to-front list per predictor slot (lion.rs:50-57, 240-262) is a chain of five such exchanges —
//            level k takes in what level k-1 displaced and the chain stops where it finds the quad — then A and B as above.
//
// so a stage can run over the WHOLE chunk before the next one starts, and a stage is "one table, the taking-part quads in stream
// order": what ds_mskor_rtn_b32 does for 64 quads in one instruction (ascending-lane service order, verified at start-up —
// rotor.hip::rotor_selftest_kernel; with a full mask it is an exchange).  A table of 64 Ki dwords does not fit the LDS, a half does:
// one work-group per (chunk, half of the slots) streams the chunk and exchanges the quads whose slot falls in its half, the others
// exchange into a sink word of their own.  Three / seven launches, 2 x n_chunks work-groups each, no dependent memory round trip per
// record; then the record sizes are scanned per chunk and the records written by as many waves as there are 256-byte blocks.
// (tools/exchange_stage_model.py restates this in Python; tests/test_exchange_stage_model.py checks it against the oracle.)
//
// What the passes cannot know is the blow-up protection (codec.rs:35-37): a raw-copy block takes its quads out of every table, and
// whether a block is copied depends on the sizes of the records before it.  Raw copies cluster where the dictionary is cold — the
// first KiBs of every chunk — so:
//   * the HEAD of every chunk is encoded in order by the one-wave kernel of serial_codec.hip — at least 8 KiB (Cheetah; rounds 3-5: 16) / 48 KiB (Lion),
//     and on to the first 4 KiB boundary where no block has been copied for 8 / 32 KiB (a chunk too short for that, or still restless
//     four heads in, is simply finished there; the numbers: no late raw copy in 100 MB of prose and 64 MB of repetitive text) — which leaves its tables in global memory; the stages load their half of their table
//     from there instead of starting from zeros;
//   * behind the head the passes run as if no block were copied; the size scan sees whether two incompressible records ever meet
//     (protection_state.rs:38-47) — such a chunk is done again, whole, by the in-order kernel (`only` filter);
//   * a RAGGED END (a chunk that is not whole 4 KiB trips) goes back to the in-order kernel as well, but only the end: the stages
//     write their tables back, the size scan advances the FSM counters over the calm blocks, the wave resumes from there.
#include "common.hpp"
#include "kernels.hpp"

#include <cstdlib>
#include <vector>

namespace density {

extern __shared__ __attribute__((aligned(16))) uint8_t stage_lds[];
bool g_force_wave_codec = false;   // density_hip_set_kernel_variant(32)
bool g_stage_audit = false;        // density_hip_set_kernel_variant(64): count the chunks handed back (reads the verdicts: synchronises)
uint64_t g_stage_stats[2] = {0, 0}; // density_hip_stage_stats: chunks through the exchange passes | of those, handed back to the in-order kernel

namespace {

constexpr uint32_t kHalfSlots = 32768;                   // slots per work-group
constexpr uint32_t kTable = kHalfSlots * 4;              // 128 KiB of dwords
constexpr uint32_t kAhead = 16;                          // blocks of 64 quads per trip: one statement of 16 exchanges, and what a wave has in flight from memory
constexpr uint32_t kTrip = kAhead * 256;                 // bytes per trip: the passes cover whole trips
// waves of a stage work-group, taking the trips in rotation: 8; Cheetah's three stages 12 (round 6, same-box A/B: 0.735 -> 0.709 ms per 100 MB; Lion's
// level stages spill registers at three waves per SIMD)
constexpr uint32_t kStageWaves = 8, kStageWavesCheetah = 12;
constexpr uint32_t kSpinLimit = 1u << 22, kPoison = 0xfffffffeu, kErrWatchdog = 16u;   // (as rotor.hip)
// LDS: the half table | a sink word per lane (the quads of the other half / of earlier stages; the token store of lanes 1..63) | the token
constexpr uint32_t stage_lds_bytes(uint32_t waves) { return kTable + waves * 64 * 4 + 16; }

// Cheetah's head: two trips (8 KiB), handed over once no block has been copied for two trips.  Rounds 3-5: four and four; round 6 (tools/gpu_head_audit.py, 64 MiB per
// kind, chunks of 256 KiB - 1 MiB): no chunk of prose, repetitive text, vocabulary draws, binary-like records, pair runs or zeros comes back from the passes with
// either; the heads are a wave per chunk on an otherwise idle device, so half the head is 0.707 -> 0.653 ms per 100 MB of prose (zeros, pair runs: -30 %); a
// text / random patchwork, which lives in the in-order kernel either way, +3 %; ONE trip never calms down inside its four heads and sends every chunk back (3.5 ms)
#ifndef DENSITY_CHEETAH_HEAD_TRIPS
#define DENSITY_CHEETAH_HEAD_TRIPS 2
#define DENSITY_CHEETAH_CALM_TRIPS 2
#endif
// per algorithm: stages, record geometry (cheetah.rs:17-23,188-196; lion.rs:17-27,317-325), the in-order head, the table slot of serial_codec.hip
template <int ALGO> struct StageGeo;
template <> struct StageGeo<DENSITY_HIP_CHEETAH> {
    static constexpr uint32_t kStages = 3, kRecQuads = 32, kSig = 8, kRec = 128, kHeadBytes = DENSITY_CHEETAH_HEAD_TRIPS * kTrip, kHeadCalm = DENSITY_CHEETAH_CALM_TRIPS * kTrip;
    static constexpr uint64_t kChunkTables = 65536ull * (8 + 4);
};
template <> struct StageGeo<DENSITY_HIP_LION> {
    static constexpr uint32_t kStages = 7, kRecQuads = 16, kSig = 6, kRec = 64, kHeadBytes = 12 * kTrip, kHeadCalm = 8 * kTrip;
    static constexpr uint64_t kChunkTables = 65536ull * (8 + 20);
};

__device__ __forceinline__ uint32_t hash16(uint32_t q) { return (q * kHashMul) >> 16; }

// The critical section of a trip: sixteen ordered exchanges (the answer comes back in the address register), the token for the next
// trip written right behind them (the LDS takes a wave's instructions in order: whoever sees the token queues up behind these
// exchanges — rotor.hip's hand-off, self-tested at start-up), then the answers.
#define DENSITY_STAGE_X16_TOKEN(ra, v, ones, tokaddr, tokval)                                                                     \
    asm volatile(                                                                                                                 \
        "ds_mskor_rtn_b32 %0, %0, %32, %16\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %1, %1, %32, %17\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %2, %2, %32, %18\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %3, %3, %32, %19\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %4, %4, %32, %20\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %5, %5, %32, %21\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %6, %6, %32, %22\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %7, %7, %32, %23\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %8, %8, %32, %24\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %9, %9, %32, %25\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %10, %10, %32, %26\n\t"                                                                                 \
        "ds_mskor_rtn_b32 %11, %11, %32, %27\n\t"                                                                                 \
        "ds_mskor_rtn_b32 %12, %12, %32, %28\n\t"                                                                                 \
        "ds_mskor_rtn_b32 %13, %13, %32, %29\n\t"                                                                                 \
        "ds_mskor_rtn_b32 %14, %14, %32, %30\n\t"                                                                                 \
        "ds_mskor_rtn_b32 %15, %15, %32, %31\n\t"                                                                                 \
        "ds_write_b32 %33, %34\n\t"                                                                                               \
        "s_waitcnt lgkmcnt(0)"                                                                                                    \
        : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]), "+v"(ra[6]), "+v"(ra[7]),                 \
          "+v"(ra[8]), "+v"(ra[9]), "+v"(ra[10]), "+v"(ra[11]), "+v"(ra[12]), "+v"(ra[13]), "+v"(ra[14]), "+v"(ra[15])            \
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),                                 \
          "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]),                           \
          "v"(ones), "v"(tokaddr), "v"(tokval)                                                                                    \
        : "memory")

// where a stage's table lives in the chunk's table slot of serial_codec.hip (64 Ki {a, b} pairs, then 64 Ki x 1 or 5 predictions):
// byte offset of slot 0's word and the stride from slot to slot
struct StagePlace { uint64_t chunk_tables; uint32_t offset, stride; };

// One stage over one chunk for one half of the slots: a work-group of W waves that take the trips of 16 blocks in rotation — loads,
// hashes, ballots and stores of one wave beside the exchanges of another; the exchanges themselves in stream order behind the token.
//   KEY_PREV    the slot is the hash of the quad BEFORE (the predictor levels) / of the quad itself (A, B)
//   OWN_VALUE   the value exchanged in is the quad (first predictor level, A) / what the previous stage displaced (vals[])
//   KEEP_OLD    the displaced value is kept in vals[] for the next stage
//   HAS_BEFORE  not the first stage: the quads an earlier stage settled take no part
// done_prev / done_out: per 64-quad block of the whole input four dwords — lanes 0..31 of half 0, of half 1, lanes 32..63 of half 0, of
// half 1 — "quad settled by this stage or an earlier one" (cumulative; a reader ORs the two halves: one 8-byte load per lane).
template <bool KEY_PREV, bool OWN_VALUE, bool KEEP_OLD, bool HAS_BEFORE, uint32_t W>
__global__ __launch_bounds__(W * 64) void exchange_stage(const uint8_t* __restrict__ in, uint64_t total, uint64_t chunk_bytes,
                                                         const uint32_t* __restrict__ done_prev, uint32_t* __restrict__ done_out,
                                                         uint32_t* __restrict__ vals, uint8_t* __restrict__ tables,
                                                         const uint32_t* __restrict__ head_state, StagePlace place, uint32_t* __restrict__ fault) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = rfl(threadIdx.x >> 6);
    const uint64_t chunk = blockIdx.x >> 1;
    const uint32_t half = blockIdx.x & 1u;
    const uint64_t base = chunk * chunk_bytes;
    const uint64_t len = (total - base) < chunk_bytes ? (total - base) : chunk_bytes;
    if (head_state[8 * chunk + 3]) return;                                        // short or restless: the one-wave kernel has finished it
    const uint32_t nb = (uint32_t)(len / kTrip) * kAhead;                         // whole trips; a ragged end is the in-order kernel's again
    const uint64_t gb0 = base >> 8;
    uint8_t* mine_tb = tables + chunk * place.chunk_tables + place.offset + (uint64_t)half * kHalfSlots * place.stride;
    uint32_t* w = reinterpret_cast<uint32_t*>(stage_lds);
    {
        // the table as the head left it: this half of the slots
#pragma unroll 8
        for (uint32_t k = threadIdx.x; k < kHalfSlots; k += W * 64) w[k] = *reinterpret_cast<const uint32_t*>(mine_tb + (uint64_t)k * place.stride);
        w[kHalfSlots + threadIdx.x] = 0u;                                         // the sinks
        if (threadIdx.x == 0) w[kHalfSlots + W * 64] = 0u;                        // the token: the trip whose exchanges go next
        __syncthreads();
    }
    const uint32_t* __restrict__ q32 = reinterpret_cast<const uint32_t*>(in + base);
    uint32_t* __restrict__ v32 = vals + gb0 * 64;
    const uint2* __restrict__ before2 = reinterpret_cast<const uint2*>(done_prev + gb0 * 4) + (lane >> 5);   // + 2 per block
    uint32_t* __restrict__ mine = done_out + gb0 * 4 + half;                                                 // + 4 per block, + 2 for the upper lanes
    const uint32_t lds0 = lds_addr(stage_lds);
    const uint32_t sink = lds0 + kTable + threadIdx.x * 4u;
    const uint32_t token = lds0 + kTable + W * 256u;
    const uint32_t ones = 0xffffffffu;
    const uint32_t hb = head_state[8 * chunk + 6] >> 8;                           // blocks of the in-order head (whole trips)
    const uint32_t trips = (nb - hb) / kAhead;
    const uint32_t last_hash = head_state[8 * chunk + 1];                         // cheetah.rs:146 / lion.rs:268 as the head left it (its last block may be a raw copy)

    if (wave < trips) {
        uint32_t qn[kAhead], pn[kAhead], vn[kAhead];
        uint2 bn[kAhead];
#pragma unroll
        for (uint32_t j = 0; j < kAhead; ++j) {
            const uint32_t i = (hb + wave * kAhead + j) * 64u + lane;
            qn[j] = q32[i];
            pn[j] = KEY_PREV ? q32[i - 1] : 0u;
            vn[j] = !OWN_VALUE ? v32[i] : 0u;
            bn[j] = HAS_BEFORE ? before2[(hb + wave * kAhead + j) * 2u] : make_uint2(0u, 0u);
        }
        for (uint32_t t = wave; t < trips; t += W) {
            const uint32_t g = hb + t * kAhead;
            uint32_t q[kAhead], key[kAhead], val[kAhead];
            bool before[kAhead];
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) {
                q[j] = qn[j];
                val[j] = OWN_VALUE ? qn[j] : vn[j];
                before[j] = HAS_BEFORE && (((bn[j].x | bn[j].y) >> (lane & 31u)) & 1u);
                // slot: cheetah.rs:125 / lion.rs:213 (the predictor is addressed by the previous quad's hash), cheetah.rs:131 / lion.rs:245
                key[j] = KEY_PREV ? hash16(pn[j]) : hash16(qn[j]);
            }
            if (KEY_PREV && t == 0 && lane == 0) key[0] = last_hash;
            {                                                                      // this wave's next trip: in flight across this one
                const uint32_t gn = hb + (t + W < trips ? t + W : t) * kAhead;     // (its last trip loads itself again: no branch, nothing out of bounds)
#pragma unroll
                for (uint32_t j = 0; j < kAhead; ++j) {
                    const uint32_t i = (gn + j) * 64u + lane;
                    qn[j] = q32[i];
                    if (KEY_PREV) pn[j] = q32[i - 1];
                    if (!OWN_VALUE) vn[j] = v32[i];
                    if (HAS_BEFORE) bn[j] = before2[(gn + j) * 2u];
                }
            }
            uint32_t ra[kAhead];                                                   // slot address in, what the slot held out
            bool part[kAhead];
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) {
                part[j] = !before[j] && (key[j] >> 15) == half;
                ra[j] = part[j] ? lds0 + (key[j] & (kHalfSlots - 1u)) * 4u : sink;
            }
            // (the operands are made HERE: left to itself the compiler sinks the hashing of a first stage — 150 instructions — behind the poll loop, into the
            // critical section every later trip of the work-group waits for)
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) asm volatile("" : "+v"(ra[j]), "+v"(val[j]));
            if (W > 1) {                                                           // my turn: every earlier trip's exchanges are queued
                bool poisoned = false;
                for (uint32_t spins = 0;; ++spins) {
                    uint32_t seen;
                    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(token) : "memory");
                    seen = rfl(seen);
                    if (seen == t) break;
                    if (seen == kPoison || spins > kSpinLimit) {
                        if (seen != kPoison && lane == 0) { atomicOr(fault, kErrWatchdog); w[kHalfSlots + W * 64] = kPoison; }
                        poisoned = true;
                        break;
                    }
                }
                if (poisoned) break;
            }
            {
                const uint32_t tokaddr = lane == 0 ? token : sink, tokval = t + 1u;
                DENSITY_STAGE_X16_TOKEN(ra, val, ones, tokaddr, tokval);
            }
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) {
                const bool hit = part[j] && ra[j] == q[j];
                const uint64_t settled = ballot64(hit || before[j]);
                if (KEEP_OLD && part[j] && !hit) v32[(g + j) * 64u + lane] = ra[j];
                if (lane < 2) mine[(g + j) * 4u + lane * 2u] = lane ? (uint32_t)(settled >> 32) : (uint32_t)settled;
            }
        }
    }
    if (len % kTrip) {                                                             // a ragged end follows: the table goes back where it came from
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < kHalfSlots; k += W * 64) *reinterpret_cast<uint32_t*>(mine_tb + (uint64_t)k * place.stride) = w[k];
    }
}

// The settled-by masks of a 64-quad block, one per stage, cumulative: predicted = through the last predictor level, coded = through
// MAP_B.  Predicted quads cost nothing, MAP_A / MAP_B two bytes, plain quads four.
template <int ALGO>
struct BlockMasks {
    uint64_t m[StageGeo<ALGO>::kStages];
    __device__ __forceinline__ void load(const uint32_t* __restrict__ done, uint64_t blocks_total, uint64_t gb) {
#pragma unroll
        for (uint32_t s = 0; s < StageGeo<ALGO>::kStages; ++s) {
            const uint4 w = *reinterpret_cast<const uint4*>(done + (s * blocks_total + gb) * 4);   // lanes 0..31: halves 0, 1; lanes 32..63: halves 0, 1
            m[s] = (uint64_t)(w.x | w.y) | ((uint64_t)(w.z | w.w) << 32);
        }
    }
    __device__ __forceinline__ uint64_t predicted() const { return m[StageGeo<ALGO>::kStages - 3]; }
    __device__ __forceinline__ uint64_t coded() const { return m[StageGeo<ALGO>::kStages - 1]; }   // everything but the plain quads
    __device__ __forceinline__ uint32_t record_bytes(uint32_t r) const {                            // record r of the block
        constexpr uint32_t Q = StageGeo<ALGO>::kRecQuads;
        constexpr uint32_t mask = Q == 32 ? 0xffffffffu : 0xffffu;
        const uint32_t p = (uint32_t)(predicted() >> (Q * r)) & mask, c = (uint32_t)(coded() >> (Q * r)) & mask;
        const uint32_t plain = Q - (uint32_t)__builtin_popcount(c), maps = (uint32_t)__builtin_popcount(c) - (uint32_t)__builtin_popcount(p);
        return StageGeo<ALGO>::kSig + 4u * plain + 2u * maps;
    }
};

// per chunk: record offsets (exclusive scan of the record sizes), the stream length, whether the passes' assumption held, and where a
// ragged end resumes
constexpr uint32_t kLayoutThreads = 256;
template <int ALGO>
__global__ __launch_bounds__(kLayoutThreads) void stage_record_layout(uint64_t total, uint64_t chunk_bytes, const uint32_t* __restrict__ done,
                                                                      uint64_t blocks_total, const uint32_t* __restrict__ head_state,
                                                                      const uint8_t* __restrict__ in, uint32_t* __restrict__ rec_off,
                                                                      uint64_t* __restrict__ sizes, uint32_t* __restrict__ redo,
                                                                      uint32_t* __restrict__ tail_state) {
    using G = StageGeo<ALGO>;
    constexpr uint32_t kRecs = 64 / G::kRecQuads;
    __shared__ uint32_t s_sum[kLayoutThreads], s_first[kLayoutThreads], s_last[kLayoutThreads];
    const uint32_t t = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const uint64_t base = chunk * chunk_bytes;
    const uint64_t len = (total - base) < chunk_bytes ? (total - base) : chunk_bytes;
    if (head_state[8 * chunk + 3]) { if (t == 0) { redo[chunk] = 0u; tail_state[8 * chunk + 6] = 0u; } return; }   // finished by the head kernel (uniform)
    const uint32_t nb = (uint32_t)(len / kTrip) * kAhead;
    const uint64_t gb0 = base >> 8;
    const uint32_t head_blocks = head_state[8 * chunk + 6] >> 8;
    const uint32_t rest = nb - head_blocks;
    const uint32_t per = (rest + kLayoutThreads - 1) / kLayoutThreads;
    const uint32_t b0 = head_blocks + (t * per < rest ? t * per : rest), b1 = b0 + per < nb ? b0 + per : nb;
    const uint32_t head_fsm = head_state[8 * chunk + 2];
    uint32_t sum = 0, first = 0, last = 0, pair = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        BlockMasks<ALGO> m;
        m.load(done, blocks_total, gb0 + b);
        for (uint32_t r = 0; r < kRecs; ++r) {
            const uint32_t bytes = m.record_bytes(r);
            const uint32_t inc = bytes >= G::kRec ? 1u : 0u;                      // codec.rs:68
            if (b == b0 && r == 0) first = inc;
            else pair |= inc & last;
            last = inc;
            sum += bytes;
        }
    }
    s_sum[t] = sum; s_first[t] = first; s_last[t] = last;
    __syncthreads();
    uint32_t off = head_state[8 * chunk + 0];                                      // the head's records are in place
    for (uint32_t i = 0; i < t; ++i) off += s_sum[i];
    if (b0 < b1) pair |= s_first[t] & (t ? s_last[t - 1] : (head_fsm >> 1) & 1u);   // (threads with blocks are contiguous from 0, each full but the last)
    if (t == 0) pair |= head_fsm & 1u;                                             // the head ended inside a penalty: its copies are not over
    for (uint32_t b = b0; b < b1; ++b) {
        BlockMasks<ALGO> m;
        m.load(done, blocks_total, gb0 + b);
        for (uint32_t r = 0; r < kRecs; ++r) {
            rec_off[(gb0 + b) * kRecs + r] = off;
            off += m.record_bytes(r);
        }
    }
    const int any_pair = __syncthreads_or((int)pair);
    if (t == kLayoutThreads - 1) {
        // two incompressible records in a row start a penalty (protection_state.rs:38-47): raw copies would follow, the passes did not
        // see them — the whole chunk is done again by the in-order kernel, which also writes its size
        redo[chunk] = any_pair ? 1u : 0u;
        if (!any_pair) sizes[chunk] = off;
        // a ragged end: the in-order kernel goes on from here — where the passes stopped, with the FSM as the calm blocks in between
        // leave it (protection_state.rs:19-27: the counter runs, penalty_start halves every 16 blocks; no penalty was started)
        const bool ragged = !any_pair && (len % kTrip) != 0;
        uint32_t* ts = tail_state + 8 * chunk;
        ts[6] = ragged ? 1u : 0u;
        if (ragged) {
            uint32_t start = head_state[8 * chunk + 4], counter = head_state[8 * chunk + 5];
            for (uint32_t r = kRecs * head_blocks; r < kRecs * nb; ++r) {
                if (start == 1) { counter += kRecs * nb - r; break; }              // (only the counter's low four bits matter from here on)
                if ((counter & 0xfu) == 0) start >>= 1;
                ++counter;
            }
            // (thread 255 owns the last blocks, or none: the last record's verdict is the last one anybody saw)
            uint32_t prev = last;
            if (b0 >= b1) for (uint32_t i = kLayoutThreads - 1; i-- > 0;) if (s_sum[i]) { prev = s_last[i]; break; }
            ts[0] = nb * 256u;
            ts[1] = off;
            ts[2] = hash16(reinterpret_cast<const uint32_t*>(in + base)[nb * 64u - 1]);   // cheetah.rs:146 / lion.rs:268
            ts[3] = prev;
            ts[4] = start;
            ts[5] = counter;
        }
    }
}

// one flag bit plane of a record -> its place in the signature (io/write_signature.rs:14-17: quad k at bits 2k.. / 3k..)
__device__ __forceinline__ uint64_t spread_by2(uint32_t x) {
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}
__device__ __forceinline__ uint64_t spread_by3(uint32_t x16) {
    uint64_t x = x16 & 0xffffu;
    x = (x | (x << 16)) & 0x00ff0000ff0000ffull;
    x = (x | (x << 8)) & 0xf00f00f00f00f00full;
    x = (x | (x << 4)) & 0x30c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x9249249249249249ull;
    return x;
}

// the records: one wave per 64-quad block (two Cheetah records, four Lion records), a quad per lane
constexpr uint32_t kEmitWaves = 4;
template <int ALGO>
__global__ __launch_bounds__(kEmitWaves * 64) void stage_emit_records(const uint8_t* __restrict__ in, uint64_t total, uint64_t chunk_bytes,
                                                                       const uint32_t* __restrict__ done, uint64_t blocks_total,
                                                                       const uint32_t* __restrict__ rec_off, const uint32_t* __restrict__ redo,
                                                                       const uint32_t* __restrict__ head_state, uint8_t* __restrict__ out, uint64_t out_stride) {
    using G = StageGeo<ALGO>;
    constexpr uint32_t Q = G::kRecQuads, kRecs = 64 / Q;
    constexpr uint32_t rec_mask = Q == 32 ? 0xffffffffu : 0xffffu;
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t gb = (uint64_t)blockIdx.x * kEmitWaves + (threadIdx.x >> 6);
    if (gb * 256 + 256 > total) return;
    const uint64_t chunk = (gb * 256) / chunk_bytes;
    const uint64_t base = chunk * chunk_bytes;
    const uint64_t len = (total - base) < chunk_bytes ? (total - base) : chunk_bytes;
    const uint64_t bw = gb - chunk * (chunk_bytes >> 8);
    if (redo[chunk] || head_state[8 * chunk + 3] || bw < (head_state[8 * chunk + 6] >> 8) || bw >= (len / kTrip) * kAhead) return;   // (the head and a ragged end are the in-order kernel's)
    BlockMasks<ALGO> m;
    m.load(done, blocks_total, gb);
    const uint32_t q = reinterpret_cast<const uint32_t*>(in)[gb * 64 + lane];
    const uint32_t r = lane / Q, k = lane % Q;
    // the stage that settled my quad = the number of stages that had not yet (the masks are cumulative); kStages: nobody, a plain quad
    uint32_t stage = 0;
#pragma unroll
    for (uint32_t s = 0; s < G::kStages; ++s) stage += ((m.m[s] >> lane) & 1ull) ? 0u : 1u;
    // flags: cheetah.rs:17-23 (predicted 3, MAP_A 1, MAP_B 2, plain 0), lion.rs:17-27 (predictions 1..5, MAP_A 6, MAP_B 7, plain 0)
    const uint32_t flag = ALGO == DENSITY_HIP_CHEETAH ? (stage == 0 ? 3u : stage == 1 ? 1u : stage == 2 ? 2u : 0u) : (stage < G::kStages ? stage + 1u : 0u);
    const uint32_t p = (uint32_t)(m.predicted() >> (Q * r)) & rec_mask, c = (uint32_t)(m.coded() >> (Q * r)) & rec_mask;
    const uint32_t below = (1u << k) - 1u;
    // bytes of the items before mine in my record: four per plain quad, two per MAP quad
    const uint32_t before = 4u * (uint32_t)__builtin_popcount(~c & below) + 2u * (uint32_t)__builtin_popcount(c & ~p & below);
    uint8_t* rec = out + chunk * out_stride + rec_off[gb * kRecs + r];
    uint8_t* at = rec + G::kSig + before;
    const bool plain = !((c >> k) & 1u), predicted = (p >> k) & 1u;
    if (plain) st32u(at, q);                                                      // cheetah.rs:136-139, lion.rs:250-253
    else if (!predicted) st16u(at, hash16(q));                                    // cheetah.rs:132-135, lion.rs:246-249
    const uint64_t plane0 = ballot64(flag & 1u), plane1 = ballot64(flag & 2u), plane2 = ballot64(flag & 4u);
    if (k == 0) {
        const uint32_t f0 = (uint32_t)(plane0 >> (Q * r)) & rec_mask, f1 = (uint32_t)(plane1 >> (Q * r)) & rec_mask, f2 = (uint32_t)(plane2 >> (Q * r)) & rec_mask;
        if (ALGO == DENSITY_HIP_CHEETAH) {
            const uint64_t sig = spread_by2(f0) | (spread_by2(f1) << 1);
            st32u(rec, (uint32_t)sig);
            st32u(rec + 4, (uint32_t)(sig >> 32));
        } else {
            const uint64_t sig = spread_by3(f0) | (spread_by3(f1) << 1) | (spread_by3(f2) << 2);   // lion.rs:334-337: six bytes
            st32u(rec, (uint32_t)sig);
            st16u(rec + 4, (uint32_t)(sig >> 32) & 0xffffu);
        }
    }
}

template <int ALGO>
hipError_t run_stages(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                      uint64_t* d_sizes, uint8_t* d_tables, uint32_t n_slots, uint8_t* d_scratch, uint32_t* d_err, hipStream_t stream) {
    using G = StageGeo<ALGO>;
    constexpr uint32_t kRecs = 64 / G::kRecQuads;
    const uint64_t blocks = (total + 255) / 256;
    uint32_t* vals = reinterpret_cast<uint32_t*>(d_scratch);
    uint32_t* done = reinterpret_cast<uint32_t*>(d_scratch + ((total + 255) & ~255ull));   // per stage: four dwords per block
    uint32_t* rec_off = done + 4 * G::kStages * blocks;
    uint32_t* redo = rec_off + kRecs * blocks;
    uint32_t* head_state = redo + n_chunks;
    uint32_t* tail_state = head_state + 8 * (size_t)n_chunks;
    // the head of every chunk in order (a wave per chunk, its tables left in d_tables), then the passes from those tables
    hipError_t e = launch_wave_encode_heads(ALGO, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, head_state, G::kHeadBytes, G::kHeadCalm, stream);
    if (e != hipSuccess) return e;
    constexpr uint32_t W = ALGO == DENSITY_HIP_CHEETAH ? kStageWavesCheetah : kStageWaves;
    auto first = exchange_stage<true, true, ALGO != DENSITY_HIP_CHEETAH, false, W>;
    // (Cheetah has one predictor level: the level kernels are Lion's — and are not even instantiated at Cheetah's wave count)
    auto level = exchange_stage<true, false, true, true, kStageWaves>, last_level = exchange_stage<true, false, false, true, kStageWaves>;
    auto stage_a = exchange_stage<false, true, true, true, W>, stage_b = exchange_stage<false, false, false, true, W>;
    constexpr uint32_t kStageLds = stage_lds_bytes(W);
    for (const void* k : {(const void*)first, (const void*)level, (const void*)last_level, (const void*)stage_a, (const void*)stage_b}) {
        e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStageLds);
        if (e != hipSuccess) return e;
    }
    const dim3 grid(2 * n_chunks), wave(W * 64);
    const uint32_t* hs = head_state;
    const uint32_t levels = G::kStages - 2;                                        // predictor levels: 1 (Cheetah), 5 (Lion: 20 bytes a slot)
    const uint32_t pred_stride = 4 * levels;
    uint32_t s = 0;
    for (uint32_t l = 0; l < levels; ++l, ++s) {
        const StagePlace place{G::kChunkTables, (uint32_t)(65536u * 8u + 4u * l), pred_stride};
        auto kernel = l == 0 ? first : (l + 1 < levels ? level : last_level);
        hipLaunchKernelGGL(kernel, grid, wave, kStageLds, stream, d_in, total, chunk_bytes, (const uint32_t*)(done + 4 * (s ? s - 1 : 0) * blocks), done + 4 * s * blocks, vals,
                           d_tables, hs, place, d_err);
    }
    for (uint32_t ab = 0; ab < 2; ++ab, ++s) {
        const StagePlace place{G::kChunkTables, 4u * ab, 8u};
        hipLaunchKernelGGL(ab == 0 ? stage_a : stage_b, grid, wave, kStageLds, stream, d_in, total, chunk_bytes, (const uint32_t*)(done + 4 * (s - 1) * blocks), done + 4 * s * blocks,
                           vals, d_tables, hs, place, d_err);
    }
    hipLaunchKernelGGL(stage_record_layout<ALGO>, dim3(n_chunks), dim3(kLayoutThreads), 0, stream, total, chunk_bytes, (const uint32_t*)done, blocks, hs, d_in, rec_off, d_sizes,
                       redo, tail_state);
    hipLaunchKernelGGL(stage_emit_records<ALGO>, dim3((uint32_t)((blocks + kEmitWaves - 1) / kEmitWaves)), dim3(kEmitWaves * 64), 0, stream, d_in, total, chunk_bytes,
                       (const uint32_t*)done, blocks, (const uint32_t*)rec_off, (const uint32_t*)redo, hs, d_out, out_stride);
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_wave_encode_tails(ALGO, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, tail_state, stream);
    if (e != hipSuccess) return e;
    if (g_stage_audit) {                                                           // tests and profiles: how many chunks the passes kept
        std::vector<uint32_t> verdicts(n_chunks);
        e = hipMemcpyAsync(verdicts.data(), redo, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        g_stage_stats[0] += n_chunks;
        for (uint32_t v : verdicts) g_stage_stats[1] += v ? 1 : 0;
    }
    // chunks whose records met the blow-up protection behind the head after all: in order, on their own tables
    return launch_wave_encode_only(ALGO, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots, redo, stream);
}

}  // namespace

bool stage_encode_eligible(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks) {
    if (algo != DENSITY_HIP_CHEETAH && algo != DENSITY_HIP_LION) return false;
    const uint64_t head = algo == DENSITY_HIP_CHEETAH ? StageGeo<DENSITY_HIP_CHEETAH>::kHeadBytes : StageGeo<DENSITY_HIP_LION>::kHeadBytes;
    const uint64_t slot = algo == DENSITY_HIP_CHEETAH ? StageGeo<DENSITY_HIP_CHEETAH>::kChunkTables : StageGeo<DENSITY_HIP_LION>::kChunkTables;
    // The passes cost time in proportion to the input (one work-group per CU at a time: the LDS holds one half table), the one-wave kernels
    // in proportion to the CHUNK while there are CUs for more waves.  Measured on 1 GiB in 2048 chunks of 512 KiB: Cheetah 6.8 ms in passes
    // against 13.1 on the one-wave kernel, Lion (seven stages, 48 KiB heads) 29.9 against 20.3; on 100 MB in 96 chunks 0.9 / 3.4 against 11.4 / 24.2.
    // Round 4 (Lion's one-wave encoder takes four blocks per step now): 100 MB in 96 / 191 chunks 2.3 / 2.4 ms in passes against 10.6 / 5.4, in 381 chunks
    // (two rounds of work-groups per stage) 5.7 against 3.0 — Lion's passes up to one work-group per CU.
    static const uint32_t most_override = debug_env("DENSITY_HIP_STAGE_MOST") ? (uint32_t)atoi(debug_env("DENSITY_HIP_STAGE_MOST")) : 0u;   // (tuning runs)
    const uint32_t most = most_override ? most_override : algo == DENSITY_HIP_CHEETAH ? 4096u : 256u;
    const uint64_t least = 4 * head > 16 * kTrip ? 4 * head : 16 * kTrip;          // (four heads and more; 64 KiB at least, as with rounds 3-5's heads of 16 KiB)
    // (chunk bases must be whole 256-byte blocks; ONE chunk — a reference stream — may have any length: its ragged end is the in-order kernel's)
    return !g_force_lane_codec && !g_force_wave_codec && !g_rotor_unsafe && n_chunks != 0 && n_chunks <= most && (uintptr_t)d_in % 4 == 0 &&
           (n_chunks == 1 || chunk_bytes % kTrip == 0) && chunk_bytes >= least && chunk_bytes < (1ull << 31) && total >= least &&
           (uint64_t)n_chunks * slot <= (8ull << 30);                             // (a table slot per chunk: api.hip::kSerialTableBudget)
}
// vals (a dword per quad) | done masks (stages x 2 halves x a qword per 64-quad block) | record offsets | per-chunk verdicts, head and tail states
uint64_t stage_scratch_bytes(int algo, uint64_t total, uint32_t n_chunks) {
    const uint64_t blocks = (total + 255) / 256;
    const uint64_t stages = algo == DENSITY_HIP_CHEETAH ? StageGeo<DENSITY_HIP_CHEETAH>::kStages : StageGeo<DENSITY_HIP_LION>::kStages;
    const uint64_t recs = algo == DENSITY_HIP_CHEETAH ? 2 : 4;
    return ((total + 255) & ~255ull) + blocks * (stages * 2 * 8 + recs * 4) + (((uint64_t)n_chunks * (4 + 32 + 32) + 255) & ~255ull) + 256;
}

hipError_t launch_stage_encode(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                               uint64_t* d_sizes, uint8_t* d_tables, uint32_t n_slots, uint8_t* d_scratch, uint32_t* d_err, hipStream_t stream) {
    return algo == DENSITY_HIP_CHEETAH ? run_stages<DENSITY_HIP_CHEETAH>(d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots, d_scratch, d_err, stream)
                                       : run_stages<DENSITY_HIP_LION>(d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots, d_scratch, d_err, stream);
}

}  // namespace density
