// exchange_stages.hip — Cheetah ENCODING as passes of ordered LDS exchanges (gfx950).
//
// The reference walks a chunk quad by quad through three tables (cheetah.rs:123-149).  Every one of those table steps is an
// unconditional EXCHANGE once it is known which quads take part:
//
//   P   old = xchg(pred[h(q[i-1])], q[i])   every quad;             predicted <=> old == q[i]   (a hit rewrites what is there)
//   A   old = xchg(a[h(q[i])], q[i])        the quads P missed;     MAP_A     <=> old == q[i]
//   B   old = xchg(b[h(q[i])], old_A)       the quads A missed too; MAP_B     <=> old == q[i]   (cheetah.rs:140-141: b = a, a = quad)
//
// so a stage can run over the WHOLE chunk before the next one starts, and a stage is "one table, the taking-part quads in stream
// order": what ds_mskor_rtn_b32 does for 64 quads in one instruction (ascending-lane service order, verified at start-up —
// rotor.hip::rotor_selftest_kernel; with a full mask it is an exchange).  A table of 64 Ki dwords does not fit the LDS, a half does:
// one work-group per (chunk, half of the slots) streams the chunk and exchanges the quads whose slot falls in its half, the others
// exchange into a sink word of their own.  Three launches, 2 x n_chunks work-groups each, no table in global memory, no dependent
// memory round trip per record; then the record sizes are scanned per chunk and the records written by as many waves as there are
// 256-byte blocks.  (tools/exchange_stage_model.py restates this in Python; tests/test_exchange_stage_model.py checks it against the
// oracle, Lion's seven-stage form included.)
//
// What the passes cannot know is the blow-up protection (codec.rs:35-37): a raw-copy block takes its quads out of every table.
// They run as if there were none; the size scan sees whether two incompressible records ever meet (protection_state.rs:38-47), and
// such a chunk — and a ragged last chunk — is encoded by the one-wave-per-stream kernel of serial_codec.hip instead (`only` filter).
#include "common.hpp"
#include "kernels.hpp"

#include <vector>

namespace density {

extern __shared__ __attribute__((aligned(16))) uint8_t stage_lds[];
bool g_force_wave_codec = false;   // density_hip_set_kernel_variant(32)
bool g_stage_audit = false;        // density_hip_set_kernel_variant(64): count the chunks handed back (reads the verdicts: synchronises)
uint64_t g_stage_stats[2] = {0, 0}; // density_hip_stage_stats: chunks through the exchange passes | of those, handed back to the in-order kernel

namespace {

constexpr uint32_t kHalfSlots = 32768;                   // slots per work-group
constexpr uint32_t kTable = kHalfSlots * 4;              // 128 KiB of dwords
constexpr uint32_t kStageLds = kTable + 64 * 4;          // + a sink word per lane for the quads of the other half / of earlier stages
constexpr uint32_t kBatch = 8;                           // blocks of 64 quads per exchange batch (one asm statement)
constexpr uint32_t kAhead = 2 * kBatch;                  // blocks per loop trip = blocks in flight from memory
constexpr uint32_t kTrip = kAhead * 256;                    // bytes per loop trip: chunks are whole trips (a ragged one is handed back)
constexpr uint32_t kHeadBytes = 4 * kTrip;               // the in-order head of every chunk (cold dictionary: incompressible records, raw copies)
constexpr uint32_t kHeadBlocks = kHeadBytes / 256;
constexpr uint64_t kChunkTables = 65536ull * 12;         // serial_codec.hip: per chunk 64 Ki {a, b} pairs, then 64 Ki predictions
constexpr uint32_t kRec = 128;                           // cheetah.rs:188-196: 32 quads per signature

__device__ __forceinline__ uint32_t hash16(uint32_t q) { return (q * kHashMul) >> 16; }

// eight ordered exchanges back to back, answers valid at the end of the statement
#define DENSITY_STAGE_XCHG8(o, a, v, ones)                                                                                        \
    asm volatile(                                                                                                                 \
        "ds_mskor_rtn_b32 %0, %8, %24, %16\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %1, %9, %24, %17\n\t"                                                                                   \
        "ds_mskor_rtn_b32 %2, %10, %24, %18\n\t"                                                                                  \
        "ds_mskor_rtn_b32 %3, %11, %24, %19\n\t"                                                                                  \
        "ds_mskor_rtn_b32 %4, %12, %24, %20\n\t"                                                                                  \
        "ds_mskor_rtn_b32 %5, %13, %24, %21\n\t"                                                                                  \
        "ds_mskor_rtn_b32 %6, %14, %24, %22\n\t"                                                                                  \
        "ds_mskor_rtn_b32 %7, %15, %24, %23\n\t"                                                                                  \
        "s_waitcnt lgkmcnt(0)"                                                                                                    \
        : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])                  \
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),                                 \
          "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "v"(ones)                       \
        : "memory")

// One stage over one chunk for one half of the slots.
//   KEY_PREV   the slot is the hash of the quad BEFORE (P) / of the quad itself (A, B)
//   OWN_VALUE  the value exchanged in is the quad (P, A) / what the previous stage displaced (B: vals[])
//   KEEP_OLD   the displaced value is kept in vals[] for the next stage (A)
// done_prev / done_out: per 64-quad block of the whole input and per half, "quad settled by this stage or an earlier one"
// (cumulative; a reader ORs the two halves).  Quads settled earlier take no part.
template <bool KEY_PREV, bool OWN_VALUE, bool KEEP_OLD>
__global__ __launch_bounds__(64) void exchange_stage(const uint8_t* __restrict__ in, uint64_t total, uint64_t chunk_bytes,
                                                     const uint64_t* __restrict__ done_prev, uint64_t* __restrict__ done_out,
                                                     uint32_t* __restrict__ vals, uint64_t blocks_total, const uint8_t* __restrict__ tables,
                                                     const uint32_t* __restrict__ head_state) {
    const uint32_t lane = threadIdx.x;
    const uint64_t chunk = blockIdx.x >> 1;
    const uint32_t half = blockIdx.x & 1u;
    const uint64_t base = chunk * chunk_bytes;
    const uint64_t len = (total - base) < chunk_bytes ? (total - base) : chunk_bytes;
    if (head_state[8 * chunk + 3]) return;                                        // too short: the one-wave kernel takes it whole
    const uint32_t nb = (uint32_t)(len / kTrip) * kAhead;                         // whole trips; a ragged end is the in-order kernel's again
    const uint64_t gb0 = base >> 8;
    {
        // the table as the head left it (serial_codec.hip: 64 Ki {a, b} pairs, then 64 Ki predictions, per chunk): this half of the slots
        const uint8_t* mine_tb = tables + chunk * kChunkTables;
        uint32_t* w = reinterpret_cast<uint32_t*>(stage_lds);
        if (KEY_PREV) {
            const uint4* src = reinterpret_cast<const uint4*>(mine_tb + 65536ull * 8 + (uint64_t)half * kTable);
            uint4* p = reinterpret_cast<uint4*>(stage_lds);
#pragma unroll 8
            for (uint32_t i = lane; i < kTable / 16; i += 64) p[i] = src[i];
        } else {
            const uint2* src = reinterpret_cast<const uint2*>(mine_tb) + (uint64_t)half * kHalfSlots;
#pragma unroll 8
            for (uint32_t k = lane; k < kHalfSlots; k += 64) { const uint2 e = src[k]; w[k] = OWN_VALUE ? e.x : e.y; }
        }
        w[kHalfSlots + lane] = 0u;                                                // the sinks
        __syncthreads();
    }
    const uint32_t* __restrict__ q32 = reinterpret_cast<const uint32_t*>(in + base);
    uint32_t* __restrict__ v32 = vals + gb0 * 64;
    const uint64_t* __restrict__ before0 = done_prev + gb0;
    const uint64_t* __restrict__ before1 = done_prev + blocks_total + gb0;
    uint64_t* __restrict__ mine = done_out + (uint64_t)half * blocks_total + gb0;
    const uint32_t lds0 = lds_addr(stage_lds);
    const uint32_t sink = lds0 + kTable + lane * 4u;
    const uint32_t ones = 0xffffffffu;

    const uint32_t last_hash = head_state[8 * chunk + 1];                         // cheetah.rs:146 as the head left it (its last block may be a raw copy)
    uint32_t qn[kAhead], pn[kAhead], vn[kAhead];
#pragma unroll
    for (uint32_t j = 0; j < kAhead; ++j) {
        const uint32_t i = (kHeadBlocks + j) * 64u + lane;
        qn[j] = q32[i];
        pn[j] = KEY_PREV ? q32[i - 1] : 0u;
        vn[j] = !OWN_VALUE ? v32[i] : 0u;
    }
    for (uint32_t g = kHeadBlocks; g < nb; g += kAhead) {
        uint32_t q[kAhead], key[kAhead], val[kAhead];
        uint64_t before[kAhead];
#pragma unroll
        for (uint32_t j = 0; j < kAhead; ++j) {
            q[j] = qn[j];
            val[j] = OWN_VALUE ? qn[j] : vn[j];
            // slot: cheetah.rs:125 (the predictor is addressed by the previous quad's hash) / :131
            key[j] = KEY_PREV ? hash16(pn[j]) : hash16(qn[j]);
        }
        if (KEY_PREV && g == kHeadBlocks && lane == 0) key[0] = last_hash;
        {                                                                          // the next trip's quads: in flight across this one
            const uint32_t gn = g + kAhead < nb ? g + kAhead : g;                  // (the last trip loads itself again: no branch, nothing out of bounds)
#pragma unroll
            for (uint32_t j = 0; j < kAhead; ++j) {
                const uint32_t i = (gn + j) * 64u + lane;
                qn[j] = q32[i];
                if (KEY_PREV) pn[j] = q32[i - 1];
                if (!OWN_VALUE) vn[j] = v32[i];
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < kAhead; ++j) before[j] = KEY_PREV ? 0ull : (before0[g + j] | before1[g + j]);   // (P is the first stage)
#pragma unroll
        for (uint32_t s = 0; s < kAhead; s += kBatch) {
            uint32_t addr[kBatch], put[kBatch], old[kBatch];
            uint64_t hits[kBatch];
            bool part[kBatch];
#pragma unroll
            for (uint32_t j = 0; j < kBatch; ++j) {
                part[j] = !((before[s + j] >> lane) & 1ull) && (key[s + j] >> 15) == half;
                addr[j] = part[j] ? lds0 + (key[s + j] & (kHalfSlots - 1u)) * 4u : sink;
                put[j] = val[s + j];
            }
            DENSITY_STAGE_XCHG8(old, addr, put, ones);
#pragma unroll
            for (uint32_t j = 0; j < kBatch; ++j) {
                const bool hit = part[j] && old[j] == q[s + j];
                hits[j] = ballot64(hit) | before[s + j];
                if (KEEP_OLD && part[j] && !hit) v32[(g + s + j) * 64u + lane] = old[j];
            }
            if (lane == 0) {
#pragma unroll
                for (uint32_t j = 0; j < kBatch; ++j) mine[g + s + j] = hits[j];
            }
        }
    }
    if (len % kTrip) {                                                             // a ragged end follows: the table goes back where it came from
        __syncthreads();
        uint8_t* mine_tb = const_cast<uint8_t*>(tables) + chunk * kChunkTables;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(stage_lds);
        if (KEY_PREV) {
            uint4* dst = reinterpret_cast<uint4*>(mine_tb + 65536ull * 8 + (uint64_t)half * kTable);
            const uint4* p = reinterpret_cast<const uint4*>(stage_lds);
            for (uint32_t i = lane; i < kTable / 16; i += 64) dst[i] = p[i];
        } else {
            uint32_t* dst = reinterpret_cast<uint32_t*>(mine_tb) + (uint64_t)half * kHalfSlots * 2 + (OWN_VALUE ? 0 : 1);
            for (uint32_t k = lane; k < kHalfSlots; k += 64) dst[2 * k] = w[k];
        }
    }
}

// record sizes of a 64-quad block (two records) from the cumulative masks: predicted quads cost nothing, MAP_A / MAP_B two bytes,
// plain quads four (cheetah.rs:123-149), behind an 8-byte signature
struct BlockMasks { uint64_t p, a, b; };
__device__ __forceinline__ BlockMasks block_masks(const uint64_t* __restrict__ done, uint64_t blocks_total, uint64_t gb) {
    BlockMasks m;
    m.p = done[gb] | done[blocks_total + gb];
    m.a = done[2 * blocks_total + gb] | done[3 * blocks_total + gb];
    m.b = done[4 * blocks_total + gb] | done[5 * blocks_total + gb];
    return m;
}
__device__ __forceinline__ uint32_t record_bytes(const BlockMasks& m, uint32_t r) {
    const uint32_t p = (uint32_t)(m.p >> (32 * r)), b = (uint32_t)(m.b >> (32 * r));
    const uint32_t plain = 32u - (uint32_t)__builtin_popcount(b), maps = (uint32_t)__builtin_popcount(b) - (uint32_t)__builtin_popcount(p);
    return 8u + 4u * plain + 2u * maps;
}

// per chunk: record offsets (exclusive scan of the record sizes), the stream length, and whether the passes' assumption held
constexpr uint32_t kLayoutThreads = 256;
__global__ __launch_bounds__(kLayoutThreads) void stage_record_layout(uint64_t total, uint64_t chunk_bytes, const uint64_t* __restrict__ done,
                                                                      uint64_t blocks_total, const uint32_t* __restrict__ head_state,
                                                                      const uint8_t* __restrict__ in, uint32_t* __restrict__ rec_off,
                                                                      uint64_t* __restrict__ sizes, uint32_t* __restrict__ redo,
                                                                      uint32_t* __restrict__ tail_state) {
    __shared__ uint32_t s_sum[kLayoutThreads], s_first[kLayoutThreads], s_last[kLayoutThreads];
    const uint32_t t = threadIdx.x;
    const uint64_t chunk = blockIdx.x;
    const uint64_t base = chunk * chunk_bytes;
    const uint64_t len = (total - base) < chunk_bytes ? (total - base) : chunk_bytes;
    if (head_state[8 * chunk + 3]) { if (t == 0) { redo[chunk] = 1u; tail_state[8 * chunk + 6] = 0u; } return; }   // (uniform)
    const uint32_t nb = (uint32_t)(len / kTrip) * kAhead;
    const uint64_t gb0 = base >> 8;
    const uint32_t rest = nb - kHeadBlocks;
    const uint32_t per = (rest + kLayoutThreads - 1) / kLayoutThreads;
    const uint32_t b0 = kHeadBlocks + (t * per < rest ? t * per : rest), b1 = b0 + per < nb ? b0 + per : nb;
    const uint32_t head_fsm = head_state[8 * chunk + 2];
    uint32_t sum = 0, first = 0, last = 0, pair = 0;
    for (uint32_t b = b0; b < b1; ++b) {
        const BlockMasks m = block_masks(done, blocks_total, gb0 + b);
        for (uint32_t r = 0; r < 2; ++r) {
            const uint32_t bytes = record_bytes(m, r);
            const uint32_t inc = bytes >= kRec ? 1u : 0u;                         // codec.rs:68
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
        const BlockMasks m = block_masks(done, blocks_total, gb0 + b);
        for (uint32_t r = 0; r < 2; ++r) {
            rec_off[(gb0 + b) * 2 + r] = off;
            off += record_bytes(m, r);
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
            for (uint32_t r = 2 * kHeadBlocks; r < 2 * nb; ++r) {
                if (start == 1) { counter += 2 * nb - r; break; }                  // (only the counter's low four bits matter from here on)
                if ((counter & 0xfu) == 0) start >>= 1;
                ++counter;
            }
            // (thread 255 owns the last blocks, or none: the last record's verdict is the last one anybody saw)
            uint32_t prev = last;
            if (b0 >= b1) for (uint32_t i = kLayoutThreads - 1; i-- > 0;) if (s_sum[i]) { prev = s_last[i]; break; }
            ts[0] = nb * 256u;
            ts[1] = off;
            ts[2] = hash16(reinterpret_cast<const uint32_t*>(in + base)[nb * 64u - 1]);   // cheetah.rs:146
            ts[3] = prev;
            ts[4] = start;
            ts[5] = counter;
        }
    }
}

// 2-bit flags of 32 quads -> the 64-bit signature (io/write_signature.rs:14-17: quad k at bits 2k, 2k+1)
__device__ __forceinline__ uint64_t spread_bits(uint32_t x) {
    uint64_t v = x;
    v = (v | (v << 16)) & 0x0000ffff0000ffffull;
    v = (v | (v << 8)) & 0x00ff00ff00ff00ffull;
    v = (v | (v << 4)) & 0x0f0f0f0f0f0f0f0full;
    v = (v | (v << 2)) & 0x3333333333333333ull;
    v = (v | (v << 1)) & 0x5555555555555555ull;
    return v;
}

// the records: one wave per 64-quad block (two records), a quad per lane
constexpr uint32_t kEmitWaves = 4;
__global__ __launch_bounds__(kEmitWaves * 64) void stage_emit_records(const uint8_t* __restrict__ in, uint64_t total, uint64_t chunk_bytes,
                                                                       const uint64_t* __restrict__ done, uint64_t blocks_total,
                                                                       const uint32_t* __restrict__ rec_off, const uint32_t* __restrict__ redo,
                                                                       uint8_t* __restrict__ out, uint64_t out_stride) {
    // (the records of a chunk's first kHeadBlocks are the in-order kernel's)
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t gb = (uint64_t)blockIdx.x * kEmitWaves + (threadIdx.x >> 6);
    if (gb * 256 + 256 > total) return;                                           // (a ragged end belongs to a chunk that is done again)
    const uint64_t chunk = (gb * 256) / chunk_bytes;
    const uint64_t base = chunk * chunk_bytes;
    const uint64_t len = (total - base) < chunk_bytes ? (total - base) : chunk_bytes;
    const uint64_t bw = gb - chunk * (chunk_bytes >> 8);
    if (redo[chunk] || bw < kHeadBlocks || bw >= (len / kTrip) * kAhead) return;  // (the head and a ragged end are the in-order kernel's)
    const BlockMasks m = block_masks(done, blocks_total, gb);
    const uint32_t q = reinterpret_cast<const uint32_t*>(in)[gb * 64 + lane];
    const uint32_t r = lane >> 5, k = lane & 31u;
    const uint32_t p = (uint32_t)(m.p >> (32 * r)), a = (uint32_t)(m.a >> (32 * r)), b = (uint32_t)(m.b >> (32 * r));
    const uint32_t below = (1u << k) - 1u;
    // bytes of the items before mine in my record: four per plain quad, two per MAP quad
    const uint32_t before = 4u * (uint32_t)__builtin_popcount(~b & below) + 2u * (uint32_t)__builtin_popcount(b & ~p & below);
    uint8_t* rec = out + chunk * out_stride + rec_off[gb * 2 + r];
    uint8_t* at = rec + 8u + before;
    const bool plain = !((b >> k) & 1u), predicted = (p >> k) & 1u;
    if (plain) st32u(at, q);                                                      // cheetah.rs:136-139
    else if (!predicted) st16u(at, hash16(q));                                    // :132-135
    if (k == 0) {
        // flags (cheetah.rs:17-23): predicted 3, MAP_A 1, MAP_B 2, plain 0 -> bit 0 = predicted | MAP_A, bit 1 = predicted | MAP_B
        const uint32_t bit0 = p | (a & ~p), bit1 = p | (b & ~a);
        const uint64_t sig = spread_bits(bit0) | (spread_bits(bit1) << 1);
        st32u(rec, (uint32_t)sig);
        st32u(rec + 4, (uint32_t)(sig >> 32));
    }
}

}  // namespace

bool stage_encode_eligible(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks) {
    return algo == DENSITY_HIP_CHEETAH && !g_force_lane_codec && !g_force_wave_codec && !g_exchange_unsafe && n_chunks != 0 &&
           (uintptr_t)d_in % 4 == 0 && chunk_bytes % kTrip == 0 && chunk_bytes >= 4 * kHeadBytes && chunk_bytes < (1ull << 31) &&
           (uint64_t)n_chunks * kChunkTables <= (8ull << 30);                     // (a table slot per chunk: api.hip::kSerialTableBudget)
}
// vals (a dword per quad) | done masks (3 stages x 2 halves x a qword per 64-quad block) | record offsets | per-chunk verdicts
uint64_t stage_scratch_bytes(uint64_t total, uint32_t n_chunks) {
    const uint64_t blocks = (total + 255) / 256;
    return ((total + 255) & ~255ull) + blocks * (6 * 8 + 2 * 4) + (((uint64_t)n_chunks * (4 + 32 + 32) + 255) & ~255ull) + 256;
}

hipError_t launch_stage_encode(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                               uint64_t* d_sizes, uint8_t* d_tables, uint32_t n_slots, uint8_t* d_scratch, hipStream_t stream) {
    const uint64_t blocks = (total + 255) / 256;
    uint32_t* vals = reinterpret_cast<uint32_t*>(d_scratch);
    uint64_t* done = reinterpret_cast<uint64_t*>(d_scratch + ((total + 255) & ~255ull));
    uint32_t* rec_off = reinterpret_cast<uint32_t*>(done + 6 * blocks);
    uint32_t* redo = rec_off + 2 * blocks;
    uint32_t* head_state = redo + n_chunks;
    uint32_t* tail_state = head_state + 8 * (size_t)n_chunks;
    auto stage_p = exchange_stage<true, true, false>, stage_a = exchange_stage<false, true, true>, stage_b = exchange_stage<false, false, false>;
    hipError_t e = hipFuncSetAttribute((const void*)stage_p, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStageLds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stage_a, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStageLds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stage_b, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kStageLds);
    if (e != hipSuccess) return e;
    // the head of every chunk in order (a wave per chunk, its tables left in d_tables), then the passes from those tables
    e = launch_cheetah_encode_heads(d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_tables, head_state, kHeadBytes, stream);
    if (e != hipSuccess) return e;
    const dim3 grid(2 * n_chunks), wave(64);
    const uint8_t* tb = d_tables;
    const uint32_t* hs = head_state;
    hipLaunchKernelGGL(stage_p, grid, wave, kStageLds, stream, d_in, total, chunk_bytes, (const uint64_t*)nullptr, done, vals, blocks, tb, hs);
    hipLaunchKernelGGL(stage_a, grid, wave, kStageLds, stream, d_in, total, chunk_bytes, (const uint64_t*)done, done + 2 * blocks, vals, blocks, tb, hs);
    hipLaunchKernelGGL(stage_b, grid, wave, kStageLds, stream, d_in, total, chunk_bytes, (const uint64_t*)(done + 2 * blocks), done + 4 * blocks, vals, blocks, tb, hs);
    hipLaunchKernelGGL(stage_record_layout, dim3(n_chunks), dim3(kLayoutThreads), 0, stream, total, chunk_bytes, (const uint64_t*)done, blocks, hs, d_in, rec_off, d_sizes, redo, tail_state);
    hipLaunchKernelGGL(stage_emit_records, dim3((uint32_t)((blocks + kEmitWaves - 1) / kEmitWaves)), dim3(kEmitWaves * 64), 0, stream, d_in, total, chunk_bytes,
                       (const uint64_t*)done, blocks, (const uint32_t*)rec_off, (const uint32_t*)redo, d_out, out_stride);
    e = hipGetLastError();
    if (e == hipSuccess) e = launch_cheetah_encode_tails(d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, tail_state, stream);
    if (e != hipSuccess) return e;
    if (g_stage_audit) {                                                           // tests and profiles: how many chunks the passes kept
        std::vector<uint32_t> verdicts(n_chunks);
        e = hipMemcpyAsync(verdicts.data(), redo, (size_t)n_chunks * 4, hipMemcpyDeviceToHost, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        g_stage_stats[0] += n_chunks;
        for (uint32_t v : verdicts) g_stage_stats[1] += v ? 1 : 0;
    }
    // chunks whose records met the blow-up protection, and a ragged last chunk: in order, on their own tables
    return launch_cheetah_encode_only(d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots, redo, stream);
}

}  // namespace density
