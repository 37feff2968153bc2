// stream_parse.hip — record boundaries of ONE Chameleon reference stream, found in parallel (api.hip::run_stream_decode_segmented).
//
// A reference stream has no framing: record k+1 starts where record k ends, and a record's length is in its signature
// (8 + 256 - 2 x MAP flags: codec.rs:28-31,92-99).  For a CALM stream — no raw-copy block, i.e. never two incompressible records in a
// row (protection_state.rs:38-47) — every 2-byte offset p is a candidate record start with a known successor
// p + 264 - 2*popcount(signature at p), whether or not a record really starts there.  So:
//   head      one lane walks the first records with the real FSM until it is calm (a stream's first blocks are incompressible) -> p0, b0;
//   windows   per window of 8 KiB behind p0, backwards over its 4096 candidates, 64 at a time (a record is 136..264 bytes, so 64
//             consecutive candidates never depend on each other): "from candidate c the chain leaves the window at offset x of the
//             next one after n records" — kept for the only possible entries, the first 132 candidates;
//   groups    the tables of 256 consecutive windows composed (one lane per entry), then the groups in sequence from entry 0,
//             then every window's entry and first block number;
//   emit      one lane per window walks forward from its entry: MAP count of every block into the block index, stream offset of
//             every 16384th block (the decoder's chunks), the end of the whole blocks;
//   check     the first pair of incompressible records in a row behind the head, if any: the parse is final up to there, and the caller
//             starts over from that block (the head walk takes the raw copies that follow with the real FSM) — or gives up.
// A numpy prototype of exactly this (tools/parse_prototype.py) is checked against the FSM walk of the oracle's streams.
#include "common.hpp"
#include "chameleon_dev.hpp"
#include "kernels.hpp"

namespace density {

namespace {

constexpr uint32_t kWin = 8192, kCand = kWin / 2, kEntries = 132, kEnd = 254, kGroup = 256, kGroupSmall = 64;   // (8 KiB windows: 8 KiB of LDS each, 20 sweeps per CU in flight; 16 KiB and 4 KiB windows measured slower)
constexpr uint32_t kHeadMaxBlocks = 4096;

__device__ __forceinline__ uint64_t ld64u(const uint8_t* p) { return *reinterpret_cast<const u64_u*>(p); }

// info words: 0 status (1 = calm head found), 1 b0, 2-3 p0, 4 total whole blocks, 5-6 end of the whole blocks (stream offset),
// 7 the first block of a pair of incompressible records behind the head (0xffffffff: none — the stream is calm from the head on),
// 8 / 9-10 in: block and stream offset the head walk starts at (everything before is final and calm)
__global__ void parse_head_kernel(const uint8_t* __restrict__ in, uint64_t E, uint8_t* __restrict__ index, uint64_t index_cap, uint32_t* __restrict__ info,
                                  uint32_t* __restrict__ pos32, uint64_t* __restrict__ chunk_offset, uint32_t chunk_blocks) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    Guard g;                                                                      // calm: no penalty, start 1 ...
    uint64_t pos = ((uint64_t)info[10] << 32) | info[9];
    uint32_t b = info[8], status = 0;
    g.counter = b & 15u;                                                          // ... and the counter counts every block of the stream (protection_state.rs:20)
    const uint32_t b_first = b;
    while (pos < E && b - b_first < kHeadMaxBlocks && b < index_cap) {
        pos32[b] = (uint32_t)pos;
        if (b % chunk_blocks == 0) chunk_offset[b / chunk_blocks] = pos;
        if (g.block_is_copy()) {                                                  // codec.rs:89-91
            if (pos + kBlock > E) break;                                          // (a raw tail: the sequential path's business)
            index[b] = (uint8_t)kIdxCopy;
            pos += kBlock;
            g.decay();
        } else {
            if (pos + kSig > E) break;
            const uint32_t pc = (uint32_t)__builtin_popcountll(ld64u(in + pos));
            const uint32_t rl = kSig + kBlock - 2u * pc;
            if (pos + rl > E) break;
            index[b] = (uint8_t)pc;
            g.update(rl >= kBlock);                                               // codec.rs:98
            pos += rl;
        }
        ++b;
        if (g.penalty == 0 && g.prev == 0 && g.start == 1 && b - b_first >= 2) { status = 1; break; }
    }
    info[0] = status; info[1] = b; info[2] = (uint32_t)pos; info[3] = (uint32_t)(pos >> 32);
    info[4] = b; info[5] = (uint32_t)pos; info[6] = (uint32_t)(pos >> 32); info[7] = 0xffffffffu;
}

__global__ __launch_bounds__(64) void parse_windows_kernel(const uint8_t* __restrict__ in, uint64_t E, const uint32_t* __restrict__ info,
                                                           uint8_t* __restrict__ T, uint8_t* __restrict__ C) {
    __shared__ uint16_t tab[kCand];                                              // per candidate: exit (low byte), records up to it (high byte)
    const uint32_t lane = threadIdx.x;
    const uint64_t p0 = ((uint64_t)info[3] << 32) | info[2];
    const uint64_t ws = p0 + (uint64_t)blockIdx.x * kWin;
    if (info[0] == 0) return;
    if (ws >= E) {                                                                // behind the stream: every entry ends at once
        for (uint32_t e = lane; e < kEntries; e += 64) { T[(uint64_t)blockIdx.x * kEntries + e] = (uint8_t)kEnd; C[(uint64_t)blockIdx.x * kEntries + e] = 0; }
        return;
    }
    // eight groups of signatures in flight ahead of the sweep (the loads are independent of the tables; the sweep is not)
    constexpr int kAhead = 8;
    uint64_t sig[kAhead], nsig[kAhead];
    auto fetch = [&](int gi) -> uint64_t {
        if (gi < 0) return 0ull;
        const uint64_t p = ws + 2ull * (64u * (uint32_t)gi + lane);
        return p + kSig <= E ? ld64u(in + p) : 0ull;
    };
#pragma unroll
    for (int k = 0; k < kAhead; ++k) sig[k] = fetch((int)(kCand / 64) - 1 - k);
    for (int g0 = (int)(kCand / 64) - 1; g0 >= 0; g0 -= kAhead) {
#pragma unroll
        for (int k = 0; k < kAhead; ++k) nsig[k] = fetch(g0 - kAhead - k);
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int gi = g0 - k;
            const uint32_t c = 64u * (uint32_t)gi + lane;
            const uint64_t p = ws + 2ull * c;
            const uint32_t pc = (uint32_t)__builtin_popcountll(sig[k]);
            const bool whole = p + kSig <= E && p + (kSig + kBlock - 2u * pc) <= E;
            const uint32_t nxt = c + 132u - pc;                                   // in candidates: 68 .. 132 ahead
            uint32_t e, n;
            if (nxt >= kCand) { e = nxt - kCand; n = 1; }
            else { const uint32_t t = tab[nxt]; e = t & 0xffu; n = 1u + (t >> 8); }
            tab[c] = whole ? (uint16_t)(e | (n << 8)) : (uint16_t)kEnd;
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < kAhead; ++k) sig[k] = nsig[k];
    }
    for (uint32_t e = lane; e < kEntries; e += 64) { T[(uint64_t)blockIdx.x * kEntries + e] = (uint8_t)(tab[e] & 0xffu); C[(uint64_t)blockIdx.x * kEntries + e] = (uint8_t)(tab[e] >> 8); }
}

// composition of the windows of one group: entry e of its first window -> (entry of the next group's first window, records)
__global__ __launch_bounds__(64) void parse_groups_kernel(const uint8_t* __restrict__ T, const uint8_t* __restrict__ C, uint32_t n_windows, uint32_t group,
                                                          uint8_t* __restrict__ GT, uint32_t* __restrict__ GC) {
    extern __shared__ uint8_t gsm[];
    uint8_t* Ts = gsm;
    uint8_t* Cs = gsm + group * kEntries;
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    const uint32_t w0 = g * group, nw = n_windows - w0 < group ? n_windows - w0 : group;
    for (uint32_t i = lane; i < nw * kEntries; i += 64) { Ts[i] = T[(uint64_t)w0 * kEntries + i]; Cs[i] = C[(uint64_t)w0 * kEntries + i]; }
    __syncthreads();
    for (uint32_t e = lane; e < kEntries; e += 64) {
        uint32_t x = e, n = 0;
        for (uint32_t w = 0; w < nw && x != kEnd; ++w) { n += Cs[w * kEntries + x]; x = Ts[w * kEntries + x]; }
        GT[g * kEntries + e] = (uint8_t)x;
        GC[g * kEntries + e] = n;
    }
}

// the groups in sequence from entry 0 of the first window (one lane), then nothing else needs an order
__global__ void parse_top_kernel(const uint8_t* __restrict__ GT, const uint32_t* __restrict__ GC, uint32_t n_groups, uint32_t* __restrict__ info,
                                 uint8_t* __restrict__ gent, uint32_t* __restrict__ gbase) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (info[0] == 0) return;
    uint32_t x = 0, base = info[1];
    for (uint32_t g = 0; g < n_groups; ++g) {
        gent[g] = (uint8_t)x; gbase[g] = base;
        if (x != kEnd) { base += GC[g * kEntries + x]; x = GT[g * kEntries + x]; }
    }
    info[4] = base;                                                               // whole blocks of the stream
}

__global__ __launch_bounds__(64) void parse_entries_kernel(const uint8_t* __restrict__ T, const uint8_t* __restrict__ C, uint32_t n_windows, uint32_t group,
                                                           const uint8_t* __restrict__ gent, const uint32_t* __restrict__ gbase,
                                                           uint8_t* __restrict__ went, uint32_t* __restrict__ wbase) {
    extern __shared__ uint8_t gsm[];
    uint8_t* Ts = gsm;
    uint8_t* Cs = gsm + group * kEntries;
    const uint32_t lane = threadIdx.x, g = blockIdx.x;
    const uint32_t w0 = g * group, nw = n_windows - w0 < group ? n_windows - w0 : group;
    for (uint32_t i = lane; i < nw * kEntries; i += 64) { Ts[i] = T[(uint64_t)w0 * kEntries + i]; Cs[i] = C[(uint64_t)w0 * kEntries + i]; }
    __syncthreads();
    if (lane == 0) {
        uint32_t x = gent[g], base = gbase[g];
        for (uint32_t w = 0; w < nw; ++w) {
            went[w0 + w] = (uint8_t)x; wbase[w0 + w] = base;
            if (x != kEnd) { base += Cs[w * kEntries + x]; x = Ts[w * kEntries + x]; }
        }
    }
}

// one wave per window: forward from the window's entry — block index, chunk offsets, the end of the whole blocks.  The window's bytes go to LDS in one
// sweep and lane 0 walks them there (a step is an LDS read, not a trip to memory: the walk of ~50 records was the longest kernel of the parse — and
// the pipelined host call pays the parse's latency once per slice)
__global__ __launch_bounds__(64) void parse_emit_kernel(const uint8_t* __restrict__ in, uint64_t E, uint32_t* __restrict__ info,
                                                        const uint8_t* __restrict__ went, const uint32_t* __restrict__ wbase, uint32_t n_windows,
                                                        uint8_t* __restrict__ index, uint64_t index_cap, uint64_t* __restrict__ chunk_offset, uint32_t chunk_blocks,
                                                        uint32_t* __restrict__ pos32) {
    typedef uint4 uint4_u __attribute__((aligned(1)));
    __shared__ __attribute__((aligned(16))) uint8_t win[kWin + 16];
    const uint32_t w = blockIdx.x, lane = threadIdx.x;
    if (w >= n_windows || info[0] == 0) return;
    uint32_t c = went[w];
    if (c == kEnd) return;
    const uint64_t p0 = ((uint64_t)info[3] << 32) | info[2];
    const uint64_t ws = p0 + (uint64_t)w * kWin;
    for (uint32_t v = lane; v < (kWin + 16) / 16; v += 64) {
        const uint64_t p = ws + 16ull * v;
        uint4 x = make_uint4(0, 0, 0, 0);
        if (p + 16 <= E) { const uint4_u* sp = reinterpret_cast<const uint4_u*>(in + p); x = make_uint4(sp->x, sp->y, sp->z, sp->w); }
        else if (p < E) {
            uint8_t t[16] = {};
            for (uint32_t i = 0; i < 16 && p + i < E; ++i) t[i] = in[p + i];
            x = *reinterpret_cast<const uint4*>(t);
        }
        *reinterpret_cast<uint4*>(win + 16u * v) = x;
    }
    __syncthreads();
    if (lane != 0) return;
    uint32_t b = wbase[w];
    while (c < kCand) {
        const uint64_t p = ws + 2ull * c;
        uint32_t pc = 0;
        bool whole = p + kSig <= E;
        if (whole) {
            const uint16_t* h = reinterpret_cast<const uint16_t*>(win + 2u * c);
            pc = (uint32_t)__builtin_popcount((uint32_t)h[0] | ((uint32_t)h[1] << 16)) + (uint32_t)__builtin_popcount((uint32_t)h[2] | ((uint32_t)h[3] << 16));
            whole = p + (kSig + kBlock - 2u * pc) <= E;
        }
        if (!whole || b >= index_cap) {                                           // the ragged end (or nothing at all) starts here
            info[5] = (uint32_t)p; info[6] = (uint32_t)(p >> 32);
            return;
        }
        index[b] = (uint8_t)pc;
        pos32[b] = (uint32_t)p;
        if (b % chunk_blocks == 0) chunk_offset[b / chunk_blocks] = p;
        ++b;
        c += 132u - pc;
    }
}

// two incompressible records in a row (a record of 256 bytes or more: at most 4 MAP flags): the FSM would have answered with raw copies
__global__ __launch_bounds__(256) void parse_check_kernel(const uint8_t* __restrict__ index, uint64_t index_cap, uint32_t* __restrict__ info) {
    // (info[4] is a count of blocks the STREAM claims: an untrusted stream may claim more than the index holds — the host then sends the call
    // down the sequential path — so it is clamped here)
    const uint32_t total = info[4] < index_cap ? info[4] : (uint32_t)index_cap, b0 = info[1];
    if (info[0] == 0) return;
    const uint32_t from = b0 ? b0 - 1 : 0;
    for (uint64_t i = from + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i + 1 < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t a = index[i], b = index[i + 1];
        if (!(a & kIdxCopy) && !(b & kIdxCopy) && a <= 4 && b <= 4) atomicMin(info + 7, (uint32_t)i);
    }
}

// where the last whole block starts (info[11]): a caller that parses a stream piece by piece resumes ONE block back, so that the head walk's fresh
// FSM sees the record in front of the new piece (a pair of incompressible records across the seam must not be missed: the check above looks at pairs
// from the head's end on, the head at pairs from its first block on)
__global__ void parse_tail_kernel(uint32_t* __restrict__ info, const uint32_t* __restrict__ pos32, uint64_t index_cap) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const uint32_t whole = info[4];
    info[11] = (info[0] != 0 && whole > 0 && whole <= index_cap) ? pos32[whole - 1] : 0u;
}

// offsets and sizes of segments [first, first + count) of a parsed stream: segment k starts at chunk_offset[k] (the stream's first at 0); the
// last one of the range ends at `end` where that is given (the stream's end; a piece's parse that stopped on a segment boundary), else where its successor starts
__global__ void seg_layout_kernel(const uint64_t* __restrict__ chunk_offset, uint32_t first, uint32_t count, uint64_t end, uint64_t* __restrict__ offsets,
                                  uint64_t* __restrict__ sizes, uint32_t* __restrict__ err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t k = first + i;
    const uint64_t off = k ? chunk_offset[k] : 0ull;
    const uint64_t next = (i + 1 < count || end == ~0ull) ? chunk_offset[k + 1] : end;
    if (next <= off) { atomicOr(err, 1u); offsets[k] = off; sizes[k] = 0; return; }   // (cannot happen; never hand the kernels a broken layout)
    offsets[k] = off; sizes[k] = next - off;
}

}  // namespace

hipError_t launch_seg_layout(const uint64_t* d_chunk_offset, uint32_t first, uint32_t count, uint64_t end, uint64_t* d_offsets, uint64_t* d_sizes, uint32_t* d_err, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(seg_layout_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, d_chunk_offset, first, count, end, d_offsets, d_sizes, d_err);
    return hipGetLastError();
}

uint64_t stream_parse_workspace(uint64_t E) {
    const uint64_t nw = E / kWin + 2, ng = (nw + kGroupSmall - 1) / kGroupSmall;   // (sized for the finer grouping)
    return nw * kEntries * 2 + ng * kEntries * 5 + ng * 5 + nw * 5 + 4096;
}

hipError_t launch_stream_parse(const uint8_t* d_in, uint64_t E, uint64_t from_pos, uint8_t* d_ws, uint8_t* d_index, uint64_t index_cap, uint64_t* d_chunk_offset,
                               uint32_t chunk_blocks, uint32_t* d_pos32, uint32_t* d_info, hipStream_t stream) {
    const uint32_t nw = (uint32_t)((E - from_pos) / kWin + 2);                       // (windows behind the head, which starts at or behind from_pos)
    // groups of 256 windows for long stretches (the groups are walked in sequence by one lane: few of them), of 64 for short ones — a piece of a
    // stream that is parsed slice by slice: the walk through a group's windows is what the two group kernels' time is made of
    const uint32_t group = nw > 64u * kGroupSmall ? kGroup : kGroupSmall, ng = (nw + group - 1) / group;
    uint8_t* T = d_ws;
    uint8_t* C = T + (uint64_t)nw * kEntries;
    uint32_t* GC = reinterpret_cast<uint32_t*>((reinterpret_cast<uintptr_t>(C + (uint64_t)nw * kEntries) + 15) & ~(uintptr_t)15);
    uint32_t* gbase = GC + (uint64_t)ng * kEntries;
    uint32_t* wbase = gbase + ng;
    uint8_t* GT = reinterpret_cast<uint8_t*>(wbase + nw);
    uint8_t* gent = GT + (uint64_t)ng * kEntries;
    uint8_t* went = gent + ng;
    const size_t group_lds = (size_t)group * kEntries * 2;
    hipError_t e = hipFuncSetAttribute((const void*)parse_groups_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)group_lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)parse_entries_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)group_lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(parse_head_kernel, dim3(1), dim3(64), 0, stream, d_in, E, d_index, index_cap, d_info, d_pos32, d_chunk_offset, chunk_blocks);
    hipLaunchKernelGGL(parse_windows_kernel, dim3(nw), dim3(64), 0, stream, d_in, E, d_info, T, C);
    hipLaunchKernelGGL(parse_groups_kernel, dim3(ng), dim3(64), group_lds, stream, T, C, nw, group, GT, GC);
    hipLaunchKernelGGL(parse_top_kernel, dim3(1), dim3(64), 0, stream, GT, GC, ng, d_info, gent, gbase);
    hipLaunchKernelGGL(parse_entries_kernel, dim3(ng), dim3(64), group_lds, stream, T, C, nw, group, gent, gbase, went, wbase);
    hipLaunchKernelGGL(parse_emit_kernel, dim3(nw), dim3(64), 0, stream, d_in, E, d_info, went, wbase, nw, d_index, index_cap, d_chunk_offset, chunk_blocks, d_pos32);
    hipLaunchKernelGGL(parse_check_kernel, dim3(256), dim3(256), 0, stream, d_index, index_cap, d_info);
    hipLaunchKernelGGL(parse_tail_kernel, dim3(1), dim3(64), 0, stream, d_info, d_pos32, index_cap);
    return hipGetLastError();
}

}  // namespace density
