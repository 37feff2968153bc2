// api_stream.hip — ONE reference stream (include/density_hip.h section 1: the reference's nine symbols, chameleon.rs:70-83, cheetah.rs:105-118,
// lion.rs:193-206, and their device-pointer forms): short streams on one work-group / wave, long Chameleon streams encoded and decoded in
// parallel segments, byte for byte the reference's stream.
#include "api_internal.hpp"

#include <chrono>

namespace density {
namespace api {
namespace {

// ---- whole-stream-exact Chameleon encode of ONE long stream, in parallel ----
// The stream is cut into segments of whole rounds.  What a segment needs from its predecessors is the dictionary as they leave it
// and the FSM state.  Speculation: no predecessor but the first has a raw-copy block (so each wrote every one of its quads, and
// "the dictionary after segments 1..k-1" is the first segment's real final dictionary overlaid with their LAST WRITERS per slot, which
// need no encoding to find) and every segment ends calm.  One pass: segment `first` for real (exact start) | last writers of the others
// in parallel -> start images by a per-slot merge -> all other segments in parallel from their start images, each reporting its
// raw-copy blocks and final FSM state.  The longest prefix whose assumptions held is final; the rest is encoded again from the exact
// final dictionary of that prefix (incompressible input degenerates to the sequential encode: once a pass after the third makes fewer
// than 8 segments final, the remainder runs as one chunk).  The output is the segments' streams concatenated byte for byte: identical to the reference's single stream.
constexpr size_t kSegMinStream = 4u << 20;
inline size_t seg_bytes_for(size_t n) {
    size_t c = (n / 256) & ~(size_t)4095;                                         // about one segment per CU, whole rounds of 16 blocks
    if (c > (4u << 20)) c = 4u << 20;
    if (c < (128u << 10)) c = 128u << 10;
    return c;
}
constexpr uint32_t kCalm = 0x80000000u;                                            // pack_guard({0, 1, 0, 0}) = 0, speculation allowed

// the geometry and the scratch of one segmented encode (the context's `seg` buffer)
struct SegEncode {
    size_t n = 0, C = 0, C0 = 0, S = 0, stride = 0;
    uint8_t *d_stage = nullptr, *d_lw = nullptr, *d_start = nullptr, *d_final = nullptr;
    uint64_t *d_sizes = nullptr, *d_offsets = nullptr, *d_carry = nullptr;       // d_carry: the running end of the stream (the pipelined host call gathers as it goes)
    uint32_t *d_gspec = nullptr, *d_gfinal = nullptr, *d_raw = nullptr, *d_err = nullptr;
    size_t seg_at(size_t k) const { return k == 0 ? 0 : C0 + (k - 1) * C; }       // where segment k starts
    size_t seg_len(size_t k) const { const size_t left = n - seg_at(k), len = k == 0 ? C0 : C; return left < len ? left : len; }
    // the first segment runs alone, ahead of everything else: a quarter of the others' length (whole rounds)
    hipError_t setup(DeviceCtx* c, size_t n_, hipStream_t s) {
        n = n_; C = seg_bytes_for(n); C0 = ((C / 4) + 4095) & ~(size_t)4095; S = 1 + (n - C0 + C - 1) / C; stride = slot_stride(DENSITY_HIP_CHAMELEON, C);
        const size_t img = kSegImageBytes;
        const size_t off_lw = align_up(S * stride, kAlign), off_start = off_lw + S * img, off_final = off_start + S * img, off_small = off_final + S * img;
        hipError_t e = c->seg.ensure(off_small + S * 64 + kAlign);
        if (e != hipSuccess) return e;
        uint8_t* base = (uint8_t*)c->seg.p;
        d_stage = base; d_lw = base + off_lw; d_start = base + off_start; d_final = base + off_final;
        d_sizes = reinterpret_cast<uint64_t*>(base + off_small);
        d_offsets = d_sizes + S;
        d_carry = d_offsets + S;
        d_gspec = reinterpret_cast<uint32_t*>(d_carry + 1);                        // start FSM states of the speculating segments
        d_gfinal = d_gspec + S;
        d_raw = d_gfinal + S;
        d_err = d_raw + S;
        const std::vector<uint32_t> calm(S, kCalm);
        e = hipMemsetAsync(d_raw, 0, (S + 1) * sizeof(uint32_t), s);              // raw counters + error word
        if (e == hipSuccess) e = hipMemsetAsync(d_carry, 0, sizeof(uint64_t), s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_gspec, calm.data(), S * sizeof(uint32_t), hipMemcpyHostToDevice, s);   // (pageable: copied before the call returns)
        return e;
    }
    // segments [a, a + count) from their start images (d_start) and start states (d_gspec), reporting final images, states and raw-copy blocks
    hipError_t speculate(const uint8_t* d_in, size_t a, size_t count, hipStream_t s) const {
        SegArgs b;
        b.init_images = d_start + a * kSegImageBytes;
        b.init_guard = d_gspec + a;
        b.final_images = d_final + a * kSegImageBytes;
        b.final_guard = d_gfinal + a;
        b.raw_blocks = d_raw + a;
        const size_t left = n - seg_at(a);
        return launch_rotor_encode_seg(d_in + seg_at(a), left < count * C ? left : count * C, C, (uint32_t)count, d_stage + a * stride, stride, d_sizes + a, d_err, b, s);
    }
};

// The passes of the segmented encode from segment `first` on, whose predecessor (if any) is final — d_final / d_gfinal of first - 1 are exact — with the
// last writers of every inner segment in d_lw.  Returns the number of slots in use through *used (a remainder encoded as one chunk sits in the slot of
// its first segment); the sizes of slots first .. used - 1 are left in h_sizes.
hipError_t seg_encode_passes(const SegEncode& L, const uint8_t* d_in, size_t first, hipStream_t s, std::vector<uint64_t>& h_sizes, size_t* used, bool trace, uint32_t* h_err = nullptr) {
    const size_t S = L.S, img = kSegImageBytes;
    bool sizes_are_current = false;
    std::vector<uint32_t> h_gfinal(S), h_raw(S);
    hipError_t e = hipSuccess;
    size_t advanced = S;                                                           // segments the previous pass made final
    const size_t first_in = first;
    for (int pass = 0; e == hipSuccess && first < S; ++pass) {
        // (a pass that gets nowhere — raw copies all over — is not repeated for long: the remainder then runs as one chunk)
        const bool rest_as_one = pass >= 16 || (pass >= 3 && advanced < 8);
        // segment `first` (or, after too many passes, everything that is left as one chunk) from its exact start
        SegArgs a;
        a.init_images = first ? L.d_final + (first - 1) * img : nullptr;
        a.init_guard = first ? L.d_gfinal + (first - 1) : nullptr;
        a.final_images = L.d_final + first * img;
        a.final_guard = L.d_gfinal + first;
        a.raw_blocks = L.d_raw + first;
        const size_t left = L.n - L.seg_at(first), len1 = first == 0 ? L.C0 : L.C;
        e = hipMemsetAsync(L.d_raw + first, 0, sizeof(uint32_t), s);
        if (e == hipSuccess)
            e = launch_rotor_encode_seg(d_in + L.seg_at(first), rest_as_one ? left : (left < len1 ? left : len1), rest_as_one ? left : len1, 1, L.d_stage + first * L.stride,
                                        rest_as_one ? 0 : L.stride, L.d_sizes + first, L.d_err, a, s);
        if (rest_as_one || first + 1 >= S) break;                                  // (a remainder's stream follows the final prefix directly)
        const size_t rest = S - first - 1;
        // start images of first+1 ..: the exact dictionary after `first`, then the last writers of first+1, first+2, ... laid over it
        if (e == hipSuccess) e = launch_merge_images(L.d_final + first * img, L.d_lw + (first + 1) * img, L.d_start + (first + 1) * img, (uint32_t)rest, s);
        if (e == hipSuccess) e = hipMemcpyAsync(L.d_gspec + first + 1, L.d_gfinal + first, sizeof(uint32_t), hipMemcpyDeviceToDevice, s);   // its successor starts from the true state
        if (e == hipSuccess) e = hipMemsetAsync(L.d_raw + first + 1, 0, rest * sizeof(uint32_t), s);
        if (e == hipSuccess) e = L.speculate(d_in, first + 1, rest, s);
        if (e == hipSuccess) e = hipMemcpyAsync(h_gfinal.data(), L.d_gfinal, S * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(h_raw.data(), L.d_raw, S * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        // (with the report: the sizes of every slot and the error word as they stand — if this pass turns out to be the last one they are final, and the
        // call saves two host round trips: round 5)
        if (e == hipSuccess) e = hipMemcpyAsync(h_sizes.data() + first_in, L.d_sizes + first_in, (S - first_in) * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && h_err) e = hipMemcpyAsync(h_err, L.d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) break;
        sizes_are_current = true;
        // segment first+1 started from the truth; k >= first+2 is final iff every segment first+1 .. k-1 coded all its blocks and k-1 ended calm
        size_t k = first + 2;
        while (k < S && h_raw[k - 1] == 0 && (h_gfinal[k - 1] & 0x7fffffffu) == 0) ++k;
        if (trace) fprintf(stderr, "[density_hip prof] segmented stream encode: pass %d, %zu segments of %zu bytes, final up to segment %zu\n", pass, S, L.C, k);
        advanced = k - first;
        first = k;                                                                // (== S: done)
        ++g_stream_stats[1];
        if (first < S) sizes_are_current = false;                                  // (another pass follows: it writes sizes again)
    }
    if (e != hipSuccess) return e;
    // passes that ran out at `first` < S: slots first .. are ONE stream in slot `first`
    *used = first < S ? first + 1 : S;
    if (sizes_are_current) return hipSuccess;                                      // the last pass's report carried them (and the error word)
    e = hipMemcpyAsync(h_sizes.data() + first_in, L.d_sizes + first_in, (*used - first_in) * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && h_err) e = hipMemcpyAsync(h_err, L.d_err, sizeof(uint32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    return e;
}

int run_stream_encode_segmented(DeviceCtx* c, const uint8_t* d_in, size_t n, uint8_t* d_out, hipStream_t s, size_t* size_out) {
    const bool trace = debug_env("DENSITY_HIP_PROF") != nullptr;
    SegEncode L;
    hipError_t e = L.setup(c, n, s);
    if (e != hipSuccess) { set_error("workspace allocation (segmented stream encode)", e); return DENSITY_HIP_ERR_RUNTIME; }
    const size_t S = L.S;
    std::vector<uint64_t> h_sizes(S), h_offsets(S);
    // last writers of every segment that has a successor and a predecessor (whole rounds: only the last segment can be short)
    // (on the context's second stream, beside the first segment's encode; joined before the first merge)
    e = hipEventRecord(c->batch_done[0], s);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stitch_stream, c->batch_done[0], 0);
    if (e == hipSuccess && S > 2) e = launch_rotor_lastwriters(d_in + L.C0, L.C, (uint32_t)(S - 2), L.d_lw + kSegImageBytes, L.d_err, c->stitch_stream);
    if (e == hipSuccess) e = hipEventRecord(c->stitch_done, c->stitch_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, c->stitch_done, 0);
    size_t used = S;
    uint32_t h_err = 0;
    if (e == hipSuccess) e = seg_encode_passes(L, d_in, 0, s, h_sizes, &used, trace, &h_err);   // (the sizes and the error word: read with the last pass's report)
    if (e != hipSuccess) { set_error("segmented stream encode", e); return DENSITY_HIP_ERR_RUNTIME; }
    if (h_err) { set_error("stream encode: device-side watchdog"); return DENSITY_HIP_ERR_RUNTIME; }
    uint64_t total = 0;
    for (size_t i = 0; i < used; ++i) { h_offsets[i] = total; total += h_sizes[i]; }
    e = hipMemcpyAsync(L.d_offsets, h_offsets.data(), used * sizeof(uint64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_compact_bytes(L.d_stage, L.stride, L.d_sizes, L.d_offsets, (uint32_t)used, d_out, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("segmented stream encode (gather)", e); return DENSITY_HIP_ERR_RUNTIME; }
    *size_out = (size_t)total;
    ++g_stream_stats[0];
    return DENSITY_HIP_OK;
}

}  // namespace

int run_stream_encode(DeviceCtx* c, int algo, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, uint8_t* ws, hipStream_t s,
                      size_t* size_out) {
    if (cap < safe_size(algo, n)) { set_error("output capacity below safe_encode_buffer_size()"); return DENSITY_HIP_ERR_CAPACITY; }
    if (algo == DENSITY_HIP_CHAMELEON && n >= kSegMinStream && n < (64ull << 30) && (reinterpret_cast<uintptr_t>(d_in) & 3) == 0 && !(g_variant & 5) && !g_rotor_unsafe) {   // (segments are at most 4 MiB: 32-bit positions inside them; 64 GiB = 16384 segments)
        *size_out = 0;
        return run_stream_encode_segmented(c, d_in, n, d_out, s, size_out);
    }
    uint64_t* d_sizes = reinterpret_cast<uint64_t*>(ws + kAlign);
    *size_out = 0;
    if (n == 0) return DENSITY_HIP_OK;
    Profiler prof(c, s);
    const DecodePlan sp = plan_decode(algo, 1);   // stream calls share the one-chunk decode layout
    uint32_t* d_err = reinterpret_cast<uint32_t*>(ws + sp.off_err);
    hipError_t e = hipMemsetAsync(d_err, 0, sizeof(uint32_t), s);
    // a long Cheetah / Lion stream is ONE chunk for the exchange passes (exchange_stages.hip): their scratch comes from the context
    uint8_t* d_stage = nullptr;
    if (e == hipSuccess && algo != DENSITY_HIP_CHAMELEON && stage_encode_eligible(algo, d_in, n, n, 1)) {
        e = c->seg.ensure(stage_scratch_bytes(algo, n, 1) + kAlign);
        d_stage = (uint8_t*)c->seg.p;
    }
    if (e == hipSuccess) e = codec_encode(algo, d_in, n, n, 1, d_out, 0, d_sizes, nullptr, ws + sp.off_tables, zmap_bytes(algo, 1) ? reinterpret_cast<uint32_t*>(ws + sp.off_zmap) : nullptr, d_stage, d_err, s);
    prof.mark(encode_kernel_name(algo));
    uint64_t h_size = 0;
    uint32_t h_err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_size, d_sizes, sizeof(h_size), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("stream encode", e); return DENSITY_HIP_ERR_RUNTIME; }
    if (h_err) { set_error("stream encode: device-side watchdog"); return DENSITY_HIP_ERR_RUNTIME; }
    *size_out = (size_t)h_size;
    return DENSITY_HIP_OK;
}

namespace {
// ---- ONE long Chameleon reference stream decoded in parallel ----
// stream_parse.hip finds the record boundaries of a calm stream in parallel: the block index and the stream offset of every 16384th
// block, i.e. the description of a container whose chunks are 4 MiB segments of the one stream.  A segment's start dictionary needs
// no speculation on decode: PLAIN quads write the dictionary whatever it holds and MAP quads never do, so it is the overlay of
// its predecessors' last PLAIN writers — which a decode pass from an EMPTY dictionary leaves behind as its final image (its MAP quads
// come out wrong, its writes are right; the real pass overwrites the output).  Passes: parse -> decode from empty dictionaries, final
// images -> per-slot merge into start images -> decode from the start images.  `handled` false: not a calm stream (or too short, or
// buffers this path does not take): the caller walks it on one work-group as before.
constexpr size_t kSegDecodeMin = 2u << 20;
// the geometry and the scratch of one segmented decode (the context's `seg` buffer)
struct SegDecode {
    uint32_t kChunkBlocks = 16384;
    size_t kChunkBytes = 0, max_chunks = 0, index_bytes = 0;
    uint8_t *base = nullptr, *d_index = nullptr, *d_lw = nullptr, *d_start = nullptr, *d_zero = nullptr, *d_carry = nullptr;
    uint32_t *d_pos32 = nullptr, *d_zmap = nullptr, *d_info = nullptr, *d_err = nullptr;
    uint64_t *d_chunk_offset = nullptr, *d_offsets = nullptr, *d_sizes = nullptr, *d_produced = nullptr;
    // false: more segments than the kernels take (the sequential path's business), or no memory (*e)
    bool setup(DeviceCtx* c, size_t E, size_t cap, hipError_t* e, size_t want_segments = 64) {
        // segments of 4 MiB of output for long streams, down to 256 KiB for short ones (about 64 segments at least; the pipelined host call, whose
        // slices each wait for ONE segment's worth of kernel time twice over, wants them short: about 1024)
        while (kChunkBlocks > 1024 && (E / 160) / kChunkBlocks < want_segments) kChunkBlocks >>= 1;
        kChunkBytes = (size_t)kChunkBlocks * 256;
        const size_t img = kSegImageBytes;
        size_t max_blocks = E / 136 + 2;
        if (cap / 256 + 2 < max_blocks) max_blocks = cap / 256 + 2;
        max_chunks = (max_blocks + kChunkBlocks - 1) / kChunkBlocks + 1;
        *e = hipSuccess;
        if (max_chunks > kMaxPipelinedChunks) return false;
        index_bytes = align_up(max_chunks * kChunkBlocks + 64, kAlign);
        const size_t parse_ws = align_up(stream_parse_workspace(E), kAlign);
        const size_t pos_bytes = align_up((max_chunks * kChunkBlocks + 64) * sizeof(uint32_t), kAlign);
        const size_t off_index = parse_ws, off_pos = off_index + index_bytes, off_lw = off_pos + pos_bytes, off_start = off_lw + max_chunks * img,
                     off_zero = off_start + (max_chunks + 1) * img, off_carry = off_zero + align_up(img, kAlign), off_zmap = off_carry + align_up(img, kAlign),
                     off_small = off_zmap + max_chunks * kZmapWordsPerChunk * 4;
        *e = c->seg.ensure(off_small + (max_chunks + 2) * 32 + 256 + kAlign);
        if (*e != hipSuccess) return false;
        base = (uint8_t*)c->seg.p;
        d_index = base + off_index;
        d_pos32 = reinterpret_cast<uint32_t*>(base + off_pos);
        d_lw = base + off_lw; d_start = base + off_start; d_zero = base + off_zero; d_carry = base + off_carry;
        d_zmap = reinterpret_cast<uint32_t*>(base + off_zmap);
        d_chunk_offset = reinterpret_cast<uint64_t*>(base + off_small);
        d_offsets = d_chunk_offset + max_chunks + 2;
        d_sizes = d_offsets + max_chunks + 2;
        d_produced = d_sizes + max_chunks + 2;
        d_info = reinterpret_cast<uint32_t*>(d_produced + max_chunks + 2);
        d_err = d_info + 16;                                                      // [0] the real pass, [1] the last-writer pass (ignored)
        return true;
    }
};

int run_stream_decode_segmented(DeviceCtx* c, const uint8_t* d_in, size_t E, uint8_t* d_out, size_t cap, hipStream_t s, size_t* size_out, bool* handled) {
    *handled = false;
    SegDecode D;
    hipError_t e;
    if (!D.setup(c, E, cap, &e)) {
        if (e != hipSuccess) { set_error("workspace allocation (segmented stream decode)", e); return DENSITY_HIP_ERR_RUNTIME; }
        return DENSITY_HIP_OK;
    }
    const uint32_t kChunkBlocks = D.kChunkBlocks;
    const size_t kChunkBytes = D.kChunkBytes, max_chunks = D.max_chunks, index_bytes = D.index_bytes, img = kSegImageBytes;
    uint8_t *base = D.base, *d_index = D.d_index;
    uint32_t *d_pos32 = D.d_pos32, *d_info = D.d_info, *d_err = D.d_err;
    uint64_t *d_chunk_offset = D.d_chunk_offset, *d_offsets = D.d_offsets, *d_sizes = D.d_sizes, *d_produced = D.d_produced;
    const bool trace = debug_env("DENSITY_HIP_PROF") != nullptr;
    e = hipMemsetAsync(d_chunk_offset, 0, (max_chunks + 2) * sizeof(uint64_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_info, 0, 18 * sizeof(uint32_t), s);
    // Parse.  A pair of incompressible records behind the head means raw copies follow: the parse is final up to that pair, the head walk
    // (real FSM) starts over from it and takes the raw copies, the parallel parse resumes behind them — up to 16 such episodes.
    uint32_t info[8] = {};
    uint32_t from_block = 0;
    uint64_t from_pos = 0;
    bool parsed = false;
    std::vector<uint64_t> h_off(max_chunks + 2), h_offsets, h_sizes;
    for (int episode = 0; e == hipSuccess && episode < 16; ++episode) {
        // ONE host round trip per episode (round 5; there were three): the start words go up in front of the parse on the same stream — they live
        // on this frame until the synchronisation below —, the verdict and the chunk offsets come down together behind it
        const uint32_t start[3] = {from_block, (uint32_t)from_pos, (uint32_t)(from_pos >> 32)};
        e = hipMemcpyAsync(d_info + 8, start, sizeof(start), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = launch_stream_parse(d_in, E, from_pos, base, d_index, max_chunks * kChunkBlocks, d_chunk_offset, kChunkBlocks, d_pos32, d_info, s);
        if (e == hipSuccess) e = hipMemcpyAsync(info, d_info, sizeof(info), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(h_off.data(), d_chunk_offset, (max_chunks + 2) * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) break;
        if (trace) fprintf(stderr, "[density_hip prof] segmented stream decode: parse from block %u: status %u, head to block %u, %u whole blocks, first incompressible pair at %d\n",
                           from_block, info[0], info[1], info[4], (int)info[7]);
        if (info[0] == 0) break;                                                  // no calm stretch within reach: the sequential path
        if (info[7] == 0xffffffffu) { parsed = true; break; }
        if (info[7] < from_block) break;                                          // (cannot happen)
        uint32_t p32 = 0;
        e = hipMemcpyAsync(&p32, d_pos32 + info[7], sizeof(p32), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        from_block = info[7]; from_pos = p32;
    }
    if (e != hipSuccess) { set_error("segmented stream decode (parse)", e); return DENSITY_HIP_ERR_RUNTIME; }
    const uint64_t whole = info[4], end_pos = ((uint64_t)info[6] << 32) | info[5];
    if (!parsed || whole < 2 * kChunkBlocks || end_pos > E) { if (trace) fprintf(stderr, "[density_hip prof]   -> sequential path (not calm enough / short)\n"); return DENSITY_HIP_OK; }
    // `whole` comes from the (untrusted) stream, the index was sized from the OUTPUT capacity: a stream that holds more blocks than the
    // output has room for is the sequential path's to refuse (a format error), before anything is sized or filled with it
    if (whole > max_chunks * kChunkBlocks || whole > index_bytes) { if (trace) fprintf(stderr, "[density_hip prof]   -> sequential path (stream longer than the output: %llu blocks)\n", (unsigned long long)whole); return DENSITY_HIP_OK; }
    // beyond the whole blocks the index says "ragged" = stop (an episode that was started over may have written further); in stream order in front of
    // the decode passes, no round trip
    e = hipMemsetAsync(d_index + whole, 0x7f, index_bytes - whole, s);
    if (e != hipSuccess) { set_error("segmented stream decode (parse)", e); return DENSITY_HIP_ERR_RUNTIME; }
    const bool ragged = end_pos < E;
    const size_t n_chunks = (whole + (ragged ? 1 : 0) + kChunkBlocks - 1) / kChunkBlocks;
    if (n_chunks > max_chunks || (n_chunks - 1) * kChunkBytes >= cap) { if (trace) fprintf(stderr, "[density_hip prof]   -> sequential path (capacity: %zu chunks, cap %zu)\n", n_chunks, cap); return DENSITY_HIP_OK; }
    if (ragged && whole % kChunkBlocks == 0) h_off[whole / kChunkBlocks] = end_pos;   // a ragged end that opens a chunk of its own
    h_off[0] = 0;
    h_offsets.resize(n_chunks); h_sizes.resize(n_chunks);
    for (size_t k = 0; k < n_chunks; ++k) {
        h_offsets[k] = h_off[k];
        h_sizes[k] = (k + 1 < n_chunks ? h_off[k + 1] : (uint64_t)E) - h_off[k];
        if (k && h_off[k] <= h_off[k - 1]) return DENSITY_HIP_OK;                     // (cannot happen; never hand the kernels a broken layout)
    }
    const uint64_t out_total = cap < n_chunks * kChunkBytes ? cap : n_chunks * kChunkBytes;
    uint32_t* d_zmap = D.d_zmap;
    if (!rotor_decode_eligible(d_out, (uint32_t)n_chunks, kChunkBytes, out_total, d_index, d_zmap) || g_rotor_unsafe) { if (trace) fprintf(stderr, "[density_hip prof]   -> sequential path (buffers not eligible)\n"); return DENSITY_HIP_OK; }
    e = hipMemcpyAsync(d_offsets, h_offsets.data(), n_chunks * sizeof(uint64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_sizes, h_sizes.data(), n_chunks * sizeof(uint64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(D.d_zero, 0, img, s);
    SegArgs lw;
    lw.final_images = D.d_lw;
    lw.lastwriters_only = 1;
    if (e == hipSuccess) e = launch_rotor_decode_seg(d_in, d_offsets, d_sizes, (uint32_t)n_chunks, d_out, kChunkBytes, out_total, d_index, d_zmap, d_produced, d_err + 1, lw, s);
    if (e == hipSuccess) e = launch_merge_images(D.d_zero, D.d_lw, D.d_start, (uint32_t)n_chunks, s);
    SegArgs real;
    real.init_images = D.d_start;
    if (e == hipSuccess) e = launch_rotor_decode_seg(d_in, d_offsets, d_sizes, (uint32_t)n_chunks, d_out, kChunkBytes, out_total, d_index, d_zmap, d_produced, d_err, real, s);
    uint64_t h_last = 0;
    uint32_t h_err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_last, d_produced + (n_chunks - 1), sizeof(h_last), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("segmented stream decode", e); return DENSITY_HIP_ERR_RUNTIME; }
    *handled = true;
    ++g_stream_stats[2];
    if (trace) fprintf(stderr, "[density_hip prof]   -> %zu segments decoded in parallel, err %u\n", n_chunks, h_err);
    if (h_err) { set_error("truncated stream or output too small"); return DENSITY_HIP_ERR_FORMAT; }
    *size_out = (n_chunks - 1) * kChunkBytes + (size_t)h_last;
    return DENSITY_HIP_OK;
}

}  // namespace

int run_stream_decode(DeviceCtx* c, int algo, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, uint8_t* ws, hipStream_t s,
                      size_t* size_out) {
    *size_out = 0;
    if (n == 0) return DENSITY_HIP_OK;
    if (algo == DENSITY_HIP_CHAMELEON && n >= kSegDecodeMin && n < (1ull << 32) && !(g_variant & 5) && !g_rotor_unsafe) {   // (the parse keeps 32-bit stream positions)
        bool handled = false;
        const int rc = run_stream_decode_segmented(c, d_in, n, d_out, cap, s, size_out, &handled);
        if (rc != DENSITY_HIP_OK || handled) return rc;
        *size_out = 0;
        ++g_stream_stats[3];
    }
    const DecodePlan p = plan_decode(algo, 1);
    uint32_t* d_err = reinterpret_cast<uint32_t*>(ws + p.off_err);
    uint64_t* d_sizes = reinterpret_cast<uint64_t*>(ws + p.off_sizes);
    uint64_t* d_offsets = reinterpret_cast<uint64_t*>(ws + p.off_offsets);
    uint64_t* d_produced = reinterpret_cast<uint64_t*>(ws + p.off_produced);
    const uint64_t h_size = n, h_off = 0;
    Profiler prof(c, s);
    hipError_t e = hipMemsetAsync(d_err, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_sizes, &h_size, 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_offsets, &h_off, 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);   // h_size/h_off live on this stack frame
    // a Cheetah stream is ONE chunk for the decode passes (decode_passes.hip): everything but its chain of contexts in parallel; their scratch
    // comes from the context
    // Their geometry and scratch follow the STREAM, not the caller's capacity (a small stream with a generous output buffer must not plan
    // passes over gigabytes): n stream bytes decode to at most 128 bytes per 8-byte signature (cheetah.rs:14-23: 32 quads per record, a
    // record of PREDICTED quads is its signature alone), plus a ragged end.  Short streams stay on one wave; so does any stream whose
    // scratch cannot be had (the passes are an optimisation, not a requirement).
    uint8_t* d_pass = nullptr;
    const size_t pass_cap = algo == DENSITY_HIP_CHEETAH ? std::min<size_t>(cap, (n / 8 + 2) * 128) : cap;
    if (e == hipSuccess && n >= 16384 && decode_pass_eligible(algo, d_out, 1, pass_cap, pass_cap)) {
        if (c->seg.ensure(decode_pass_scratch_bytes(align_up(pass_cap, 256), 1) + kAlign) == hipSuccess) d_pass = (uint8_t*)c->seg.p;
        else (void)hipGetLastError();                                                // (out of memory for the scratch: the one-wave decoder needs none)
    }
    const size_t dec_cap = d_pass ? pass_cap : cap;
    if (e == hipSuccess) e = codec_decode(algo, d_in, d_offsets, d_sizes, 1, d_out, dec_cap, dec_cap, false, nullptr, d_produced, d_err, ws + p.off_tables, zmap_bytes(algo, 1) ? reinterpret_cast<uint32_t*>(ws + p.off_zmap) : nullptr, s, d_pass);
    prof.mark(decode_kernel_name(algo));
    uint64_t h_prod = 0;
    uint32_t h_err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_prod, d_produced, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("stream decode", e); return DENSITY_HIP_ERR_RUNTIME; }
    if (h_err) { set_error("truncated stream or output too small"); return DENSITY_HIP_ERR_FORMAT; }
    *size_out = (size_t)h_prod;
    return DENSITY_HIP_OK;
}

// ---- host-pointer front ends ----
// (experiments: the number of slices of the pipelined calls below)
inline size_t slices_from_env(const char* name, size_t fallback) {
    const char* v = debug_env(name);
    const long k = v ? atol(v) : 0;
    return k >= 3 && k <= (long)kPipeMaxSlices ? (size_t)k : fallback;
}
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// chameleon_encode() of a long stream with the transfers beside the kernels.  The caller's buffers are pinned in place; the input goes up in slices
// of segments; behind every slice's arrival its last writers, start images and speculating segments are queued — a slice starts from the final
// dictionary of the segment in front of it, which is the truth exactly when everything before it coded all its blocks and ended calm, the same
// speculation as run_stream_encode_segmented's — and its streams are gathered behind the running end of the stream at once, on the same speculation;
// as soon as the host has seen a slice's verdicts, what was gathered goes down, while later slices are still on their way up.  The first segment that
// is not final ends the pipeline: what is left is encoded by the passes of the segmented encode from there (every slice has arrived by then), and
// follows in one piece.  Nothing on the device ever waits for an event that has not happened yet — the host queues a slice's kernels when its upload
// is through, its download when its kernels are: streams share a few hardware queues, and a wait at the head of one holds up whoever sits behind it.
constexpr size_t kPipeMinStream = 32u << 20;
size_t host_stream_encode_pipelined(DeviceCtx* c, const uint8_t* in, size_t n, uint8_t* out, size_t cap, bool* handled) {
    *handled = false;
    const size_t safe = safe_size(DENSITY_HIP_CHAMELEON, n);
    if (n < kPipeMinStream || n >= (64ull << 30) || (g_variant & (5 | 512)) || g_rotor_unsafe || cap < safe) return 0;
    const bool trace = debug_env("DENSITY_HIP_PROF") != nullptr;
    const double t0 = trace ? now_ms() : 0;
    PinnedInPlace pin_in(in, n), pin_out(out, safe);
    if (!pin_in || !pin_out) return 0;
    hipStream_t s = c->stream;
    hipError_t e = c->stage_in.ensure(n);
    if (e == hipSuccess) e = c->stage_out.ensure(safe);
    SegEncode L;
    if (e == hipSuccess) e = L.setup(c, n, s);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }                   // (the staged path reports what it cannot have either)
    const size_t S = L.S, img = kSegImageBytes;
    const size_t want_slices = slices_from_env("DENSITY_HIP_ENCODE_SLICES", 12);
    size_t per = (std::max<size_t>(n / want_slices, 2u << 20) + L.C - 1) / L.C;
    while ((S + per - 1) / per > kPipeMaxSlices) ++per;
    const uint32_t slices = (uint32_t)((S + per - 1) / per);
    if (slices < 3 || !pipe_streams(c, 2 * slices) || pin_meta_ensure(c, S * 16 + 64) != hipSuccess) { (void)hipGetLastError(); return 0; }
    *handled = true;
    uint64_t* p_sizes = reinterpret_cast<uint64_t*>(c->pin_meta);
    uint32_t* p_gfinal = reinterpret_cast<uint32_t*>(p_sizes + S);
    uint32_t* p_raw = p_gfinal + S;
    const uint8_t* d_in = (const uint8_t*)c->stage_in.p;
    uint8_t* d_out = (uint8_t*)c->stage_out.p;
    for (uint32_t j = 0; j < slices && e == hipSuccess; ++j) {
        const size_t a = (size_t)j * per, b = std::min(S, a + per);
        const size_t from = L.seg_at(a), to = b < S ? L.seg_at(b) : n;
        e = hipMemcpyAsync(const_cast<uint8_t*>(d_in) + from, in + from, to - from, hipMemcpyHostToDevice, c->up);
        if (e == hipSuccess) e = hipEventRecord(c->pipe_events[2 * j], c->up);
    }
    uint64_t total = 0;
    size_t final_to = 0;                                                          // segments [0, final_to) are final, gathered and on their way down
    bool broke = false;
    // a slice's verdicts: segments 0 and 1 started from the truth; k >= 2 is final iff every segment 1 .. k-1 coded all its blocks and k-1 ended calm.
    // What is final goes down from where the device gathered it.
    auto verdicts = [&](uint32_t j) -> hipError_t {
        const size_t a = (size_t)j * per, b = std::min(S, a + per);
        hipError_t x = hipEventSynchronize(c->pipe_events[2 * j + 1]);
        if (x != hipSuccess) return x;
        if (trace) fprintf(stderr, "[density_hip prof]   slice %u verdicts at %.3f ms\n", j, now_ms() - t0);
        size_t k = a;
        while (k < b && (k < 2 || (p_raw[k - 1] == 0 && (p_gfinal[k - 1] & 0x7fffffffu) == 0))) ++k;
        const uint64_t begin = total;
        for (size_t i = a; i < k; ++i) total += p_sizes[i];
        if (total > begin) x = hipMemcpyAsync(out + begin, d_out + begin, total - begin, hipMemcpyDeviceToHost, c->down);
        final_to = k;
        broke = k < b;
        return x;
    };
    for (uint32_t j = 0; j < slices && e == hipSuccess && !broke; ++j) {
        const size_t a = (size_t)j * per, b = std::min(S, a + per);
        e = hipEventSynchronize(c->pipe_events[2 * j]);                           // the slice has arrived
        // last writers of the slice's inner segments (every segment but the stream's first and last)
        const size_t lw_a = std::max<size_t>(a, 1), lw_b = std::min(b, S - 1);
        if (e == hipSuccess && lw_b > lw_a) e = launch_rotor_lastwriters(d_in + L.seg_at(lw_a), L.C, (uint32_t)(lw_b - lw_a), L.d_lw + lw_a * img, L.d_err, s);
        size_t sa = a;                                                            // the first speculating segment of the slice
        if (j == 0 && e == hipSuccess) {
            SegArgs a0;                                                           // the stream's first segment: fresh tables, fresh FSM
            a0.final_images = L.d_final; a0.final_guard = L.d_gfinal; a0.raw_blocks = L.d_raw;
            e = launch_rotor_encode_seg(d_in, L.seg_len(0), L.C0, 1, L.d_stage, L.stride, L.d_sizes, L.d_err, a0, s);
            if (e == hipSuccess) e = hipMemcpyAsync(L.d_gspec + 1, L.d_gfinal, sizeof(uint32_t), hipMemcpyDeviceToDevice, s);   // the second starts from the true state
            sa = 1;
        }
        if (e == hipSuccess && b > sa) e = launch_merge_images(L.d_final + (sa - 1) * img, L.d_lw + sa * img, L.d_start + sa * img, (uint32_t)(b - sa), s);
        if (e == hipSuccess && b > sa) e = L.speculate(d_in, sa, b - sa, s);
        if (e == hipSuccess) e = launch_scan_offsets(L.d_sizes, (uint32_t)a, (uint32_t)(b - a), L.d_carry, L.d_offsets, s);
        if (e == hipSuccess) e = launch_compact_bytes(L.d_stage + a * L.stride, L.stride, L.d_sizes + a, L.d_offsets + a, (uint32_t)(b - a), d_out, s);
        if (e == hipSuccess) e = hipMemcpyAsync(p_sizes + a, L.d_sizes + a, (b - a) * 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(p_gfinal + a, L.d_gfinal + a, (b - a) * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(p_raw + a, L.d_raw + a, (b - a) * 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipEventRecord(c->pipe_events[2 * j + 1], s);
        if (e == hipSuccess && j) e = verdicts(j - 1);                            // (beside this slice's kernels and the next one's upload)
    }
    if (e == hipSuccess && !broke) e = verdicts(slices - 1);
    if (trace) fprintf(stderr, "[density_hip prof] pipelined stream encode: %zu segments of %zu bytes in %u slices, final up to segment %zu at %.3f ms\n", S, L.C, slices, final_to, now_ms() - t0);
    ++g_stream_stats[1];
    // (always drained: nothing may still be reading or writing the caller's buffers when they are unpinned)
    const hipError_t e1 = hipStreamSynchronize(c->up), e3 = hipStreamSynchronize(s);
    if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e3;
    if (e == hipSuccess && final_to < S) {
        // (every last-writer image the passes need is there: the slices behind the break were queued or are queued now)
        for (uint32_t j = 0; j < slices && e == hipSuccess; ++j) {
            const size_t a = (size_t)j * per, b = std::min(S, a + per);
            const size_t lw_a = std::max<size_t>(std::max<size_t>(a, 1), final_to), lw_b = std::min(b, S - 1);
            if (lw_b > lw_a) e = launch_rotor_lastwriters(d_in + L.seg_at(lw_a), L.C, (uint32_t)(lw_b - lw_a), L.d_lw + lw_a * img, L.d_err, s);
        }
        std::vector<uint64_t> h_sizes(S), h_offsets(S);
        size_t used = S;
        if (e == hipSuccess) e = seg_encode_passes(L, d_in, final_to, s, h_sizes, &used, trace);
        if (e == hipSuccess) {
            const uint64_t begin = total;
            for (size_t k = final_to; k < used; ++k) { h_offsets[k] = total; total += h_sizes[k]; }
            e = hipMemcpy(L.d_offsets + final_to, h_offsets.data() + final_to, (used - final_to) * 8, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = launch_compact_bytes(L.d_stage + final_to * L.stride, L.stride, L.d_sizes + final_to, L.d_offsets + final_to, (uint32_t)(used - final_to), d_out, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e == hipSuccess && total > begin) e = hipMemcpyAsync(out + begin, d_out + begin, total - begin, hipMemcpyDeviceToHost, c->down);
        }
    }
    const hipError_t e5 = hipStreamSynchronize(c->down);
    if (e == hipSuccess) e = e5;
    uint32_t h_err = 0;
    if (e == hipSuccess) e = hipMemcpy(&h_err, L.d_err, sizeof(h_err), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("stream encode (pipelined host path)", e); return 0; }
    if (h_err) { set_error("stream encode: device-side watchdog"); return 0; }
    if (trace) fprintf(stderr, "[density_hip prof]   all down at %.3f ms\n", now_ms() - t0);
    ++g_stream_stats[0];
    return (size_t)total;
}

// chameleon_decode() of a long calm stream with the transfers beside the kernels.  The stream goes up in slices; behind every slice the parallel parse
// (stream_parse.hip) is taken up where it stopped — one block back, so that the head walk sees the record in front of the seam — over what has arrived
// so far, and the segments that are now whole are decoded: last writers from empty dictionaries, start images laid over the image carried on from the
// slice in front, the real pass, and their output on its way down while later slices are still coming up.  Anything but a calm stream that fits — raw
// copies, a parse that finds no calm head, an error flag — drops the attempt: `handled` stays false and the staged call gives the verdict.
// `uploaded`: set when the attempt was dropped AFTER the whole stream had gone up — it is in c->stage_in then, and the staged call that follows decodes
// it from there instead of sending it again (streams that turn out not to be calm late: raw copies in the last slice, an incompressible pair).
size_t host_stream_decode_pipelined(DeviceCtx* c, const uint8_t* in, size_t E, uint8_t* out, size_t cap, bool* handled, bool* uploaded) {
    *handled = false;
    *uploaded = false;
    if (E < kPipeMinStream / 2 || E >= (1ull << 32) || (g_variant & (5 | 512)) || g_rotor_unsafe || cap == 0) return 0;
    cap = std::min<size_t>(cap, (E / 136 + 2) * 256);                              // what the stream can decode to: 256 bytes per record of 136 bytes and more
    const size_t bound = cap;
    SegDecode D;
    hipError_t e;
    if (!D.setup(c, E, cap, &e, 1024)) { (void)hipGetLastError(); return 0; }
    const size_t kCB = D.kChunkBlocks, kCBy = D.kChunkBytes, img = kSegImageBytes;
    // (few slices: each pays the latency of the parse's kernels once, ≈0.2 ms, whatever its length — about 10 MiB of stream per slice, eight slices at
    // most: 64 MiB of text in 4 slices 37 GB/s, in 8 28; 256 MiB in 8 slices 45, in 3 39)
    const size_t want_slices = slices_from_env("DENSITY_HIP_DECODE_SLICES", std::min<size_t>(8, std::max<size_t>(3, E / (10u << 20))));
    const size_t slice = (std::max<size_t>((E + want_slices - 1) / want_slices, 4u << 20) + 255) & ~(size_t)255;
    const uint32_t slices = (uint32_t)((E + slice - 1) / slice);
    if (slices < 3 || slices > kPipeMaxSlices) return 0;
    PinnedInPlace pin_in(in, E), pin_out(out, bound);
    if (!pin_in || !pin_out) return 0;
    if (!pipe_streams(c, 2 * slices + 2) || pin_meta_ensure(c, 4096) != hipSuccess) { (void)hipGetLastError(); return 0; }
    e = c->stage_in.ensure(E);
    if (e == hipSuccess) e = c->stage_out.ensure(bound);
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }
    const bool trace = debug_env("DENSITY_HIP_PROF") != nullptr;
    const double t0 = trace ? now_ms() : 0;
    uint8_t* d_in = (uint8_t*)c->stage_in.p;
    uint8_t* d_out = (uint8_t*)c->stage_out.p;
    uint32_t* p_info = reinterpret_cast<uint32_t*>(c->pin_meta);                  // 16 words back, then three words per slice up
    uint64_t* p_last = reinterpret_cast<uint64_t*>(c->pin_meta + 64);
    uint32_t* p_err = reinterpret_cast<uint32_t*>(c->pin_meta + 80);
    uint64_t* p_open = reinterpret_cast<uint64_t*>(c->pin_meta + 96);
    uint32_t* p_start = reinterpret_cast<uint32_t*>(c->pin_meta + 128);
    hipStream_t s = c->stream, q = c->kern[1];                                     // the parse, slice by slice | the segments' kernels behind it
    e = hipMemsetAsync(D.d_chunk_offset, 0, (D.max_chunks + 2) * sizeof(uint64_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(D.d_info, 0, 18 * sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(D.d_carry, 0, img, s);                // the dictionary in front of the stream: empty
    if (e == hipSuccess) e = hipStreamSynchronize(s);                             // (the clears, before anything on the other stream)
    for (uint32_t j = 0; j < slices && e == hipSuccess; ++j) {
        const size_t from = (size_t)j * slice, to = std::min(E, from + slice);
        e = hipMemcpyAsync(d_in + from, in + from, to - from, hipMemcpyHostToDevice, c->up);
        if (e == hipSuccess) e = hipEventRecord(c->pipe_events[2 * j], c->up);
    }
    bool give_up = false;
    uint32_t from_block = 0;
    uint64_t from_pos = 0;
    size_t done = 0, n_chunks = 0;                                                // segments decoded so far; of the whole stream (known behind the last slice)
    size_t down_from = 0, down_bytes = 0;                                         // output that is being made and has yet to go down
    uint32_t down_slice = 0;
    for (uint32_t j = 0; j < slices && e == hipSuccess && !give_up; ++j) {
        const bool last = j + 1 == slices;
        const size_t have = last ? E : (size_t)(j + 1) * slice;
        p_start[3 * j] = from_block; p_start[3 * j + 1] = (uint32_t)from_pos; p_start[3 * j + 2] = (uint32_t)(from_pos >> 32);
        e = hipEventSynchronize(c->pipe_events[2 * j]);                           // the slice has arrived (nothing on the device waits: see the encoder above)
        if (e == hipSuccess) e = hipMemcpyAsync(D.d_info + 8, p_start + 3 * j, 12, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = launch_stream_parse(d_in, have, from_pos, D.base, D.d_index, D.max_chunks * kCB, D.d_chunk_offset, (uint32_t)kCB, D.d_pos32, D.d_info, s);
        if (e == hipSuccess) e = hipMemcpyAsync(p_info, D.d_info, 64, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && down_bytes) {                                      // the slice in front: its segments' output goes down once they are through
            e = hipEventSynchronize(c->pipe_events[2 * down_slice + 1]);
            if (e == hipSuccess) e = hipMemcpyAsync(out + down_from, d_out + down_from, down_bytes, hipMemcpyDeviceToHost, c->down);
            down_bytes = 0;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) break;
        // (from here on this slice's segments are `q`'s business, beside the next slice's parse on `s` — which rewrites the entries of block whole - 1
        // with what they hold already, and writes on behind them)
        const uint64_t whole = p_info[4], end_pos = ((uint64_t)p_info[6] << 32) | p_info[5];
        if (trace) fprintf(stderr, "[density_hip prof] pipelined stream decode: slice %u at %.3f ms, parse from block %u: status %u, %llu whole blocks, end %llu of %zu, first incompressible pair at %d\n",
                           j, now_ms() - t0, from_block, p_info[0], (unsigned long long)whole, (unsigned long long)end_pos, have, (int)p_info[7]);
        if (p_info[0] == 0 || p_info[7] != 0xffffffffu || end_pos > have || whole < from_block || whole > D.max_chunks * kCB || whole > D.index_bytes) { give_up = true; break; }
        const bool ragged = last && end_pos < E;
        const size_t upto = last ? (whole + (ragged ? 1 : 0) + kCB - 1) / kCB : whole / kCB;   // segments that are whole now (behind the last slice: all, the ragged end with them)
        if (last) {
            n_chunks = upto;
            if (whole < 2 * kCB || n_chunks > D.max_chunks || (n_chunks - 1) * kCBy >= cap) { give_up = true; break; }
        }
        if (upto > done) {
            const size_t count = upto - done;
            if (!last && upto * kCBy > cap) { give_up = true; break; }              // (the output is too small for what is there already: the staged call says so)
            const uint64_t out_off = done * kCBy, out_total = std::min<uint64_t>(cap - out_off, count * kCBy);
            if (!rotor_decode_eligible(d_out + out_off, (uint32_t)count, kCBy, out_total, D.d_index + done * kCB, D.d_zmap + done * kZmapWordsPerChunk)) { give_up = true; break; }
            if (last) {
                // beyond the whole blocks the index says "ragged" = stop; a ragged end that opens a segment of its own starts where the whole blocks end
                e = hipMemsetAsync(D.d_index + whole, 0x7f, D.index_bytes - whole, q);
                if (ragged && whole % kCB == 0) { *p_open = end_pos; if (e == hipSuccess) e = hipMemcpyAsync(D.d_chunk_offset + whole / kCB, p_open, 8, hipMemcpyHostToDevice, q); }
            }
            const uint64_t range_end = last ? (uint64_t)E : (whole % kCB == 0 ? end_pos : ~0ull);
            if (e == hipSuccess) e = launch_seg_layout(D.d_chunk_offset, (uint32_t)done, (uint32_t)count, range_end, D.d_offsets, D.d_sizes, D.d_err, q);
            SegArgs lw;
            lw.final_images = D.d_lw + done * img;
            lw.lastwriters_only = 1;
            if (e == hipSuccess) e = launch_rotor_decode_seg(d_in, D.d_offsets + done, D.d_sizes + done, (uint32_t)count, d_out + out_off, kCBy, out_total, D.d_index + done * kCB,
                                                             D.d_zmap + done * kZmapWordsPerChunk, D.d_produced + done, D.d_err + 1, lw, q);
            // start images of the range and, one more, the image the next range starts from
            if (e == hipSuccess) e = launch_merge_images(D.d_carry, D.d_lw + done * img, D.d_start + done * img, (uint32_t)count + 1, q);
            if (e == hipSuccess) e = hipMemcpyAsync(D.d_carry, D.d_start + upto * img, img, hipMemcpyDeviceToDevice, q);
            SegArgs real;
            real.init_images = D.d_start + done * img;
            if (e == hipSuccess) e = launch_rotor_decode_seg(d_in, D.d_offsets + done, D.d_sizes + done, (uint32_t)count, d_out + out_off, kCBy, out_total, D.d_index + done * kCB,
                                                             D.d_zmap + done * kZmapWordsPerChunk, D.d_produced + done, D.d_err, real, q);
            if (e == hipSuccess) e = hipEventRecord(c->pipe_events[2 * j + 1], q);
            const size_t full = last ? count - 1 : count;                         // (the stream's last segment goes down once its length is known)
            down_from = out_off; down_bytes = full * kCBy; down_slice = j;
            done = upto;
        }
        if (whole > 0) { from_block = (uint32_t)(whole - 1); from_pos = p_info[11]; }
    }
    size_t produced = 0;
    if (e == hipSuccess && !give_up) {
        if (done != n_chunks || n_chunks == 0) give_up = true;
        else {
            e = hipMemcpyAsync(p_last, D.d_produced + (n_chunks - 1), 8, hipMemcpyDeviceToHost, q);
            if (e == hipSuccess && down_bytes) {
                e = hipEventSynchronize(c->pipe_events[2 * down_slice + 1]);
                if (e == hipSuccess) e = hipMemcpyAsync(out + down_from, d_out + down_from, down_bytes, hipMemcpyDeviceToHost, c->down);
                down_bytes = 0;
            }
            if (e == hipSuccess) e = hipMemcpyAsync(p_err, D.d_err, 4, hipMemcpyDeviceToHost, q);
            if (e == hipSuccess) e = hipStreamSynchronize(q);
            if (e == hipSuccess && (*p_err || *p_last > kCBy || (n_chunks - 1) * kCBy + *p_last > cap)) give_up = true;   // (the staged call words the refusal)
            if (e == hipSuccess && !give_up) {
                produced = (n_chunks - 1) * kCBy + (size_t)*p_last;
                if (*p_last) e = hipMemcpyAsync(out + (n_chunks - 1) * kCBy, d_out + (n_chunks - 1) * kCBy, *p_last, hipMemcpyDeviceToHost, c->down);
            }
        }
    }
    // (always drained: nothing may still be reading or writing the caller's buffers when they are unpinned)
    const hipError_t e1 = hipStreamSynchronize(c->up), e2 = hipStreamSynchronize(s), e2b = hipStreamSynchronize(q), e3 = hipStreamSynchronize(c->down);
    if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e2b != hipSuccess ? e2b : e3;
    if (e != hipSuccess) { (void)hipGetLastError(); return 0; }                   // (handled stays false: the staged call reports what is wrong)
    if (give_up) { *uploaded = true; if (trace) fprintf(stderr, "[density_hip prof]   -> staged path (the stream stays where it is: on the device)\n"); return 0; }
    if (trace) fprintf(stderr, "[density_hip prof]   %zu segments of %zu bytes, all down at %.3f ms\n", n_chunks, kCBy, now_ms() - t0);
    *handled = true;
    ++g_stream_stats[2];
    return produced;
}

size_t host_stream_codec(int algo, bool encode, const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    g_last_error.clear();
    if (!in || !out || !valid_algo(algo)) { set_error("null pointer or bad algorithm"); return 0; }
    if (n == 0) return 0;
    DeviceCtx* c = acquire_ctx();
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    if (encode && algo == DENSITY_HIP_CHAMELEON) {
        bool handled = false;
        const size_t r = host_stream_encode_pipelined(c, in, n, out, cap, &handled);
        if (handled) return r;
    }
    bool uploaded = false;
    if (!encode && algo == DENSITY_HIP_CHAMELEON) {
        bool handled = false;
        const size_t r = host_stream_decode_pipelined(c, in, n, out, cap, &handled, &uploaded);
        if (handled) return r;
    }
    const size_t dev_cap = encode ? safe_size(algo, n) : cap;
    hipError_t e = c->stage_in.ensure(n);
    if (e == hipSuccess) e = c->stage_out.ensure(dev_cap ? dev_cap : 1);
    if (e == hipSuccess) e = c->work.ensure(plan_decode(algo, 1).total + kAlign);
    if (e == hipSuccess && !uploaded) e = copy_host_side_pinned(c->stage_in.p, in, n, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { set_error("staging (H2D)", e); return 0; }
    size_t produced = 0;
    const int rc = encode ? run_stream_encode(c, algo, (const uint8_t*)c->stage_in.p, n, (uint8_t*)c->stage_out.p, dev_cap, (uint8_t*)c->work.p, c->stream, &produced)
                          : run_stream_decode(c, algo, (const uint8_t*)c->stage_in.p, n, (uint8_t*)c->stage_out.p, dev_cap, (uint8_t*)c->work.p, c->stream, &produced);
    if (rc != DENSITY_HIP_OK) return 0;
    if (produced > cap) { set_error("output buffer too small"); return 0; }   // reference: slice-index panic (write_buffer.rs:19)
    if (produced) {
        e = copy_host_side_pinned(out, c->stage_out.p, produced, hipMemcpyDeviceToHost, c->stream);
        if (e != hipSuccess) { set_error("staging (D2H)", e); return 0; }
    }
    return produced;
}

}  // namespace api
}  // namespace density

using namespace density;
using namespace density::api;

extern "C" {

// ---- section 1: the reference's nine symbols ----
size_t chameleon_encode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size) { return host_stream_codec(DENSITY_HIP_CHAMELEON, true, input, input_size, output, output_size); }
size_t chameleon_decode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size) { return host_stream_codec(DENSITY_HIP_CHAMELEON, false, input, input_size, output, output_size); }
size_t chameleon_safe_encode_buffer_size(size_t size) { return safe_size(DENSITY_HIP_CHAMELEON, size); }
size_t cheetah_encode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size) { return host_stream_codec(DENSITY_HIP_CHEETAH, true, input, input_size, output, output_size); }
size_t cheetah_decode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size) { return host_stream_codec(DENSITY_HIP_CHEETAH, false, input, input_size, output, output_size); }
size_t cheetah_safe_encode_buffer_size(size_t size) { return safe_size(DENSITY_HIP_CHEETAH, size); }
size_t lion_encode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size) { return host_stream_codec(DENSITY_HIP_LION, true, input, input_size, output, output_size); }
size_t lion_decode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size) { return host_stream_codec(DENSITY_HIP_LION, false, input, input_size, output, output_size); }
size_t lion_safe_encode_buffer_size(size_t size) { return safe_size(DENSITY_HIP_LION, size); }

int density_hip_stream_encode_device(int algo, const void* d_input, size_t input_size, void* d_output, size_t output_capacity,
                                     void* stream, size_t* size_out) {
    g_last_error.clear();
    if (!valid_algo(algo) || !size_out || (!d_input && input_size) || (!d_output && output_capacity)) { set_error("bad argument"); return DENSITY_HIP_ERR_ARGUMENT; }
    DeviceCtx* c = acquire_ctx();
    if (!c) return DENSITY_HIP_ERR_RUNTIME;
    std::lock_guard<std::mutex> lk(c->mu);
    hipError_t e = c->work.ensure(plan_decode(algo, 1).total + kAlign);
    if (e != hipSuccess) { set_error("workspace allocation", e); return DENSITY_HIP_ERR_RUNTIME; }
    return run_stream_encode(c, algo, (const uint8_t*)d_input, input_size, (uint8_t*)d_output, output_capacity, (uint8_t*)c->work.p,
                             stream ? (hipStream_t)stream : c->stream, size_out);
}

int density_hip_stream_decode_device(int algo, const void* d_input, size_t input_size, void* d_output, size_t output_capacity,
                                     void* stream, size_t* size_out) {
    g_last_error.clear();
    if (!valid_algo(algo) || !size_out || (!d_input && input_size) || (!d_output && output_capacity)) { set_error("bad argument"); return DENSITY_HIP_ERR_ARGUMENT; }
    DeviceCtx* c = acquire_ctx();
    if (!c) return DENSITY_HIP_ERR_RUNTIME;
    std::lock_guard<std::mutex> lk(c->mu);
    hipError_t e = c->work.ensure(plan_decode(algo, 1).total + kAlign);
    if (e != hipSuccess) { set_error("workspace allocation", e); return DENSITY_HIP_ERR_RUNTIME; }
    return run_stream_decode(c, algo, (const uint8_t*)d_input, input_size, (uint8_t*)d_output, output_capacity, (uint8_t*)c->work.p,
                             stream ? (hipStream_t)stream : c->stream, size_out);
}

}  // extern "C"
