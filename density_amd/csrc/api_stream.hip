// api_stream.hip — ONE reference stream (include/density_hip.h section 1: the reference's nine symbols, chameleon.rs:70-83, cheetah.rs:105-118,
// lion.rs:193-206, and their device-pointer forms): short streams on one work-group / wave, long Chameleon streams encoded and decoded in
// parallel segments, byte for byte the reference's stream.
#include "api_internal.hpp"

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
int run_stream_encode_segmented(DeviceCtx* c, const uint8_t* d_in, size_t n, uint8_t* d_out, hipStream_t s, size_t* size_out) {
    // the first segment runs alone, ahead of everything else: a quarter of the others' length (whole rounds)
    const bool trace = getenv("DENSITY_HIP_PROF") != nullptr;
    const size_t C = seg_bytes_for(n), C0 = ((C / 4) + 4095) & ~(size_t)4095, S = 1 + (n - C0 + C - 1) / C, stride = slot_stride(DENSITY_HIP_CHAMELEON, C), img = kSegImageBytes;
    auto seg_at = [&](size_t k) -> size_t { return k == 0 ? 0 : C0 + (k - 1) * C; };   // where segment k starts
    const size_t off_lw = align_up(S * stride, kAlign), off_start = off_lw + S * img, off_final = off_start + S * img, off_small = off_final + S * img;
    hipError_t e = c->seg.ensure(off_small + S * 64 + kAlign);
    if (e != hipSuccess) { set_error("workspace allocation (segmented stream encode)", e); return DENSITY_HIP_ERR_RUNTIME; }
    uint8_t* base = (uint8_t*)c->seg.p;
    uint8_t *d_stage = base, *d_lw = base + off_lw, *d_start = base + off_start, *d_final = base + off_final;
    uint64_t* d_sizes = reinterpret_cast<uint64_t*>(base + off_small);
    uint64_t* d_offsets = d_sizes + S;
    uint32_t* d_gspec = reinterpret_cast<uint32_t*>(d_offsets + S);              // start FSM states of the speculating segments
    uint32_t* d_gfinal = d_gspec + S;
    uint32_t* d_raw = d_gfinal + S;
    uint32_t* d_err = d_raw + S;
    const uint32_t calm = 0x80000000u;                                             // pack_guard({0, 1, 0, 0}) = 0, speculation allowed
    std::vector<uint32_t> h_gspec(S, calm), h_gfinal(S), h_raw(S);
    std::vector<uint64_t> h_sizes(S), h_offsets(S);
    e = hipMemsetAsync(d_raw, 0, (S + 1) * sizeof(uint32_t), s);                  // raw counters + error word
    if (e == hipSuccess) e = hipMemcpyAsync(d_gspec, h_gspec.data(), S * sizeof(uint32_t), hipMemcpyHostToDevice, s);
    // last writers of every segment that has a successor and a predecessor (whole rounds: only the last segment can be short)
    // (on the context's second stream, beside the first segment's encode; joined before the first merge)
    if (e == hipSuccess) e = hipEventRecord(c->batch_done[0], s);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stitch_stream, c->batch_done[0], 0);
    if (e == hipSuccess && S > 2) e = launch_rotor_lastwriters(d_in + C0, C, (uint32_t)(S - 2), d_lw + img, d_err, c->stitch_stream);
    if (e == hipSuccess) e = hipEventRecord(c->stitch_done, c->stitch_stream);
    bool joined = false;
    size_t first = 0;
    size_t advanced = S;                                                           // segments the previous pass made final
    for (int pass = 0; e == hipSuccess && first < S; ++pass) {
        // (a pass that gets nowhere — raw copies all over — is not repeated for long: the remainder then runs as one chunk)
        const bool rest_as_one = pass >= 16 || (pass >= 3 && advanced < 8);
        // segment `first` (or, after too many passes, everything that is left as one chunk) from its exact start
        SegArgs a;
        a.init_images = first ? d_final + (first - 1) * img : nullptr;
        a.init_guard = first ? d_gfinal + (first - 1) : nullptr;
        a.final_images = d_final + first * img;
        a.final_guard = d_gfinal + first;
        a.raw_blocks = d_raw + first;
        const size_t left = n - seg_at(first), len1 = first == 0 ? C0 : C;
        e = launch_rotor_encode_seg(d_in + seg_at(first), rest_as_one ? left : (left < len1 ? left : len1), rest_as_one ? left : len1, 1, d_stage + first * stride,
                                    rest_as_one ? 0 : stride, d_sizes + first, d_err, a, s);
        if (rest_as_one || first + 1 >= S) { if (rest_as_one) { /* the remainder's stream follows the final prefix directly */ } break; }
        const size_t rest = S - first - 1;
        // start images of first+1 ..: the exact dictionary after `first`, then the last writers of first+1, first+2, ... laid over it
        if (e == hipSuccess && !joined) { e = hipStreamWaitEvent(s, c->stitch_done, 0); joined = true; }
        if (e == hipSuccess) e = launch_merge_images(d_final + first * img, d_lw + (first + 1) * img, d_start + (first + 1) * img, (uint32_t)rest, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_gspec + first + 1, d_gfinal + first, sizeof(uint32_t), hipMemcpyDeviceToDevice, s);   // its successor starts from the true state
        if (e == hipSuccess) e = hipMemsetAsync(d_raw + first + 1, 0, rest * sizeof(uint32_t), s);
        SegArgs b;
        b.init_images = d_start + (first + 1) * img;
        b.init_guard = d_gspec + first + 1;
        b.final_images = d_final + (first + 1) * img;
        b.final_guard = d_gfinal + first + 1;
        b.raw_blocks = d_raw + first + 1;
        if (e == hipSuccess) e = launch_rotor_encode_seg(d_in + seg_at(first + 1), n - seg_at(first + 1), C, (uint32_t)rest, d_stage + (first + 1) * stride, stride,
                                                         d_sizes + first + 1, d_err, b, s);
        if (e == hipSuccess) e = hipMemcpyAsync(h_gfinal.data(), d_gfinal, S * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(h_raw.data(), d_raw, S * sizeof(uint32_t), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) break;
        // segment first+1 started from the truth; k >= first+2 is final iff every segment first+1 .. k-1 coded all its blocks and k-1 ended calm
        size_t k = first + 2;
        while (k < S && h_raw[k - 1] == 0 && (h_gfinal[k - 1] & 0x7fffffffu) == 0) ++k;
        if (trace) fprintf(stderr, "[density_hip prof] segmented stream encode: pass %d, %zu segments of %zu bytes, final up to segment %zu\n", pass, S, C, k);
        advanced = k - first;
        first = k;                                                                // (== S: done)
        ++g_stream_stats[1];
    }
    if (!joined) { hipError_t j = hipStreamWaitEvent(s, c->stitch_done, 0); if (e == hipSuccess) e = j; }
    uint32_t h_err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(h_sizes.data(), d_sizes, S * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("segmented stream encode", e); return DENSITY_HIP_ERR_RUNTIME; }
    if (h_err) { set_error("stream encode: device-side watchdog"); return DENSITY_HIP_ERR_RUNTIME; }
    // a remainder encoded as one chunk sits in the slot of its first segment; the slots behind it are unused
    size_t used = S;
    if (first < S && first > 0) {
        // passes ran out at `first`: slots first .. are one stream in slot `first`
        used = first + 1;
    }
    uint64_t total = 0;
    for (size_t i = 0; i < used; ++i) { h_offsets[i] = total; total += h_sizes[i]; }
    e = hipMemcpyAsync(d_offsets, h_offsets.data(), used * sizeof(uint64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = launch_compact_bytes(d_stage, stride, d_sizes, d_offsets, (uint32_t)used, d_out, s);
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
int run_stream_decode_segmented(DeviceCtx* c, const uint8_t* d_in, size_t E, uint8_t* d_out, size_t cap, hipStream_t s, size_t* size_out, bool* handled) {
    *handled = false;
    // segments of 4 MiB of output for long streams, down to 256 KiB for short ones (about 64 segments at least)
    uint32_t kChunkBlocks = 16384;
    while (kChunkBlocks > 1024 && (E / 160) / kChunkBlocks < 64) kChunkBlocks >>= 1;
    const size_t kChunkBytes = (size_t)kChunkBlocks * 256, img = kSegImageBytes;
    size_t max_blocks = E / 136 + 2;
    if (cap / 256 + 2 < max_blocks) max_blocks = cap / 256 + 2;
    const size_t max_chunks = (max_blocks + kChunkBlocks - 1) / kChunkBlocks + 1;
    if (max_chunks > kMaxPipelinedChunks) return DENSITY_HIP_OK;
    const size_t index_bytes = align_up(max_chunks * kChunkBlocks + 64, kAlign), parse_ws = align_up(stream_parse_workspace(E), kAlign);
    const size_t pos_bytes = align_up((max_chunks * kChunkBlocks + 64) * sizeof(uint32_t), kAlign);
    const size_t off_index = parse_ws, off_pos = off_index + index_bytes, off_lw = off_pos + pos_bytes, off_start = off_lw + max_chunks * img,
                 off_zero = off_start + max_chunks * img, off_zmap = off_zero + align_up(img, kAlign), off_small = off_zmap + max_chunks * kZmapWordsPerChunk * 4;
    hipError_t e = c->seg.ensure(off_small + (max_chunks + 2) * 32 + 256 + kAlign);
    if (e != hipSuccess) { set_error("workspace allocation (segmented stream decode)", e); return DENSITY_HIP_ERR_RUNTIME; }
    uint8_t* base = (uint8_t*)c->seg.p;
    uint8_t* d_index = base + off_index;
    uint32_t* d_pos32 = reinterpret_cast<uint32_t*>(base + off_pos);
    uint64_t* d_chunk_offset = reinterpret_cast<uint64_t*>(base + off_small);
    uint64_t* d_offsets = d_chunk_offset + max_chunks + 2;
    uint64_t* d_sizes = d_offsets + max_chunks + 2;
    uint64_t* d_produced = d_sizes + max_chunks + 2;
    uint32_t* d_info = reinterpret_cast<uint32_t*>(d_produced + max_chunks + 2);
    uint32_t* d_err = d_info + 16;                                                // [0] the real pass, [1] the last-writer pass (ignored)
    const bool trace = getenv("DENSITY_HIP_PROF") != nullptr;
    e = hipMemsetAsync(d_chunk_offset, 0, (max_chunks + 2) * sizeof(uint64_t), s);
    if (e == hipSuccess) e = hipMemsetAsync(d_info, 0, 18 * sizeof(uint32_t), s);
    // Parse.  A pair of incompressible records behind the head means raw copies follow: the parse is final up to that pair, the head walk
    // (real FSM) starts over from it and takes the raw copies, the parallel parse resumes behind them — up to 16 such episodes.
    uint32_t info[8] = {};
    uint32_t from_block = 0;
    uint64_t from_pos = 0;
    bool parsed = false;
    for (int episode = 0; e == hipSuccess && episode < 16; ++episode) {
        const uint32_t start[3] = {from_block, (uint32_t)from_pos, (uint32_t)(from_pos >> 32)};
        e = hipMemcpyAsync(d_info + 8, start, sizeof(start), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);                         // (`start` lives on this frame)
        if (e == hipSuccess) e = launch_stream_parse(d_in, E, from_pos, base, d_index, max_chunks * kChunkBlocks, d_chunk_offset, kChunkBlocks, d_pos32, d_info, s);
        if (e == hipSuccess) e = hipMemcpyAsync(info, d_info, sizeof(info), hipMemcpyDeviceToHost, s);
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
    std::vector<uint64_t> h_off(max_chunks + 2), h_offsets, h_sizes;
    e = hipMemcpyAsync(h_off.data(), d_chunk_offset, (max_chunks + 2) * sizeof(uint64_t), hipMemcpyDeviceToHost, s);
    // beyond the whole blocks the index says "ragged" = stop (an episode that was started over may have written further)
    if (e == hipSuccess) e = hipMemsetAsync(d_index + whole, 0x7f, index_bytes - whole, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
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
    uint32_t* d_zmap = reinterpret_cast<uint32_t*>(base + off_zmap);
    if (!rotor_decode_eligible(d_out, (uint32_t)n_chunks, kChunkBytes, out_total, d_index, d_zmap) || g_rotor_unsafe) { if (trace) fprintf(stderr, "[density_hip prof]   -> sequential path (buffers not eligible)\n"); return DENSITY_HIP_OK; }
    e = hipMemcpyAsync(d_offsets, h_offsets.data(), n_chunks * sizeof(uint64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_sizes, h_sizes.data(), n_chunks * sizeof(uint64_t), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(base + off_zero, 0, img, s);
    SegArgs lw;
    lw.final_images = base + off_lw;
    lw.lastwriters_only = 1;
    if (e == hipSuccess) e = launch_rotor_decode_seg(d_in, d_offsets, d_sizes, (uint32_t)n_chunks, d_out, kChunkBytes, out_total, d_index, d_zmap, d_produced, d_err + 1, lw, s);
    if (e == hipSuccess) e = launch_merge_images(base + off_zero, base + off_lw, base + off_start, (uint32_t)n_chunks, s);
    SegArgs real;
    real.init_images = base + off_start;
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

size_t host_stream_codec(int algo, bool encode, const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    g_last_error.clear();
    if (!in || !out || !valid_algo(algo)) { set_error("null pointer or bad algorithm"); return 0; }
    if (n == 0) return 0;
    DeviceCtx* c = acquire_ctx();
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    const size_t dev_cap = encode ? safe_size(algo, n) : cap;
    hipError_t e = c->stage_in.ensure(n);
    if (e == hipSuccess) e = c->stage_out.ensure(dev_cap ? dev_cap : 1);
    if (e == hipSuccess) e = c->work.ensure(plan_decode(algo, 1).total + kAlign);
    if (e == hipSuccess) e = hipMemcpyAsync(c->stage_in.p, in, n, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { set_error("staging (H2D)", e); return 0; }
    size_t produced = 0;
    const int rc = encode ? run_stream_encode(c, algo, (const uint8_t*)c->stage_in.p, n, (uint8_t*)c->stage_out.p, dev_cap, (uint8_t*)c->work.p, c->stream, &produced)
                          : run_stream_decode(c, algo, (const uint8_t*)c->stage_in.p, n, (uint8_t*)c->stage_out.p, dev_cap, (uint8_t*)c->work.p, c->stream, &produced);
    if (rc != DENSITY_HIP_OK) return 0;
    if (produced > cap) { set_error("output buffer too small"); return 0; }   // reference: slice-index panic (write_buffer.rs:19)
    if (produced) {
        e = hipMemcpy(out, c->stage_out.p, produced, hipMemcpyDeviceToHost);
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
