// api.hip — the C ABI of libdensity_hip.so (include/density_hip.h), first of three units: per-device context, workspace plans, the
// container's device-side drivers and the device-pointer / bookkeeping entry points.  (api_stream.hip: one reference stream — the
// reference's nine symbols; api_host.hip: the host-pointer container calls.)  No CPU codec lives here: every byte is produced by the
// gfx950 kernels, and every entry point fails (returns 0 / an error code) when no usable HIP device is present.
#include "api_internal.hpp"

namespace density {
namespace api {

thread_local std::string g_last_error;
int g_profiling = 0;
int g_variant = 0;
uint64_t g_pass_decodes = 0;
uint64_t g_stream_stats[4] = {0, 0, 0, 0};
DeviceCtx g_ctx[kMaxDevices];

void set_error(const char* what, hipError_t e) {
    g_last_error = what;
    if (e != hipSuccess) { g_last_error += ": "; g_last_error += hipGetErrorString(e); }
}

DeviceCtx* acquire_ctx() {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) { set_error("no HIP device available", e); return nullptr; }
    DeviceCtx* c = &g_ctx[dev];
    std::lock_guard<std::mutex> lk(c->mu);
    if (!c->ready) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, dev);
        if (e != hipSuccess) { set_error("hipGetDeviceProperties", e); return nullptr; }
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) { set_error("device is not gfx950 (MI355X); this library has no other code path"); return nullptr; }
        e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stitch_stream, hipStreamNonBlocking);
        for (int i = 0; i < 8 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&c->batch_done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->stitch_done, hipEventDisableTiming);
        if (e != hipSuccess) { set_error("hipStreamCreate", e); return nullptr; }
        // kernels rely on ascending-lane service order of same-address LDS accesses: verify on this device
        uint32_t* d_fail = nullptr;
        uint32_t h_fail = 1;
        e = hipMalloc((void**)&d_fail, sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(d_fail, 0, sizeof(uint32_t), c->stream);
        if (e == hipSuccess) e = launch_selftest(d_fail, c->stream);
        if (e == hipSuccess) e = launch_rotor_selftest(d_fail, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(&h_fail, d_fail, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (d_fail) (void)hipFree(d_fail);
        if (e != hipSuccess) { set_error("LDS self-test launch", e); return nullptr; }
        // bits 0..7 (container.hip): plain 16-bit LDS writes in lane order — every kernel needs it; bits 8..11 (rotor.hip): lane order of the
        // ordered exchange ds_mskor_rtn_b32 — all but the one-wavefront kernels need it; bits 12, 13: lane-reversed rollback and the token
        // hand-off behind the exchanges — only the wave-rotation kernels need them.  A device that fails a later group runs on what is left.
        c->selftest_bits = h_fail;
        c->selftest_ok = (h_fail & 0xffu) == 0;
        density::g_exchange_unsafe = (h_fail & 0x0f00u) != 0;
        density::g_rotor_unsafe = (h_fail & 0xff00u) != 0;
        c->ready = true;
    }
    if (!c->selftest_ok) { set_error("LDS write-order self-test failed on this device; refusing to run"); return nullptr; }
    return c;
}

EncodePlan plan_encode(int algo, size_t n, size_t chunk) {
    EncodePlan p{};
    p.chunk = chunk;
    p.n_chunks = chunk_count(n, chunk);
    p.stride = slot_stride(algo, chunk);
    p.off_err = 0;
    p.off_sizes = kAlign;
    p.off_offsets = p.off_sizes + align_up(8 * p.n_chunks, kAlign);
    p.off_slots = p.off_offsets + align_up(8 * (p.n_chunks + 1), kAlign);
    p.off_tables = p.off_slots + (p.n_chunks > 1 ? p.n_chunks * p.stride : 0);   // one chunk encodes straight into the container
    p.off_zmap = p.off_tables + serial_tables(algo, p.n_chunks ? p.n_chunks : 1);
    p.off_stage = p.off_zmap + zmap_bytes(algo, p.n_chunks);
    // the exchange passes of Cheetah / Lion (exchange_stages.hip): a dword per quad, the per-block masks, the record offsets
    // — only where the passes will run: every condition of stage_encode_eligible but the input pointer's alignment is known here (chunk count
    // limits, head size, table budget, forced variants), and a caller sizing its own workspace should not pay 1.25-1.5 x the input for nothing
    const bool passes = algo != DENSITY_HIP_CHAMELEON && p.n_chunks <= 0xffffffffull && serial_slots(algo, p.n_chunks) == p.n_chunks &&   // (the passes keep a table slot per CHUNK)
                        stage_encode_eligible(algo, nullptr, n, chunk, (uint32_t)p.n_chunks);
    p.total = p.off_stage + (passes ? align_up(stage_scratch_bytes(algo, n, (uint32_t)p.n_chunks), kAlign) : 0);
    return p;
}
DecodePlan plan_decode(int algo, size_t n_chunks, size_t out_stride) {
    DecodePlan p{};
    p.off_err = 0;
    p.off_sizes = kAlign;
    p.off_offsets = p.off_sizes + align_up(8 * n_chunks, kAlign);
    p.off_produced = p.off_offsets + align_up(8 * (n_chunks + 1), kAlign);
    p.off_tables = p.off_produced + align_up(8 * (n_chunks ? n_chunks : 1), kAlign);
    p.off_zmap = p.off_tables + serial_tables(algo, n_chunks ? n_chunks : 1);
    p.total = p.off_zmap + zmap_bytes(algo, n_chunks);
    p.off_pass = p.total;
    p.total_with_passes = p.total;
    if (algo == DENSITY_HIP_CHEETAH && out_stride && n_chunks)
        p.total_with_passes = p.off_pass + align_up(decode_pass_scratch_bytes(n_chunks == 1 ? align_up(out_stride, 256) : out_stride, (uint32_t)n_chunks), kAlign);
    return p;
}

// algorithm dispatch: Chameleon has the LDS-resident pipelined kernels, Cheetah/Lion the functional one-lane-per-stream kernels
hipError_t codec_encode(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out,
                        uint64_t out_stride, uint64_t* d_sizes, uint8_t* d_index, uint8_t* d_tables, uint32_t* d_zmap, uint8_t* d_stage, uint32_t* d_err, hipStream_t s) {
    if (algo == DENSITY_HIP_CHAMELEON) return launch_chameleon_encode(d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_index, d_zmap, d_err, s);
    if (d_stage && serial_slots(algo, n_chunks) == n_chunks && stage_encode_eligible(algo, d_in, total, chunk_bytes, n_chunks))   // Cheetah / Lion: passes of ordered LDS exchanges, the one-wave kernels for what they hand back
        return launch_stage_encode(algo, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, (uint32_t)serial_slots(algo, n_chunks), d_stage, d_err, s);
    return launch_serial_encode(algo, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, (uint32_t)serial_slots(algo, n_chunks), s);
}
hipError_t codec_decode(int algo, const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out,
                        uint64_t out_stride, uint64_t out_total, bool exact, const uint8_t* d_index, uint64_t* d_produced, uint32_t* d_err,
                        uint8_t* d_tables, uint32_t* d_zmap, hipStream_t s, uint8_t* d_pass) {
    if (algo == DENSITY_HIP_CHAMELEON) return launch_chameleon_decode(d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact, d_index, d_zmap, d_produced, d_err, s);
    if (d_pass && decode_pass_eligible(algo, d_out, n_chunks, out_stride, out_total) && ++g_pass_decodes)   // Cheetah: parallel inside the chunk but for the chain of contexts
        return launch_decode_passes(algo, d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact, d_produced, d_err, d_pass, s);
    return launch_serial_decode(algo, d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact, d_produced, d_err, d_tables, (uint32_t)serial_slots(algo, n_chunks), s);
}
const char* encode_kernel_name(int algo) { return algo == DENSITY_HIP_CHAMELEON ? "chameleon_encode_chunks" : algo == DENSITY_HIP_CHEETAH ? "cheetah_encode_chunks" : "lion_encode_chunks"; }
const char* decode_kernel_name(int algo) { return algo == DENSITY_HIP_CHAMELEON ? "chameleon_decode_chunks" : algo == DENSITY_HIP_CHEETAH ? "cheetah_decode_chunks" : "lion_decode_chunks"; }

size_t container_bound(int algo, size_t n, size_t chunk) {
    const size_t nc = chunk_count(n, chunk);
    size_t bound = payload_base(nc, n, true);      // an upper bound for both flavours
    if (nc) bound += (nc - 1) * align_up(safe_size(algo, chunk), 16) + safe_size(algo, n - (nc - 1) * chunk);
    return bound;
}

// a slotted container: every payload in its worst-case slot (the last one as long as its chunk can get)
size_t container_bound_slotted(int algo, size_t n, size_t chunk) {
    const size_t nc = chunk_count(n, chunk);
    size_t bound = payload_base(nc, n, true);
    if (nc) bound += (nc - 1) * slot_stride(algo, chunk) + safe_size(algo, n - (nc - 1) * chunk);
    return bound;
}

// a paged container: every chunk as many pages as its worst case needs (they are taken as the streams grow: a container of text ends far below this)
size_t container_bound_paged(int algo, size_t n, size_t chunk) {
    if (!paged_eligible(algo, n, chunk)) return container_bound_slotted(algo, n, chunk);
    const size_t nc = chunk_count(n, chunk);
    return paged_pages_base(nc, n, chunk) + nc * (size_t)paged_pages_per_chunk(chunk) * kPageBytes;
}

int check_header(const density_hip_header_t& h, size_t container_size) {
    if (h.magic != DENSITY_HIP_MAGIC || h.version != 1 || !valid_algo(h.algo)) return DENSITY_HIP_ERR_FORMAT;
    if (!valid_chunk(h.chunk_size)) return DENSITY_HIP_ERR_FORMAT;
    if (h.n_chunks != chunk_count(h.total_len, h.chunk_size)) return DENSITY_HIP_ERR_FORMAT;
    if (h.flags & ~(DENSITY_HIP_FLAG_BLOCK_INDEX | DENSITY_HIP_FLAG_SLOTTED | DENSITY_HIP_FLAG_PAGED)) return DENSITY_HIP_ERR_FORMAT;
    if (h.flags & DENSITY_HIP_FLAG_PAGED) {                                        // pages: Chameleon with its block index, whole pages behind the directory
        if (h.algo != DENSITY_HIP_CHAMELEON || (h.flags & (DENSITY_HIP_FLAG_BLOCK_INDEX | DENSITY_HIP_FLAG_SLOTTED)) != DENSITY_HIP_FLAG_BLOCK_INDEX) return DENSITY_HIP_ERR_FORMAT;
        const size_t pb = paged_pages_base(h.n_chunks, h.total_len, h.chunk_size);
        if (h.container_len > container_size || h.container_len < pb || (h.container_len - pb) % kPageBytes != 0 || h.container_len - pb >= (1ull << 32)) return DENSITY_HIP_ERR_FORMAT;
        return DENSITY_HIP_OK;
    }
    if (h.container_len > container_size || h.container_len < payload_base(h.n_chunks, h.total_len, h.flags & DENSITY_HIP_FLAG_BLOCK_INDEX)) return DENSITY_HIP_ERR_FORMAT;
    return DENSITY_HIP_OK;
}

// ---- device-side drivers (ctx already acquired; `ws` points at a workspace of sufficient size) ----

int run_encode_container(DeviceCtx* c, int algo, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, size_t chunk,
                         uint8_t* ws, hipStream_t s, density_hip_header_t* header_out, bool slotted, bool paged) {
    const EncodePlan p = plan_encode(algo, n, chunk);
    if (p.n_chunks > 0xffffffffull) { set_error("too many chunks"); return DENSITY_HIP_ERR_ARGUMENT; }
    if (p.n_chunks <= 1) slotted = false;                                              // (one chunk encodes straight into place either way)
    if (paged && !(paged_eligible(algo, n, chunk) && rotor_encode_eligible(d_in, n, chunk, (uint32_t)p.n_chunks) && !g_rotor_unsafe && !(g_variant & 5))) { paged = false; slotted = p.n_chunks > 1; }   // (what the paged form is not for: the slotted one)
    if (cap < (paged ? container_bound_paged(algo, n, chunk) : slotted ? container_bound_slotted(algo, n, chunk) : container_bound(algo, n, chunk))) { set_error("output capacity below density_hip_container_bound()"); return DENSITY_HIP_ERR_CAPACITY; }
    uint32_t* d_err = reinterpret_cast<uint32_t*>(ws + p.off_err);
    uint64_t* d_sizes = reinterpret_cast<uint64_t*>(ws + p.off_sizes);
    uint64_t* d_offsets = reinterpret_cast<uint64_t*>(ws + p.off_offsets);
    uint8_t* d_slots = ws + p.off_slots;
    uint32_t* d_zmap = zmap_bytes(algo, p.n_chunks) ? reinterpret_cast<uint32_t*>(ws + p.off_zmap) : nullptr;
    density_hip_header_t hdr{};
    hdr.magic = DENSITY_HIP_MAGIC; hdr.algo = (uint8_t)algo; hdr.version = 1; hdr.flags = (want_index(algo) ? DENSITY_HIP_FLAG_BLOCK_INDEX : 0) | (slotted ? DENSITY_HIP_FLAG_SLOTTED : 0);
    hdr.chunk_size = (uint32_t)chunk; hdr.n_chunks = (uint32_t)p.n_chunks; hdr.total_len = n; hdr.container_len = 0;

    const bool with_index = hdr.flags & DENSITY_HIP_FLAG_BLOCK_INDEX;
    const uint64_t pbase = payload_base(p.n_chunks, n, with_index);
    uint8_t* d_index = with_index ? d_out + index_base(p.n_chunks) : nullptr;
    Profiler prof(c, s);
    hipError_t e = hipMemsetAsync(d_err, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) { set_error("hipMemsetAsync", e); return DENSITY_HIP_ERR_RUNTIME; }
    if (paged) {
        // Paged container (round 5): the wire form WITHOUT a stitch.  The encode kernel places the streams itself, page by page (64 KiB, from one
        // counter, in the order the chunks ask for them): dense but for the unused tails of the pages, and a chunk's stream is its pages' used bytes
        // in directory order.  write_buffer.rs:29-31's running total lives in the directory.
        hdr.flags = DENSITY_HIP_FLAG_BLOCK_INDEX | DENSITY_HIP_FLAG_PAGED;
        const uint64_t dir_base = paged_dir_base(p.n_chunks, n), pages_base = paged_pages_base(p.n_chunks, n, chunk);
        const uint32_t ppc = paged_pages_per_chunk(chunk);
        uint32_t* d_counter = d_err + 4;
        e = hipMemsetAsync(d_counter, 0, sizeof(uint32_t), s);
        if (e == hipSuccess) e = launch_rotor_encode_paged(d_in, n, chunk, (uint32_t)p.n_chunks, d_out + pages_base, (uint32_t)std::min<uint64_t>((cap - pages_base) / kPageBytes, 0xffffu),
                                                         d_counter, reinterpret_cast<uint32_t*>(d_out + dir_base), page_dir_words(ppc), d_sizes, d_index, d_err, s);
        prof.mark(encode_kernel_name(algo));
        if (e == hipSuccess) e = launch_layout_encode_paged(d_sizes, (uint32_t)p.n_chunks, hdr, dir_base, dir_base + paged_dir_bytes(p.n_chunks, chunk), pages_base, d_out, cap, d_counter, d_err, s);
        prof.mark("layout_encode");
    } else if (p.n_chunks == 1) {
        // single chunk: its stream goes straight to its final place, no stitch pass
        e = codec_encode(algo, d_in, n, chunk, 1, d_out + pbase, 0, d_sizes, d_index, ws + p.off_tables, d_zmap, p.total > p.off_stage ? ws + p.off_stage : nullptr, d_err, s);
        prof.mark(encode_kernel_name(algo));
        if (e == hipSuccess) e = launch_layout_encode(d_sizes, 1, hdr, pbase, d_out, cap, d_offsets, d_err, s);
        prof.mark("layout_encode");
    } else if (slotted) {
        // Slotted container: the chunk streams stay where the encoder put them — worst-case slots INSIDE the container, at payload_base +
        // i * slot_stride — and the size table says how much of each slot is stream.  No gather: the decoder reads the slots through the
        // same arithmetic; the packed wire form is made when the container leaves the device (density_hip_pack_device: a copy happens there anyway).
        e = codec_encode(algo, d_in, n, chunk, (uint32_t)p.n_chunks, d_out + pbase, p.stride, d_sizes, d_index, ws + p.off_tables, d_zmap,
                         p.total > p.off_stage ? ws + p.off_stage : nullptr, d_err, s);
        prof.mark(encode_kernel_name(algo));
        if (e == hipSuccess) e = launch_layout_encode(d_sizes, (uint32_t)p.n_chunks, hdr, pbase, d_out, cap, d_offsets, d_err, s, p.stride);
        prof.mark("layout_encode");
    } else {
        // Chunk streams go to worst-case slots; their sizes are known only afterwards (write_buffer.rs:29-31 keeps a running total: in
        // parallel an exclusive scan), then the streams are gathered into the packed container.  Optional (kernel variant bit 3, measured, not
        // the default): large inputs encoded in up to kStitchBatches batches of whole multiples of 256 chunks (one per CU) with the gather of
        // batch k on a second stream beside the encoding of batch k+1.  It hides the gather but the encoder — whose dictionary chain is
        // sensitive to load latency — slows down by as much (0.68 + 0.10 ms against 0.56 + 0.25 ms per GiB): DESIGN.md.
        constexpr uint32_t kStitchBatches = 4;
        const uint32_t nch = (uint32_t)p.n_chunks;
        uint32_t per = nch, batches = 1;
        if (algo == DENSITY_HIP_CHAMELEON && nch >= 512 && (g_variant & 8)) {
            batches = nch / 256 < kStitchBatches ? nch / 256 : kStitchBatches;
            per = ((nch + batches - 1) / batches + 255) / 256 * 256;
            batches = (nch + per - 1) / per;
        }
        uint64_t* d_carry = d_offsets + p.n_chunks;                      // (the extra entry of the offsets array)
        for (uint32_t k = 0; k < batches && e == hipSuccess; ++k) {
            const uint32_t first = k * per, count = (first + per <= nch) ? per : nch - first;
            const uint64_t in_off = (uint64_t)first * chunk;
            e = codec_encode(algo, d_in + in_off, n - in_off < (uint64_t)count * chunk ? n - in_off : (uint64_t)count * chunk, chunk, count, d_slots + (uint64_t)first * p.stride,
                             p.stride, d_sizes + first, d_index ? d_index + in_off / 256 : nullptr, ws + p.off_tables,
                             d_zmap ? d_zmap + (uint64_t)first * kZmapWordsPerChunk : nullptr, p.total > p.off_stage ? ws + p.off_stage : nullptr, d_err, s);
            prof.mark(encode_kernel_name(algo));
            if (e == hipSuccess) e = launch_layout_encode_batch(d_sizes, first, count, k == 0, k + 1 == batches, hdr, pbase, d_out, cap, d_offsets, d_carry, d_err, s);
            prof.mark("layout_encode");
            if (e != hipSuccess) break;
            if (batches == 1) {
                e = launch_compact(d_slots, p.stride, d_sizes, d_offsets, count, d_out, d_err, s);
                prof.mark("compact");
            } else {
                e = hipEventRecord(c->batch_done[k], s);
                if (e == hipSuccess) e = hipStreamWaitEvent(c->stitch_stream, c->batch_done[k], 0);
                if (e == hipSuccess) e = launch_compact(d_slots + (uint64_t)first * p.stride, p.stride, d_sizes + first, d_offsets + first, count, d_out, d_err, c->stitch_stream, k + 1 < batches);
            }
        }
        if (e == hipSuccess && batches > 1) {                               // the caller's stream continues when the last gather is done
            e = hipEventRecord(c->stitch_done, c->stitch_stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(s, c->stitch_done, 0);
            prof.mark("stitch_tail");
        }
    }
    if (e != hipSuccess) { set_error("kernel launch (encode)", e); return DENSITY_HIP_ERR_RUNTIME; }
    if (header_out) {
        uint32_t h_err = 0;
        e = hipMemcpyAsync(header_out, d_out, sizeof(*header_out), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("encode (device)", e); return DENSITY_HIP_ERR_RUNTIME; }
        if (h_err & 16u) { set_error("encode: device-side watchdog"); return DENSITY_HIP_ERR_RUNTIME; }
        if (h_err) { set_error("container does not fit the output capacity"); return DENSITY_HIP_ERR_CAPACITY; }
    }
    return DENSITY_HIP_OK;
}

int run_decode_container(DeviceCtx* c, const uint8_t* d_in, size_t container_size, const density_hip_header_t& h, uint8_t* d_out,
                         size_t cap, uint8_t* ws, hipStream_t s, size_t* decoded_out, size_t ws_size) {
    if (cap < h.total_len) { set_error("output capacity below the container's total_len"); return DENSITY_HIP_ERR_CAPACITY; }
    const DecodePlan p = plan_decode(h.algo, h.n_chunks, h.chunk_size);
    uint8_t* d_pass = (ws_size >= p.total_with_passes && p.total_with_passes > p.total) ? ws + p.off_pass : nullptr;   // (a caller's smaller workspace: the one-wave decoder)
    uint32_t* d_err = reinterpret_cast<uint32_t*>(ws + p.off_err);
    uint64_t* d_sizes = reinterpret_cast<uint64_t*>(ws + p.off_sizes);
    uint64_t* d_offsets = reinterpret_cast<uint64_t*>(ws + p.off_offsets);
    uint64_t* d_produced = reinterpret_cast<uint64_t*>(ws + p.off_produced);
    Profiler prof(c, s);
    hipError_t e = hipMemsetAsync(d_err, 0, sizeof(uint32_t), s);
    const bool with_index = h.flags & DENSITY_HIP_FLAG_BLOCK_INDEX;
    const uint8_t* d_index = with_index ? d_in + index_base(h.n_chunks) : nullptr;
    if (h.flags & DENSITY_HIP_FLAG_PAGED) {
        // the pages are read where they lie: the rotation decoder turns stream positions into page offsets through the chunk's directory
        uint32_t* d_zmap = zmap_bytes(h.algo, h.n_chunks) ? reinterpret_cast<uint32_t*>(ws + p.off_zmap) : nullptr;
        const size_t dir_base = paged_dir_base(h.n_chunks, h.total_len), pages_base = paged_pages_base(h.n_chunks, h.total_len, h.chunk_size);
        if (g_rotor_unsafe || !rotor_decode_eligible(d_out, h.n_chunks, h.chunk_size, h.total_len, d_index, d_zmap) || (uintptr_t)(d_in + pages_base) % 4 != 0) {
            set_error("a paged container needs the rotation decoder (chunks of at most 4 MiB, 4-byte aligned buffers)"); return DENSITY_HIP_ERR_UNSUPPORTED;
        }
        if (e == hipSuccess) e = launch_layout_decode_paged(d_in, h.n_chunks, d_sizes, d_offsets, s);
        prof.mark("layout_decode");
        if (e == hipSuccess) e = launch_rotor_decode_paged(d_in + pages_base, d_offsets, d_sizes, h.n_chunks, d_out, h.chunk_size, h.total_len, d_index,
                                                         reinterpret_cast<const uint32_t*>(d_in + dir_base), page_dir_words(paged_pages_per_chunk(h.chunk_size)),
                                                         (uint32_t)((h.container_len - pages_base) / kPageBytes), d_zmap, d_produced, d_err, s);
        prof.mark(decode_kernel_name(h.algo));
    } else {
    if (e == hipSuccess) e = launch_layout_decode(d_in, container_size, h.n_chunks, payload_base(h.n_chunks, h.total_len, with_index), d_sizes, d_offsets, d_err, s,
                                                  (h.flags & DENSITY_HIP_FLAG_SLOTTED) ? slot_stride(h.algo, h.chunk_size) : 0);
    prof.mark("layout_decode");
    if (e == hipSuccess) e = codec_decode(h.algo, d_in, d_offsets, d_sizes, h.n_chunks, d_out, h.chunk_size, h.total_len, true, d_index, d_produced, d_err, ws + p.off_tables, zmap_bytes(h.algo, h.n_chunks) ? reinterpret_cast<uint32_t*>(ws + p.off_zmap) : nullptr, s, d_pass);
    prof.mark(decode_kernel_name(h.algo));
    }
    if (e != hipSuccess) { set_error("kernel launch (decode)", e); return DENSITY_HIP_ERR_RUNTIME; }
    if (decoded_out) {
        uint32_t h_err = 0;
        e = hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("decode (device)", e); return DENSITY_HIP_ERR_RUNTIME; }
        if (h_err) { set_error("malformed or truncated container payload"); *decoded_out = 0; return DENSITY_HIP_ERR_FORMAT; }
        *decoded_out = h.total_len;
    }
    return DENSITY_HIP_OK;
}

// slotted container -> packed container (the wire form): header, size table and block index are copied, the payloads gathered
int run_pack_container(DeviceCtx* c, const uint8_t* d_in, size_t container_size, const density_hip_header_t& h, uint8_t* d_out, size_t cap, uint8_t* ws,
                       hipStream_t s, density_hip_header_t* header_out) {
    const DecodePlan p = plan_decode(h.algo, h.n_chunks);
    uint32_t* d_err = reinterpret_cast<uint32_t*>(ws + p.off_err);
    uint64_t* d_sizes = reinterpret_cast<uint64_t*>(ws + p.off_sizes);
    uint64_t* d_offsets = reinterpret_cast<uint64_t*>(ws + p.off_offsets);
    uint64_t* d_sizes64 = reinterpret_cast<uint64_t*>(ws + p.off_produced);           // (the u64 sizes the layout kernel wants)
    const bool with_index = h.flags & DENSITY_HIP_FLAG_BLOCK_INDEX;
    const uint64_t pbase = payload_base(h.n_chunks, h.total_len, with_index);
    if (cap < container_bound(h.algo, h.total_len, h.chunk_size)) { set_error("output capacity below density_hip_container_bound()"); return DENSITY_HIP_ERR_CAPACITY; }
    Profiler prof(c, s);
    hipError_t e = hipMemsetAsync(d_err, 0, sizeof(uint32_t), s);
    const uint64_t stride = (h.flags & DENSITY_HIP_FLAG_SLOTTED) ? slot_stride(h.algo, h.chunk_size) : 0;
    // sizes (u64) and source offsets from the slotted container's table, then the packed layout into the output
    if (e == hipSuccess) e = launch_layout_decode(d_in, container_size, h.n_chunks, pbase, d_sizes64, d_offsets, d_err, s, stride);
    if (e == hipSuccess) e = hipMemcpyAsync(d_out + sizeof(h), d_in + sizeof(h), pbase - sizeof(h), hipMemcpyDeviceToDevice, s);   // size table + block index
    density_hip_header_t out_h = h;
    out_h.flags = h.flags & ~DENSITY_HIP_FLAG_SLOTTED;
    out_h.container_len = 0;
    if (e == hipSuccess) e = launch_layout_encode(d_sizes64, h.n_chunks, out_h, pbase, d_out, cap, d_sizes /* packed offsets */, d_err, s);
    prof.mark("layout_encode");
    if (e == hipSuccess) {
        if (stride) e = launch_compact(d_in + pbase, stride, d_sizes64, d_sizes, h.n_chunks, d_out, d_err, s);
        else if (h.container_len > cap || h.container_len < pbase) { set_error("output capacity below the container's length"); return DENSITY_HIP_ERR_CAPACITY; }
        else e = hipMemcpyAsync(d_out + pbase, d_in + pbase, h.container_len - pbase, hipMemcpyDeviceToDevice, s);   // (already packed: its own bytes, no more)
    }
    prof.mark("compact");
    if (e != hipSuccess) { set_error("kernel launch (pack)", e); return DENSITY_HIP_ERR_RUNTIME; }
    if (header_out) {
        uint32_t h_err = 0;
        e = hipMemcpyAsync(header_out, d_out, sizeof(*header_out), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("pack (device)", e); return DENSITY_HIP_ERR_RUNTIME; }
        if (h_err) { set_error("malformed slotted container"); return DENSITY_HIP_ERR_FORMAT; }
    }
    return DENSITY_HIP_OK;
}

}  // namespace api
}  // namespace density

using namespace density;
using namespace density::api;

extern "C" {

// ---- section 2: container + device API ----
size_t density_hip_container_bound(int algo, size_t input_size, size_t chunk_size) {
    chunk_size = normalise_chunk(chunk_size, input_size, algo);
    if (!valid_algo(algo) || !valid_chunk(chunk_size)) return 0;
    return container_bound(algo, input_size, chunk_size);
}

size_t density_hip_encode_workspace_size(int algo, size_t input_size, size_t chunk_size) {
    chunk_size = normalise_chunk(chunk_size, input_size, algo);
    if (!valid_algo(algo) || !valid_chunk(chunk_size)) return 0;
    return plan_encode(algo, input_size, chunk_size).total;
}

size_t density_hip_decode_workspace_size(uint32_t n_chunks) {   // the largest of the three algorithms
    const size_t a = plan_decode(DENSITY_HIP_LION, n_chunks).total, b = plan_decode(DENSITY_HIP_CHAMELEON, n_chunks).total;
    return a > b ? a : b;
}

int density_hip_encode_device(int algo, const void* d_input, size_t input_size, void* d_output, size_t output_capacity,
                              size_t chunk_size, void* d_workspace, size_t workspace_size, void* stream,
                              density_hip_header_t* header_out) {
    g_last_error.clear();
    chunk_size = normalise_chunk(chunk_size, input_size, algo);
    if (!valid_algo(algo) || !valid_chunk(chunk_size) || (!d_input && input_size) || !d_output) { set_error("bad argument"); return DENSITY_HIP_ERR_ARGUMENT; }
    DeviceCtx* c = acquire_ctx();
    if (!c) return DENSITY_HIP_ERR_RUNTIME;
    std::lock_guard<std::mutex> lk(c->mu);
    const size_t need = plan_encode(algo, input_size, chunk_size).total;
    uint8_t* ws = (uint8_t*)d_workspace;
    if (ws) { if (workspace_size < need) { set_error("workspace too small"); return DENSITY_HIP_ERR_CAPACITY; } }
    else { hipError_t e = c->work.ensure(need); if (e != hipSuccess) { set_error("workspace allocation", e); return DENSITY_HIP_ERR_RUNTIME; } ws = (uint8_t*)c->work.p; }
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return run_encode_container(c, algo, (const uint8_t*)d_input, input_size, (uint8_t*)d_output, output_capacity, chunk_size, ws, s, header_out);
}

size_t density_hip_decode_workspace_size_for(int algo, size_t total_len, size_t chunk_size) {
    chunk_size = normalise_chunk(chunk_size, total_len, algo);
    if (!valid_algo(algo) || !valid_chunk(chunk_size)) return 0;
    return plan_decode(algo, chunk_count(total_len, chunk_size), chunk_size).total_with_passes;
}

int density_hip_encode_device_slotted(int algo, const void* d_input, size_t input_size, void* d_output, size_t output_capacity,
                                      size_t chunk_size, void* d_workspace, size_t workspace_size, void* stream,
                                      density_hip_header_t* header_out) {
    g_last_error.clear();
    chunk_size = normalise_chunk(chunk_size, input_size, algo);
    if (!valid_algo(algo) || !valid_chunk(chunk_size) || (!d_input && input_size) || !d_output) { set_error("bad argument"); return DENSITY_HIP_ERR_ARGUMENT; }
    DeviceCtx* c = acquire_ctx();
    if (!c) return DENSITY_HIP_ERR_RUNTIME;
    std::lock_guard<std::mutex> lk(c->mu);
    const size_t need = plan_encode(algo, input_size, chunk_size).total;
    uint8_t* ws = (uint8_t*)d_workspace;
    if (ws) { if (workspace_size < need) { set_error("workspace too small"); return DENSITY_HIP_ERR_CAPACITY; } }
    else { hipError_t e = c->work.ensure(need); if (e != hipSuccess) { set_error("workspace allocation", e); return DENSITY_HIP_ERR_RUNTIME; } ws = (uint8_t*)c->work.p; }
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return run_encode_container(c, algo, (const uint8_t*)d_input, input_size, (uint8_t*)d_output, output_capacity, chunk_size, ws, s, header_out, true);
}

int density_hip_encode_device_paged(int algo, const void* d_input, size_t input_size, void* d_output, size_t output_capacity,
                                    size_t chunk_size, void* d_workspace, size_t workspace_size, void* stream,
                                    density_hip_header_t* header_out) {
    g_last_error.clear();
    chunk_size = normalise_chunk(chunk_size, input_size, algo);
    if (!valid_algo(algo) || !valid_chunk(chunk_size) || (!d_input && input_size) || !d_output) { set_error("bad argument"); return DENSITY_HIP_ERR_ARGUMENT; }
    DeviceCtx* c = acquire_ctx();
    if (!c) return DENSITY_HIP_ERR_RUNTIME;
    std::lock_guard<std::mutex> lk(c->mu);
    const size_t need = plan_encode(algo, input_size, chunk_size).total;
    uint8_t* ws = (uint8_t*)d_workspace;
    if (ws) { if (workspace_size < need) { set_error("workspace too small"); return DENSITY_HIP_ERR_CAPACITY; } }
    else { hipError_t e = c->work.ensure(need); if (e != hipSuccess) { set_error("workspace allocation", e); return DENSITY_HIP_ERR_RUNTIME; } ws = (uint8_t*)c->work.p; }
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return run_encode_container(c, algo, (const uint8_t*)d_input, input_size, (uint8_t*)d_output, output_capacity, chunk_size, ws, s, header_out, false, true);
}
size_t density_hip_paged_pages_per_chunk(size_t chunk_size) { return valid_chunk(chunk_size) ? paged_pages_per_chunk(chunk_size) : 0; }
size_t density_hip_container_bound_paged(int algo, size_t input_size, size_t chunk_size) {
    if (!valid_algo(algo)) return 0;
    chunk_size = normalise_chunk(chunk_size, input_size, algo);
    if (!valid_chunk(chunk_size)) return 0;
    return container_bound_paged(algo, input_size, chunk_size);
}

size_t density_hip_container_bound_slotted(int algo, size_t input_size, size_t chunk_size) {
    chunk_size = normalise_chunk(chunk_size, input_size, algo);
    if (!valid_algo(algo) || !valid_chunk(chunk_size)) return 0;
    const size_t a = container_bound_slotted(algo, input_size, chunk_size), b = container_bound(algo, input_size, chunk_size);
    return a > b ? a : b;
}

int density_hip_pack_device(const void* d_container, size_t container_size, const density_hip_header_t* header, void* d_output,
                            size_t output_capacity, void* d_workspace, size_t workspace_size, void* stream, density_hip_header_t* header_out) {
    g_last_error.clear();
    if (!d_container || container_size < sizeof(density_hip_header_t) || !d_output) { set_error("bad argument"); return DENSITY_HIP_ERR_ARGUMENT; }
    DeviceCtx* c = acquire_ctx();
    if (!c) return DENSITY_HIP_ERR_RUNTIME;
    std::lock_guard<std::mutex> lk(c->mu);
    density_hip_header_t h;
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    if (header) h = *header;
    else {
        hipError_t e = hipMemcpyAsync(&h, d_container, sizeof(h), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("header read-back", e); return DENSITY_HIP_ERR_RUNTIME; }
    }
    if (check_header(h, container_size) != DENSITY_HIP_OK) { set_error("bad container header"); return DENSITY_HIP_ERR_FORMAT; }
    // a PAGED container is wire-ready as it stands, and its streams are not where the packed / slotted arithmetic looks for them
    if (h.flags & DENSITY_HIP_FLAG_PAGED) { set_error("density_hip_pack_device: a paged container is not packed (it is wire-ready; decode it or read its pages)"); return DENSITY_HIP_ERR_UNSUPPORTED; }
    const size_t need = plan_decode(h.algo, h.n_chunks).total;
    uint8_t* ws = (uint8_t*)d_workspace;
    if (ws) { if (workspace_size < need) { set_error("workspace too small"); return DENSITY_HIP_ERR_CAPACITY; } }
    else { hipError_t e = c->work.ensure(need); if (e != hipSuccess) { set_error("workspace allocation", e); return DENSITY_HIP_ERR_RUNTIME; } ws = (uint8_t*)c->work.p; }
    return run_pack_container(c, (const uint8_t*)d_container, container_size, h, (uint8_t*)d_output, output_capacity, ws, s, header_out);
}

int density_hip_decode_device(const void* d_container, size_t container_size, const density_hip_header_t* header, void* d_output,
                              size_t output_capacity, void* d_workspace, size_t workspace_size, void* stream, size_t* decoded_size_out) {
    g_last_error.clear();
    if (!d_container || container_size < sizeof(density_hip_header_t) || (!d_output && output_capacity)) { set_error("bad argument"); return DENSITY_HIP_ERR_ARGUMENT; }
    DeviceCtx* c = acquire_ctx();
    if (!c) return DENSITY_HIP_ERR_RUNTIME;
    std::lock_guard<std::mutex> lk(c->mu);
    density_hip_header_t h;
    if (header) h = *header;
    else {
        // ordered behind whatever produced the container on the caller's stream (streams here are non-blocking: a plain hipMemcpy is not)
        hipStream_t hs = stream ? (hipStream_t)stream : c->stream;
        hipError_t e = hipMemcpyAsync(&h, d_container, sizeof(h), hipMemcpyDeviceToHost, hs);
        if (e == hipSuccess) e = hipStreamSynchronize(hs);
        if (e != hipSuccess) { set_error("header read-back", e); return DENSITY_HIP_ERR_RUNTIME; }
    }
    if (check_header(h, container_size) != DENSITY_HIP_OK) { set_error("bad container header"); return DENSITY_HIP_ERR_FORMAT; }
    const DecodePlan dp = plan_decode(h.algo, h.n_chunks, h.chunk_size);
    const size_t need = dp.total;
    uint8_t* ws = (uint8_t*)d_workspace;
    if (ws) { if (workspace_size < need) { set_error("workspace too small"); return DENSITY_HIP_ERR_CAPACITY; } }
    else { hipError_t e = c->work.ensure(dp.total_with_passes); if (e != hipSuccess) { set_error("workspace allocation", e); return DENSITY_HIP_ERR_RUNTIME; } ws = (uint8_t*)c->work.p; workspace_size = c->work.cap; }
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return run_decode_container(c, (const uint8_t*)d_container, container_size, h, (uint8_t*)d_output, output_capacity, ws, s, decoded_size_out, workspace_size);
}

uint64_t density_hip_decode_pass_count(void) { return g_pass_decodes; }
void density_hip_stage_stats(uint64_t* out2) { if (out2) { out2[0] = density::g_stage_stats[0]; out2[1] = density::g_stage_stats[1]; } }
void density_hip_stream_stats(uint64_t* out4) { if (out4) for (int i = 0; i < 4; ++i) out4[i] = g_stream_stats[i]; }
size_t density_hip_auto_chunk(size_t input_size) { return auto_chunk(input_size); }
size_t density_hip_auto_chunk_for(int algo, size_t input_size) { return valid_algo(algo) ? auto_chunk(input_size, algo) : 0; }

void density_hip_set_profiling(int enabled) { g_profiling = enabled; }
void density_hip_set_kernel_variant(int variant) { g_variant = variant;   /* bit 3 (8): encode in batches with the stitch of one batch beside the encoding of the next */ density::g_force_simple = (variant & 1) != 0; density::g_force_pipeline = (variant & 4) != 0; density::g_force_lane_codec = (variant & 16) != 0; density::g_force_wave_codec = (variant & 32) != 0; density::g_stage_audit = (variant & 64) != 0; density::g_force_serial_decode = (variant & 128) != 0; density::g_serial_parse = (variant & 1024) != 0; density::g_chain_walk = (variant & 4096) != 0; density::g_lion_one_wave = (variant & 32768) != 0; density::g_walk_blocks = (variant & 8192) ? 1 : (variant & 16384) ? 4 : 2; density::g_rotor_split = density::kRotorSplitDefault != ((variant & 2048) != 0); }

int density_hip_last_timings(float* milliseconds, const char** names, int capacity) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return 0;
    DeviceCtx* c = &g_ctx[dev];
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->n_marks < 2) { c->n_marks = 0; return 0; }
    if (hipEventSynchronize(c->events[c->n_marks - 1]) != hipSuccess) { c->n_marks = 0; return 0; }
    int n = 0;
    for (size_t i = 1; i < c->n_marks && n < capacity; ++i) {
        if (!c->names[i]) continue;                       // start-of-call mark
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, c->events[i - 1], c->events[i]);
        if (milliseconds) milliseconds[n] = ms;
        if (names) names[n] = c->names[i];
        ++n;
    }
    c->n_marks = 0;
    return n;
}

int density_hip_selftest(void) {
    g_last_error.clear();
    return acquire_ctx() ? 0 : 1;
}

int density_hip_selftest_bits(void) {
    g_last_error.clear();
    int dev = -1;
    (void)acquire_ctx();
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices || !g_ctx[dev].ready) return -1;
    return (int)g_ctx[dev].selftest_bits;
}

void density_hip_shutdown(void) {
    for (int d = 0; d < kMaxDevices; ++d) {
        DeviceCtx* c = &g_ctx[d];
        std::lock_guard<std::mutex> lk(c->mu);
        if (!c->ready) continue;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || hipSetDevice(d) != hipSuccess) { (void)hipGetLastError(); continue; }
        (void)hipDeviceSynchronize();
        for (Buffer* b : {&c->work, &c->stage_in, &c->stage_out, &c->seg}) { if (b->p) (void)hipFree(b->p); b->p = nullptr; b->cap = 0; }
        if (c->pin_sizes) (void)hipHostFree(c->pin_sizes);
        c->pin_sizes = nullptr; c->pin_sizes_cap = 0;
        if (c->pin_meta) (void)hipHostFree(c->pin_meta);
        c->pin_meta = nullptr; c->pin_meta_cap = 0;
        for (hipEvent_t ev : c->pipe_events) (void)hipEventDestroy(ev);
        c->pipe_events.clear();
        for (hipEvent_t ev : c->events) (void)hipEventDestroy(ev);
        c->events.clear(); c->names.clear(); c->n_marks = 0;
        for (hipStream_t* sp : {&c->up, &c->down, &c->kern[0], &c->kern[1], &c->kern[2], &c->kern[3], &c->stream, &c->stitch_stream}) { if (*sp) (void)hipStreamDestroy(*sp); *sp = nullptr; }
        for (auto& ev : c->batch_done) { if (ev) (void)hipEventDestroy(ev); ev = nullptr; }
        if (c->stitch_done) (void)hipEventDestroy(c->stitch_done);
        c->stitch_done = nullptr;
        c->ready = false;                                                           // the next call sets the context up again (self-test included)
        (void)hipSetDevice(cur);
    }
}
const char* density_hip_last_error(void) { return g_last_error.c_str(); }
#ifndef DENSITY_HIP_KERNELS_ID
#define DENSITY_HIP_KERNELS_ID "unversioned"
#endif
const char* density_hip_version(void) { return "density_hip 0.2 (gfx950; reference: density-rs 0.16.6; kernels " DENSITY_HIP_KERNELS_ID ")"; }

}  // extern "C"
