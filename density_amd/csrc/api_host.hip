// api_host.hip — the host-pointer container calls (include/density_hip.h: density_hip_encode / density_hip_decode / density_hip_decoded_size):
// staged whole through device memory, or — Chameleon inputs worth three slices — pipelined in slices with the caller's buffers pinned in place.
#include "api_internal.hpp"

namespace density {
namespace api {

bool pipe_streams(DeviceCtx* c, uint32_t n_events) {
    hipError_t e = hipSuccess;
    if (!c->up) {
        e = hipStreamCreateWithFlags(&c->up, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->down, hipStreamNonBlocking);
        for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipStreamCreateWithFlags(&c->kern[i], hipStreamNonBlocking);
    }
    while (e == hipSuccess && c->pipe_events.size() < n_events) {
        hipEvent_t ev;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e == hipSuccess) c->pipe_events.push_back(ev);
    }
    if (e != hipSuccess) { set_error("pipelined host path: streams / events", e); return false; }
    return true;
}
hipError_t pin_meta_ensure(DeviceCtx* c, size_t bytes) {
    if (bytes <= c->pin_meta_cap) return hipSuccess;
    if (c->pin_meta) (void)hipHostFree(c->pin_meta);
    c->pin_meta = nullptr; c->pin_meta_cap = 0;
    const size_t want = align_up(bytes + bytes / 2, 4096);
    const hipError_t e = hipHostMalloc((void**)&c->pin_meta, want, hipHostMallocDefault);
    if (e == hipSuccess) c->pin_meta_cap = want;
    return e;
}

namespace {

// ---------------------------------------------------------------------------------------------------------------
// The host-pointer container calls, pipelined (Chameleon, inputs worth three slices and more: pipe_wanted).
// The caller's buffers are pinned in place for the duration of the call (hipHostRegister: microseconds on this platform,
// probes/host_register.hip), so that copies from and to them are asynchronous; the input goes up in slices of chunks on one stream,
// every slice is encoded / decoded on one of four kernel streams as soon as it has arrived — chunks are independent, the slices'
// kernels run side by side — and its result goes down on a third stream while later slices are still on their way up.  A PCIe link
// moves 56 GB/s one way and 46 each way at once (probes/pcie_duplex.hip): a call that moves N up and E down in sequence cannot do
// better than N / (N + E) x 56 = 35 GB/s; overlapped it is bound by the larger of the two.  Where pinning fails (memory that is
// already registered, read-only mappings) the plain staged path below is taken.
// ---------------------------------------------------------------------------------------------------------------
// Slices of whole chunks.  A slice's kernel takes as long as ONE chunk takes (0.11 ms per MiB of chunk: chunks run side by side, a chunk is a
// chain) and the kernels of different slices mostly queue up behind one another (the streams share a few hardware queues), so a slice
// must be worth ~10 chunk lengths of transfer or the kernels, not the link, set the pace: a twelfth of the input, ten chunks, 2 MiB at least —
// and the call is pipelined only where that makes three slices or more (measured, 4 MiB chunks: 64 MiB staged 28 / 29 GB/s, pipelined
// in slices of two chunks 31 / 21; 256 MiB 33 / 33 -> 46 / 45; 1 GiB 34 / 35 -> 50 / 48).
inline size_t pipe_slice_bytes(size_t total, size_t chunk) {
    size_t target = total / 12;
    if (target < 10 * chunk) target = 10 * chunk;
    if (target < (2u << 20)) target = 2u << 20;
    if (g_variant & 256) target = chunk;                                              // (tests: a slice per chunk, whatever the size)
    return target;
}
inline bool pipe_wanted(int algo, size_t n, size_t chunk, size_t n_chunks) {
    if (algo != DENSITY_HIP_CHAMELEON || n_chunks < 4 || (g_variant & 512)) return false;
    return (g_variant & 256) || (n >= (32u << 20) && n >= 3 * pipe_slice_bytes(n, chunk));
}
inline uint32_t pipe_slice_chunks(size_t total, size_t chunk, size_t n_chunks) {
    const size_t target = pipe_slice_bytes(total, chunk);
    size_t per = (target + chunk - 1) / chunk;
    if (per < 1) per = 1;
    while ((n_chunks + per - 1) / per > kPipeMaxSlices) ++per;
    return (uint32_t)per;
}

// returns bytes decoded, 0 with the error set; *handled = false: not taken (the caller falls back to the staged path)
size_t decode_container_pipelined(DeviceCtx* c, const uint8_t* container, const density_hip_header_t& h, uint8_t* output, bool* handled) {
    *handled = false;
    const uint32_t nc = h.n_chunks;
    const size_t chunk = h.chunk_size, total = h.total_len;
    if (!pipe_wanted(h.algo, total, chunk, nc) || (h.flags & (DENSITY_HIP_FLAG_SLOTTED | DENSITY_HIP_FLAG_PAGED))) return 0;   // (slices follow the PACKED layout: a paged blob's streams lie in pages — the staged call reads them in place)
    PinnedInPlace pin_in(container, h.container_len), pin_out(output, total);
    if (!pin_in || !pin_out) return 0;
    const bool with_index = h.flags & DENSITY_HIP_FLAG_BLOCK_INDEX;
    const size_t pbase = payload_base(nc, total, with_index);
    // where every chunk stream lies: the size table, read here on the host (the device's layout pass reads and checks it again)
    std::vector<uint64_t> offs(nc + 1);
    uint64_t off = pbase;
    for (uint32_t i = 0; i < nc; ++i) {
        uint32_t sz;
        std::memcpy(&sz, container + sizeof(density_hip_header_t) + 4 * (size_t)i, 4);
        offs[i] = off;
        if (sz > h.container_len || off > h.container_len - sz) { *handled = true; set_error("malformed or truncated container payload"); return 0; }
        off += sz;
        if (i + 1 < nc) off = align_up(off, 16);
    }
    offs[nc] = off;
    const uint32_t per = pipe_slice_chunks(total, chunk, nc), slices = (nc + per - 1) / per;
    if (!pipe_streams(c, 2 + 2 * slices)) return 0;
    const DecodePlan p = plan_decode(h.algo, nc, chunk);
    hipError_t e = c->stage_in.ensure(h.container_len);
    if (e == hipSuccess) e = c->stage_out.ensure(total);
    if (e == hipSuccess) e = c->work.ensure(p.total);
    if (e != hipSuccess) { set_error("staging buffers", e); return 0; }
    *handled = true;
    uint8_t* d_in = (uint8_t*)c->stage_in.p;
    uint8_t* d_out = (uint8_t*)c->stage_out.p;
    uint8_t* ws = (uint8_t*)c->work.p;
    uint32_t* d_err = reinterpret_cast<uint32_t*>(ws + p.off_err);
    uint64_t* d_sizes = reinterpret_cast<uint64_t*>(ws + p.off_sizes);
    uint64_t* d_offsets = reinterpret_cast<uint64_t*>(ws + p.off_offsets);
    uint64_t* d_produced = reinterpret_cast<uint64_t*>(ws + p.off_produced);
    uint32_t* d_zmap = zmap_bytes(h.algo, nc) ? reinterpret_cast<uint32_t*>(ws + p.off_zmap) : nullptr;
    const uint8_t* d_index = with_index ? d_in + index_base(nc) : nullptr;
    hipStream_t s = c->stream;
    hipEvent_t ev_head = c->pipe_events[0], ev_layout = c->pipe_events[1];
    e = hipMemsetAsync(d_err, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipMemcpyAsync(d_in, container, pbase, hipMemcpyHostToDevice, c->up);          // header, size table, block index
    if (e == hipSuccess) e = hipEventRecord(ev_head, c->up);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, ev_head, 0);
    if (e == hipSuccess) e = launch_layout_decode(d_in, h.container_len, nc, pbase, d_sizes, d_offsets, d_err, s, 0);
    if (e == hipSuccess) e = hipEventRecord(ev_layout, s);
    for (uint32_t k = 0; k < slices && e == hipSuccess; ++k) {
        const uint32_t first = k * per, count = first + per <= nc ? per : nc - first;
        hipEvent_t ev_up = c->pipe_events[2 + 2 * k], ev_dec = c->pipe_events[3 + 2 * k];
        hipStream_t ks = c->kern[k & 3u];
        e = hipMemcpyAsync(d_in + offs[first], container + offs[first], offs[first + count] - offs[first], hipMemcpyHostToDevice, c->up);
        if (e == hipSuccess) e = hipEventRecord(ev_up, c->up);
        if (e == hipSuccess) e = hipStreamWaitEvent(ks, ev_layout, 0);
        if (e == hipSuccess) e = hipStreamWaitEvent(ks, ev_up, 0);
        const uint64_t out_off = (uint64_t)first * chunk;
        if (e == hipSuccess) e = codec_decode(h.algo, d_in, d_offsets + first, d_sizes + first, count, d_out + out_off, chunk, total - out_off, true,
                                              d_index ? d_index + out_off / 256 : nullptr, d_produced + first, d_err, nullptr,
                                              d_zmap ? d_zmap + (uint64_t)first * kZmapWordsPerChunk : nullptr, ks);
        if (e == hipSuccess) e = hipEventRecord(ev_dec, ks);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->down, ev_dec, 0);
        const uint64_t bytes = total - out_off < (uint64_t)count * chunk ? total - out_off : (uint64_t)count * chunk;
        if (e == hipSuccess) e = hipMemcpyAsync(output + out_off, d_out + out_off, bytes, hipMemcpyDeviceToHost, c->down);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, ev_dec, 0);
    }
    uint32_t h_err = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost, s);
    // (always drained: nothing may still be reading or writing the caller's buffers when they are unpinned)
    const hipError_t e1 = hipStreamSynchronize(c->up), e2 = hipStreamSynchronize(s), e3 = hipStreamSynchronize(c->down);
    hipError_t e4 = hipSuccess;
    for (int i = 0; i < 4; ++i) { const hipError_t x = hipStreamSynchronize(c->kern[i]); if (x != hipSuccess) e4 = x; }
    if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e2 != hipSuccess ? e2 : e3 != hipSuccess ? e3 : e4;
    if (e != hipSuccess) { set_error("decode (pipelined host path)", e); return 0; }
    if (h_err) { set_error("malformed or truncated container payload"); return 0; }
    return total;
}

size_t encode_container_pipelined(DeviceCtx* c, int algo, const uint8_t* input, size_t n, uint8_t* output, size_t cap, size_t chunk, bool* handled) {
    *handled = false;
    const EncodePlan p = plan_encode(algo, n, chunk);
    const size_t nc = p.n_chunks;
    if (!pipe_wanted(algo, n, chunk, nc) || nc > 0xffffffffull) return 0;
    const bool with_index = want_index(algo);
    const size_t pbase = payload_base(nc, n, with_index), bound = container_bound(algo, n, chunk);
    if (cap < pbase) return 0;                                                        // (the staged path reports it)
    PinnedInPlace pin_in(input, n), pin_out(output, std::min(cap, bound));            // (what the container can reach, not the caller's whole capacity)
    if (!pin_in || !pin_out) return 0;
    const uint32_t per = pipe_slice_chunks(n, chunk, nc), slices = (uint32_t)((nc + per - 1) / per);
    if (!pipe_streams(c, 3 * slices)) return 0;
    hipError_t e = c->stage_in.ensure(n);
    if (e == hipSuccess) e = c->stage_out.ensure(bound);
    if (e == hipSuccess) e = c->work.ensure(p.total);
    if (e == hipSuccess && c->pin_sizes_cap < slices) {
        if (c->pin_sizes) (void)hipHostFree(c->pin_sizes);
        c->pin_sizes = nullptr; c->pin_sizes_cap = 0;
        e = hipHostMalloc((void**)&c->pin_sizes, 8 * (size_t)(kPipeMaxSlices + 16), hipHostMallocDefault);
        if (e == hipSuccess) c->pin_sizes_cap = kPipeMaxSlices + 16;
    }
    if (e != hipSuccess) { set_error("staging buffers", e); return 0; }
    *handled = true;
    uint8_t* d_in = (uint8_t*)c->stage_in.p;
    uint8_t* d_out = (uint8_t*)c->stage_out.p;                                       // the packed container, assembled on the device slice by slice
    uint8_t* d_index = with_index ? d_out + index_base(nc) : nullptr;
    uint8_t* ws = (uint8_t*)c->work.p;
    uint32_t* d_err = reinterpret_cast<uint32_t*>(ws + p.off_err);
    uint64_t* d_sizes = reinterpret_cast<uint64_t*>(ws + p.off_sizes);
    uint64_t* d_offsets = reinterpret_cast<uint64_t*>(ws + p.off_offsets);
    uint64_t* d_carry = d_offsets + nc;                                               // (the extra entry of the offsets array: the running end)
    uint8_t* d_slots = ws + p.off_slots;
    uint32_t* d_zmap = zmap_bytes(algo, nc) ? reinterpret_cast<uint32_t*>(ws + p.off_zmap) : nullptr;
    density_hip_header_t hdr{};
    hdr.magic = DENSITY_HIP_MAGIC; hdr.algo = (uint8_t)algo; hdr.version = 1; hdr.flags = with_index ? DENSITY_HIP_FLAG_BLOCK_INDEX : 0;
    hdr.chunk_size = (uint32_t)chunk; hdr.n_chunks = (uint32_t)nc; hdr.total_len = n; hdr.container_len = 0;
    hipStream_t s = c->stream;
    e = hipMemsetAsync(d_err, 0, sizeof(uint32_t), s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);                                 // (the kernel streams below do not wait for s)
    // Per slice: its input goes up, its chunks are encoded into their slots, the slice's place in the packed container follows from the running
    // end of its predecessor (write_buffer.rs:29-31's running total: the one sequential step, a scan over a few sizes), its streams are gathered
    // there, and the new running end comes back to the host, which then knows what to bring down.
    for (uint32_t k = 0; k < slices && e == hipSuccess; ++k) {
        const uint32_t first = k * per, count = first + per <= nc ? per : (uint32_t)(nc - first);
        const uint64_t in_off = (uint64_t)first * chunk, len = n - in_off < (uint64_t)count * chunk ? n - in_off : (uint64_t)count * chunk;
        hipEvent_t ev_up = c->pipe_events[3 * k], ev_lay = c->pipe_events[3 * k + 1], ev_enc = c->pipe_events[3 * k + 2];
        hipStream_t ks = c->kern[k & 3u];
        e = hipMemcpyAsync(d_in + in_off, input + in_off, len, hipMemcpyHostToDevice, c->up);
        if (e == hipSuccess) e = hipEventRecord(ev_up, c->up);
        if (e == hipSuccess) e = hipStreamWaitEvent(ks, ev_up, 0);
        if (e == hipSuccess) e = codec_encode(algo, d_in + in_off, len, chunk, count, d_slots + (uint64_t)first * p.stride, p.stride, d_sizes + first,
                                              d_index ? d_index + in_off / 256 : nullptr, nullptr, d_zmap ? d_zmap + (uint64_t)first * kZmapWordsPerChunk : nullptr, nullptr, d_err, ks);
        if (e == hipSuccess && k) e = hipStreamWaitEvent(ks, c->pipe_events[3 * (k - 1) + 1], 0);   // the predecessor's running end
        if (e == hipSuccess) e = launch_layout_encode_batch(d_sizes, first, count, k == 0, k + 1 == slices, hdr, pbase, d_out, bound, d_offsets, d_carry, d_err, ks);
        if (e == hipSuccess) e = hipMemcpyAsync(c->pin_sizes + k, d_carry, 8, hipMemcpyDeviceToHost, ks);
        if (e == hipSuccess) e = hipEventRecord(ev_lay, ks);
        if (e == hipSuccess) e = launch_compact(d_slots + (uint64_t)first * p.stride, p.stride, d_sizes + first, d_offsets + first, count, d_out, d_err, ks, k + 1 < slices);
        if (e == hipSuccess) e = hipEventRecord(ev_enc, ks);
    }
    uint64_t begin = pbase, end = pbase;
    bool too_small = false;
    for (uint32_t k = 0; k < slices && e == hipSuccess && !too_small; ++k) {
        e = hipEventSynchronize(c->pipe_events[3 * k + 2]);
        if (e != hipSuccess) break;
        end = c->pin_sizes[k];
        if (end > cap || end > bound || end < begin) { too_small = true; break; }
        e = hipMemcpyAsync(output + begin, d_out + begin, end - begin, hipMemcpyDeviceToHost, c->down);
        begin = align_up(end, 16);
        if (k + 1 < slices) {
            if (begin > cap) { too_small = true; break; }
            std::memset(output + end, 0, begin - end);                                // the gap behind a slice's last stream (the gather zeroes those inside a slice)
        }
    }
    const hipError_t e1 = hipStreamSynchronize(c->up);
    hipError_t e4 = hipSuccess;
    for (int i = 0; i < 4; ++i) { const hipError_t x = hipStreamSynchronize(c->kern[i]); if (x != hipSuccess) e4 = x; }
    if (e == hipSuccess && !too_small) e = hipMemcpyAsync(output, d_out, pbase, hipMemcpyDeviceToHost, c->down);   // header (written with the last slice), size table, block index
    const hipError_t e3 = hipStreamSynchronize(c->down);
    if (e == hipSuccess) e = e1 != hipSuccess ? e1 : e4 != hipSuccess ? e4 : e3;
    uint32_t h_err = 0;
    if (e == hipSuccess) e = hipMemcpy(&h_err, d_err, sizeof(h_err), hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("encode (pipelined host path)", e); return 0; }
    if (h_err & 16u) { set_error("encode: device-side watchdog"); return 0; }
    if (h_err || too_small) { set_error("output buffer too small"); return 0; }
    return (size_t)end;
}

}  // namespace
}  // namespace api
}  // namespace density

using namespace density;
using namespace density::api;

extern "C" {

size_t density_hip_encode(int algo, const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size, size_t chunk_size) {
    g_last_error.clear();
    chunk_size = normalise_chunk(chunk_size, input_size, algo);
    if (!valid_algo(algo) || !valid_chunk(chunk_size) || (!input && input_size) || !output) { set_error("bad argument"); return 0; }
    DeviceCtx* c = acquire_ctx();
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    {
        bool handled = false;
        const size_t r = encode_container_pipelined(c, algo, input, input_size, output, output_size, chunk_size, &handled);
        if (handled) return r;
    }
    const size_t bound = container_bound(algo, input_size, chunk_size);
    hipError_t e = c->stage_in.ensure(input_size ? input_size : 1);
    if (e == hipSuccess) e = c->stage_out.ensure(bound);
    if (e == hipSuccess) e = c->work.ensure(plan_encode(algo, input_size, chunk_size).total);
    if (e == hipSuccess && input_size) e = copy_host_side_pinned(c->stage_in.p, input, input_size, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { set_error("staging (H2D)", e); return 0; }
    density_hip_header_t h;
    if (run_encode_container(c, algo, (const uint8_t*)c->stage_in.p, input_size, (uint8_t*)c->stage_out.p, bound, chunk_size, (uint8_t*)c->work.p, c->stream, &h) != DENSITY_HIP_OK) return 0;
    if (h.container_len > output_size) { set_error("output buffer too small"); return 0; }
    e = copy_host_side_pinned(output, c->stage_out.p, h.container_len, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) { set_error("staging (D2H)", e); return 0; }
    return (size_t)h.container_len;
}

size_t density_hip_decoded_size(const uint8_t* container, size_t container_size) {
    if (!container || container_size < sizeof(density_hip_header_t)) return 0;
    density_hip_header_t h;
    std::memcpy(&h, container, sizeof(h));
    return check_header(h, container_size) == DENSITY_HIP_OK ? (size_t)h.total_len : 0;
}

size_t density_hip_decode(const uint8_t* container, size_t container_size, uint8_t* output, size_t output_size) {
    g_last_error.clear();
    if (!container || container_size < sizeof(density_hip_header_t) || (!output && output_size)) { set_error("bad argument"); return 0; }
    density_hip_header_t h;
    std::memcpy(&h, container, sizeof(h));
    if (check_header(h, container_size) != DENSITY_HIP_OK) { set_error("bad container header"); return 0; }
    if (h.total_len > output_size) { set_error("output buffer too small"); return 0; }
    if (h.total_len == 0) return 0;
    DeviceCtx* c = acquire_ctx();
    if (!c) return 0;
    std::lock_guard<std::mutex> lk(c->mu);
    {
        bool handled = false;
        const size_t r = decode_container_pipelined(c, container, h, output, &handled);
        if (handled) return r;
    }
    hipError_t e = c->stage_in.ensure(h.container_len);
    if (e == hipSuccess) e = c->stage_out.ensure(h.total_len);
    if (e == hipSuccess) e = c->work.ensure(plan_decode(h.algo, h.n_chunks, h.chunk_size).total_with_passes);
    if (e == hipSuccess) e = copy_host_side_pinned(c->stage_in.p, container, h.container_len, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { set_error("staging (H2D)", e); return 0; }
    size_t produced = 0;
    if (run_decode_container(c, (const uint8_t*)c->stage_in.p, h.container_len, h, (uint8_t*)c->stage_out.p, h.total_len, (uint8_t*)c->work.p, c->stream, &produced, c->work.cap) != DENSITY_HIP_OK) return 0;
    e = copy_host_side_pinned(output, c->stage_out.p, produced, hipMemcpyDeviceToHost, c->stream);
    if (e != hipSuccess) { set_error("staging (D2H)", e); return 0; }
    return produced;
}

}  // extern "C"
