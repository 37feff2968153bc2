// api.hip — the C ABI of libdensity_hip.so (include/density_hip.h): per-device context, workspace management,
// host-pointer staging, container assembly.  No CPU codec lives here: every byte is produced by the gfx950 kernels,
// and every entry point fails (returns 0 / an error code) when no usable HIP device is present.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/density_hip.h"
#include "kernels.hpp"

namespace {

using namespace density;

constexpr int kMaxDevices = 16;
constexpr size_t kAlign = 256;
constexpr size_t kMaxChunk = 1u << 30;   // u32 size table: a chunk stream must stay below 4 GiB

thread_local std::string g_last_error;
int g_profiling = 0;

void set_error(const char* what, hipError_t e = hipSuccess) {
    g_last_error = what;
    if (e != hipSuccess) { g_last_error += ": "; g_last_error += hipGetErrorString(e); }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// codec/codec.rs:18-21 with the geometry of chameleon.rs:138-146, cheetah.rs:188-196, lion.rs:317-325
inline size_t block_bytes(int algo) { return algo == DENSITY_HIP_CHAMELEON ? 256 : algo == DENSITY_HIP_CHEETAH ? 128 : 64; }
inline size_t sig_bytes(int algo) { return algo == DENSITY_HIP_LION ? 6 : 8; }
inline size_t safe_size(int algo, size_t n) {
    const size_t b = block_bytes(algo), s = sig_bytes(algo);
    return n + (n / b) * s + ((n % b) ? s : 0);
}
inline bool valid_algo(int algo) { return algo >= DENSITY_HIP_CHAMELEON && algo <= DENSITY_HIP_LION; }
// chunk_size 0 = automatic: one chunk is one work-group on one CU, so an input should be cut into at least as many chunks as the device has
// CUs (256) where that is possible without dropping below 64 KiB (small chunks restart the dictionary and cost ratio), and no finer than that
// (every chunk start costs a table clear and a few in-order rounds): never above 4 MiB, the largest chunk the index-fed decoder takes.
// Power of two: 10 MB -> 64 KiB (153 chunks), 100 MB -> 256 KiB (382), 256 MiB -> 1 MiB, >= 1 GiB -> 4 MiB.
// Lion runs one WAVE per chunk stream and is bound by memory latency per stream, not by a CU's LDS: it wants eight streams per CU (2048)
// and starts from 1 MiB.  Cheetah's decode passes (decode_passes.hip) walk one chunk per CU, in time proportional to the chunk: one
// chunk per CU exactly — the input over 256, up to whole 4 KiB trips of the encoder's passes — between 64 KiB and 1 MiB (100 MB -> 384 KiB:
// ratio 1.67 where 64 KiB chunks gave 1.34, and a faster round trip).
inline size_t auto_chunk(size_t n, int algo = DENSITY_HIP_CHAMELEON) {
    if (algo == DENSITY_HIP_CHEETAH) {
        size_t c = align_up((n + 255) / 256, 4096);
        if (c < (64u << 10)) c = 64u << 10;
        if (c > (1u << 20)) c = 1u << 20;
        return c;
    }
    const bool lds = algo == DENSITY_HIP_CHAMELEON;
    size_t c = lds ? (4u << 20) : (1u << 20);
    const size_t streams = lds ? 256 : 2048;
    while (c > (64u << 10) && n / c < streams) c >>= 1;
    return c;
}
inline size_t normalise_chunk(size_t chunk, size_t n, int algo = DENSITY_HIP_CHAMELEON) { return chunk == 0 ? auto_chunk(n, algo) : chunk; }
inline bool valid_chunk(size_t chunk) { return chunk >= 256 && chunk % 256 == 0 && chunk <= kMaxChunk; }
inline size_t chunk_count(size_t n, size_t chunk) { return (n + chunk - 1) / chunk; }
inline size_t index_base(size_t n_chunks) { return align_up(sizeof(density_hip_header_t) + 4 * n_chunks, 16); }
inline size_t index_bytes(size_t total_len, bool with_index) { return with_index ? (total_len + 255) / 256 : 0; }
inline size_t payload_base(size_t n_chunks, size_t total_len, bool with_index) { return align_up(index_base(n_chunks) + index_bytes(total_len, with_index), 16); }
int g_variant = 0;   // density_hip_set_kernel_variant
uint64_t g_pass_decodes = 0;                 // density_hip_decode_pass_count: Cheetah decodes served by the decode passes
uint64_t g_stream_stats[4] = {0, 0, 0, 0};   // density_hip_stream_stats: long streams encoded in segments | encode passes | decoded in segments | long streams decoded sequentially
inline bool want_index(int algo) { return algo == DENSITY_HIP_CHAMELEON && !(g_variant & 2); }
inline size_t slot_stride(int algo, size_t chunk) { return align_up(safe_size(algo, chunk), kAlign); }

struct Buffer {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        const size_t want = align_up(n + n / 8, 1 << 20);
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) { p = nullptr; return e; }
        cap = want;
        return hipSuccess;
    }
};

struct DeviceCtx {
    std::mutex mu;
    bool ready = false, selftest_ok = false;
    uint32_t selftest_bits = 0;
    hipStream_t stream = nullptr, stitch_stream = nullptr;   // stitch_stream: the compaction of one batch of chunks beside the encoding of the next
    hipEvent_t batch_done[8] = {}, stitch_done = nullptr;
    Buffer work, stage_in, stage_out, seg;   // seg: scratch of the segmented stream encode
    // the pipelined host-pointer container calls: an upload, a download and four kernel streams, events per slice, the slice sizes in pinned memory
    hipStream_t up = nullptr, down = nullptr, kern[4] = {};
    std::vector<hipEvent_t> pipe_events;
    uint64_t* pin_sizes = nullptr;
    size_t pin_sizes_cap = 0;
    // profiling: event marks accumulated since the last density_hip_last_timings() (name == nullptr opens a call)
    std::vector<hipEvent_t> events;
    std::vector<const char*> names;
    size_t n_marks = 0;
};
constexpr size_t kMaxMarks = 8192;

DeviceCtx g_ctx[kMaxDevices];

// Returns the context of the current device with its internal stream created and the LDS self-test passed.
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

struct Profiler {
    DeviceCtx* c;
    hipStream_t s;
    bool on;
    Profiler(DeviceCtx* ctx, hipStream_t stream) : c(ctx), s(stream), on(g_profiling != 0) { mark(nullptr); }
    void mark(const char* name) {
        if (!on) return;
        if (c->n_marks >= kMaxMarks) { on = false; return; }
        if (c->n_marks >= c->events.size()) {
            hipEvent_t ev;
            if (hipEventCreate(&ev) != hipSuccess) { on = false; return; }
            c->events.push_back(ev);
            c->names.push_back(nullptr);
        }
        c->names[c->n_marks] = name;
        (void)hipEventRecord(c->events[c->n_marks++], s);
    }
};

constexpr size_t kSerialSlots = 16384;   // concurrent chunk streams of the functional Cheetah/Lion kernels (one lane each; 12 / 28 GiB of tables when all are in use)
constexpr size_t kSerialTableBudget = 8ull << 30;   // ... but never more than 8 GiB of tables (the count comes from an untrusted header on decode): Cheetah 10922 streams, Lion 4681
inline size_t serial_slots(int algo, size_t n_chunks) {
    if (algo == DENSITY_HIP_CHAMELEON) return 0;
    const size_t by_memory = kSerialTableBudget / serial_table_bytes(algo);
    const size_t cap = kSerialSlots < by_memory ? kSerialSlots : by_memory;
    return n_chunks < cap ? n_chunks : cap;
}
inline size_t serial_tables(int algo, size_t n_chunks) { return algo == DENSITY_HIP_CHAMELEON ? 0 : align_up(serial_slots(algo, n_chunks) * serial_table_bytes(algo), kAlign); }

inline size_t zmap_bytes(int algo, size_t n_chunks) { return (algo == DENSITY_HIP_CHAMELEON && n_chunks <= kMaxPipelinedChunks) ? align_up((n_chunks ? n_chunks : 1) * kZmapWordsPerChunk * 4, kAlign) : 0; }

struct EncodePlan {
    size_t chunk, n_chunks, stride, off_err, off_sizes, off_offsets, off_slots, off_tables, off_zmap, off_stage, total;
};
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
    const bool passes = algo != DENSITY_HIP_CHAMELEON && p.n_chunks <= 0xffffffffull && stage_encode_eligible(algo, nullptr, n, chunk, (uint32_t)p.n_chunks);
    p.total = p.off_stage + (passes ? align_up(stage_scratch_bytes(algo, n, (uint32_t)p.n_chunks), kAlign) : 0);
    return p;
}
struct DecodePlan {
    size_t off_err, off_sizes, off_offsets, off_produced, off_tables, off_zmap, off_pass, total, total_with_passes;
};
// out_stride != 0 (the container's chunk size / a stream's output capacity): Cheetah's decode passes (decode_passes.hip) want a dword and
// a half per quad of scratch behind everything else; `total` is what the one-wave decoders need, `total_with_passes` what the passes need
DecodePlan plan_decode(int algo, size_t n_chunks, size_t out_stride = 0) {
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
    if (d_stage && stage_encode_eligible(algo, d_in, total, chunk_bytes, n_chunks))   // Cheetah / Lion: passes of ordered LDS exchanges, the one-wave kernels for what they hand back
        return launch_stage_encode(algo, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, (uint32_t)serial_slots(algo, n_chunks), d_stage, d_err, s);
    return launch_serial_encode(algo, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, (uint32_t)serial_slots(algo, n_chunks), s);
}
hipError_t codec_decode(int algo, const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out,
                        uint64_t out_stride, uint64_t out_total, bool exact, const uint8_t* d_index, uint64_t* d_produced, uint32_t* d_err,
                        uint8_t* d_tables, uint32_t* d_zmap, hipStream_t s, uint8_t* d_pass = nullptr) {
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

int check_header(const density_hip_header_t& h, size_t container_size) {
    if (h.magic != DENSITY_HIP_MAGIC || h.version != 1 || !valid_algo(h.algo)) return DENSITY_HIP_ERR_FORMAT;
    if (!valid_chunk(h.chunk_size)) return DENSITY_HIP_ERR_FORMAT;
    if (h.n_chunks != chunk_count(h.total_len, h.chunk_size)) return DENSITY_HIP_ERR_FORMAT;
    if (h.flags & ~(DENSITY_HIP_FLAG_BLOCK_INDEX | DENSITY_HIP_FLAG_SLOTTED)) return DENSITY_HIP_ERR_FORMAT;
    if (h.container_len > container_size || h.container_len < payload_base(h.n_chunks, h.total_len, h.flags & DENSITY_HIP_FLAG_BLOCK_INDEX)) return DENSITY_HIP_ERR_FORMAT;
    return DENSITY_HIP_OK;
}

// ---- device-side drivers (ctx already acquired; `ws` points at a workspace of sufficient size) ----

int run_encode_container(DeviceCtx* c, int algo, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, size_t chunk,
                         uint8_t* ws, hipStream_t s, density_hip_header_t* header_out, bool slotted = false) {
    const EncodePlan p = plan_encode(algo, n, chunk);
    if (p.n_chunks > 0xffffffffull) { set_error("too many chunks"); return DENSITY_HIP_ERR_ARGUMENT; }
    if (p.n_chunks <= 1) slotted = false;                                              // (one chunk encodes straight into place either way)
    if (cap < (slotted ? container_bound_slotted(algo, n, chunk) : container_bound(algo, n, chunk))) { set_error("output capacity below density_hip_container_bound()"); return DENSITY_HIP_ERR_CAPACITY; }
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
    if (p.n_chunks == 1) {
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
                         size_t cap, uint8_t* ws, hipStream_t s, size_t* decoded_out, size_t ws_size = 0) {
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
    if (e == hipSuccess) e = launch_layout_decode(d_in, container_size, h.n_chunks, payload_base(h.n_chunks, h.total_len, with_index), d_sizes, d_offsets, d_err, s,
                                                  (h.flags & DENSITY_HIP_FLAG_SLOTTED) ? slot_stride(h.algo, h.chunk_size) : 0);
    prof.mark("layout_decode");
    if (e == hipSuccess) e = codec_decode(h.algo, d_in, d_offsets, d_sizes, h.n_chunks, d_out, h.chunk_size, h.total_len, true, d_index, d_produced, d_err, ws + p.off_tables, zmap_bytes(h.algo, h.n_chunks) ? reinterpret_cast<uint32_t*>(ws + p.off_zmap) : nullptr, s, d_pass);
    prof.mark(decode_kernel_name(h.algo));
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
        else e = hipMemcpyAsync(d_out + pbase, d_in + pbase, container_size - pbase, hipMemcpyDeviceToDevice, s);
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

}  // namespace

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
struct PinnedInPlace {
    void* p = nullptr;
    PinnedInPlace(const void* q, size_t n) {
        if (q && n && hipHostRegister(const_cast<void*>(q), n, hipHostRegisterDefault) == hipSuccess) p = const_cast<void*>(q);
        else (void)hipGetLastError();
    }
    ~PinnedInPlace() { if (p) (void)hipHostUnregister(p); }
    explicit operator bool() const { return p != nullptr; }
};
constexpr uint32_t kPipeMaxSlices = 48;
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
    if (!pipe_wanted(h.algo, total, chunk, nc) || (h.flags & DENSITY_HIP_FLAG_SLOTTED)) return 0;
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
    if (e == hipSuccess && input_size) e = hipMemcpyAsync(c->stage_in.p, input, input_size, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { set_error("staging (H2D)", e); return 0; }
    density_hip_header_t h;
    if (run_encode_container(c, algo, (const uint8_t*)c->stage_in.p, input_size, (uint8_t*)c->stage_out.p, bound, chunk_size, (uint8_t*)c->work.p, c->stream, &h) != DENSITY_HIP_OK) return 0;
    if (h.container_len > output_size) { set_error("output buffer too small"); return 0; }
    e = hipMemcpy(output, c->stage_out.p, h.container_len, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("staging (D2H)", e); return 0; }
    return (size_t)h.container_len;
}

uint64_t density_hip_decode_pass_count(void) { return g_pass_decodes; }
void density_hip_stage_stats(uint64_t* out2) { if (out2) { out2[0] = density::g_stage_stats[0]; out2[1] = density::g_stage_stats[1]; } }
void density_hip_stream_stats(uint64_t* out4) { if (out4) for (int i = 0; i < 4; ++i) out4[i] = g_stream_stats[i]; }
size_t density_hip_auto_chunk(size_t input_size) { return auto_chunk(input_size); }
size_t density_hip_auto_chunk_for(int algo, size_t input_size) { return valid_algo(algo) ? auto_chunk(input_size, algo) : 0; }

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
    if (e == hipSuccess) e = hipMemcpyAsync(c->stage_in.p, container, h.container_len, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { set_error("staging (H2D)", e); return 0; }
    size_t produced = 0;
    if (run_decode_container(c, (const uint8_t*)c->stage_in.p, h.container_len, h, (uint8_t*)c->stage_out.p, h.total_len, (uint8_t*)c->work.p, c->stream, &produced, c->work.cap) != DENSITY_HIP_OK) return 0;
    e = hipMemcpy(output, c->stage_out.p, produced, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { set_error("staging (D2H)", e); return 0; }
    return produced;
}

void density_hip_set_profiling(int enabled) { g_profiling = enabled; }
void density_hip_set_kernel_variant(int variant) { g_variant = variant;   /* bit 3 (8): encode in batches with the stitch of one batch beside the encoding of the next */ density::g_force_simple = (variant & 1) != 0; density::g_force_pipeline = (variant & 4) != 0; density::g_force_lane_codec = (variant & 16) != 0; density::g_force_wave_codec = (variant & 32) != 0; density::g_stage_audit = (variant & 64) != 0; density::g_force_serial_decode = (variant & 128) != 0; }

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
