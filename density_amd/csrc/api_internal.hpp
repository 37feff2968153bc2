// api_internal.hpp — what the three units of the C ABI share (api.hip: per-device context, plans, the container's device-side drivers and
// entry points; api_stream.hip: ONE reference stream — the reference's nine symbols, the parallel segments of long Chameleon streams;
// api_host.hip: the host-pointer container calls, staged or pipelined in slices).  Internal to libdensity_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/density_hip.h"
#include "kernels.hpp"

namespace density {
namespace api {

constexpr int kMaxDevices = 16;
constexpr size_t kAlign = 256;
constexpr size_t kMaxChunk = 1u << 30;   // u32 size table: a chunk stream must stay below 4 GiB

extern thread_local std::string g_last_error;
extern int g_profiling;
extern int g_variant;                 // density_hip_set_kernel_variant
extern uint64_t g_pass_decodes;       // density_hip_decode_pass_count: Cheetah decodes served by the decode passes
extern uint64_t g_stream_stats[4];    // density_hip_stream_stats: long streams encoded in segments | encode passes | decoded in segments | long streams decoded sequentially
void set_error(const char* what, hipError_t e = hipSuccess);

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
// 10 MB -> 64 KiB (156 chunks), 100 MB -> 384 KiB (255), 256 MiB -> 1 MiB, 1 GiB -> 4 MiB, 1.5 GiB -> 3 MiB (512).
// Lion runs one WAVE per chunk stream and is bound by memory latency per stream, not by a CU's LDS (below).  Cheetah's decode passes (decode_passes.hip) walk one chunk per CU, in time proportional to the chunk: one
// chunk per CU exactly — the input over 256, up to whole 4 KiB trips of the encoder's passes — between 64 KiB and 1 MiB (100 MB -> 384 KiB:
// ratio 1.67 where 64 KiB chunks gave 1.34, and a faster round trip).
inline size_t auto_chunk(size_t n, int algo = DENSITY_HIP_CHAMELEON) {
    if (algo == DENSITY_HIP_CHEETAH) {
        size_t c = align_up((n + 255) / 256, 4096);
        if (c < (64u << 10)) c = 64u << 10;
        if (c > (1u << 20)) c = 1u << 20;
        return c;
    }
    if (algo == DENSITY_HIP_CHAMELEON) {
        // (round 4: not a power of two any more.  A chunk is a work-group is a CU, so what counts is WHOLE WAVES of 256 chunks: 100 MB in 382 chunks
        // of 256 KiB is two rounds of work-groups, the second half empty; in 255 chunks of 384 KiB it is one — a quarter less time, and a better
        // ratio.  The fewest whole waves of chunks of at most 4 MiB, the input spread evenly over them in whole rounds of 16 blocks.)
        if (n <= 256u * (size_t)(64u << 10)) return 64u << 10;
        const size_t waves = (n + 256u * (size_t)(4u << 20) - 1) / (256u * (size_t)(4u << 20));
        size_t c = align_up((n + 256 * waves - 1) / (256 * waves), 4096);
        if (c < (64u << 10)) c = 64u << 10;
        if (c > (4u << 20)) c = 4u << 20;
        return c;
    }
    // Lion: one wave per chunk stream, bound by memory latency per stream — the largest power of two that still gives the device 700 streams (about three per CU)
    // (round 4, with two records per step: 100 MB in 128 KiB chunks runs as fast as round 3's 64 KiB chunks did, at ratio 1.41 instead of 1.28)
    size_t c = 1u << 20;
    while (c > (64u << 10) && n / c < 700) c >>= 1;
    return c;
}
inline size_t normalise_chunk(size_t chunk, size_t n, int algo = DENSITY_HIP_CHAMELEON) { return chunk == 0 ? auto_chunk(n, algo) : chunk; }
inline bool valid_chunk(size_t chunk) { return chunk >= 256 && chunk % 256 == 0 && chunk <= kMaxChunk; }
inline size_t chunk_count(size_t n, size_t chunk) { return (n + chunk - 1) / chunk; }
inline size_t index_base(size_t n_chunks) { return align_up(sizeof(density_hip_header_t) + 4 * n_chunks, 16); }
inline size_t index_bytes(size_t total_len, bool with_index) { return with_index ? (total_len + 255) / 256 : 0; }
inline size_t payload_base(size_t n_chunks, size_t total_len, bool with_index) { return align_up(index_base(n_chunks) + index_bytes(total_len, with_index), 16); }
inline bool want_index(int algo) { return algo == DENSITY_HIP_CHAMELEON && !(g_variant & 2); }
// paged container (DENSITY_HIP_FLAG_PAGED): behind the block index the page directory (per chunk: 16 bytes {n_pages}, then 16 bytes per page), then,
// on the next page-size boundary of the container, the pages
inline size_t paged_dir_base(size_t n_chunks, size_t total_len) { return payload_base(n_chunks, total_len, true); }
inline uint32_t paged_pages_per_chunk(size_t chunk) { return pages_per_chunk(safe_size(DENSITY_HIP_CHAMELEON, chunk)); }
inline size_t paged_dir_bytes(size_t n_chunks, size_t chunk) { return 4 * (size_t)page_dir_words(paged_pages_per_chunk(chunk)) * n_chunks; }
inline size_t paged_pages_base(size_t n_chunks, size_t total_len, size_t chunk) { return align_up(paged_dir_base(n_chunks, total_len) + paged_dir_bytes(n_chunks, chunk), kAlign); }
// what the paged form is for: Chameleon, chunks of 1 MiB and more (a chunk's last page is half empty on average: 3 % of a MiB of text), 32-bit positions
// — and of at most 4 MiB: only the index-fed rotation decoder reads pages in place (kPagedMaxChunk = its kRotMaxBlocks blocks, its directory copy holds
// kPagedMaxPages pages), and the library writes no container it cannot read
inline bool paged_eligible(int algo, size_t n, size_t chunk) {
    const size_t nc = chunk_count(n, chunk);
    return algo == DENSITY_HIP_CHAMELEON && want_index(algo) && nc > 1 && chunk >= (1u << 20) && chunk <= kPagedMaxChunk && paged_pages_per_chunk(chunk) <= kPagedMaxPages &&
           (uint64_t)nc * paged_pages_per_chunk(chunk) * kPageBytes < (1ull << 32);
}
size_t container_bound_paged(int algo, size_t n, size_t chunk);
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
    uint8_t* pin_meta = nullptr;             // the pipelined stream calls: per segment sizes, offsets, states as the device reports them
    size_t pin_meta_cap = 0;
    // profiling: event marks accumulated since the last density_hip_last_timings() (name == nullptr opens a call)
    std::vector<hipEvent_t> events;
    std::vector<const char*> names;
    size_t n_marks = 0;
};
constexpr size_t kMaxMarks = 8192;

extern DeviceCtx g_ctx[kMaxDevices];
// Returns the context of the current device with its internal stream created and the LDS self-test passed.
DeviceCtx* acquire_ctx();

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

// A caller's buffer pinned in place for the duration of a call (hipHostRegister: microseconds on this platform, probes/host_register.hip), so that
// copies from and to it are asynchronous.  Where pinning fails (memory that is already registered, read-only mappings) the staged paths are taken.
// What a successful registration is worth: the runtime keeps pins of its own pageable copies in a cache, and a register over a larger extent of a
// buffer it has such a pin for can "succeed" with the cached pin's pages and no more — the pages behind them stay unmapped and the first access
// faults the GPU (round 4).  Round 5 looked for a check (probes/host_register_trap.hip, profiles/r05_probe_host_register_trap.txt): for a clean and
// for a suspect registration alike hipHostGetDevicePointer translates the first and the last byte, hipMemGetAddressRange reports base 0 and the
// full span, hipPointerGetAttributes the same type and pointers — nothing the API says tells them apart, and the minimal sequence (a pageable copy
// of part of a buffer, then the register) does not fault by itself.  So the defence stays where round 4 put it: THIS library never lets the runtime
// pin a caller's memory (copy_host_side_pinned below), and the translation of both ends is checked because it is free.  A caller whose OTHER
// libraries copy part of a buffer pageably and then hand us the whole of it can still meet the runtime's cache; bounce buffers of our own would
// close that at the price of a CPU copy of every byte (10 GB/s against the 40-50 the pipelined calls move).
struct PinnedInPlace {
    void* p = nullptr;
    PinnedInPlace(const void* q, size_t n) {
        if (!q || !n) return;
        void* host = const_cast<void*>(q);
        if (hipHostRegister(host, n, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return; }
        void *d0 = nullptr, *d1 = nullptr;
        const bool whole = hipHostGetDevicePointer(&d0, host, 0) == hipSuccess &&
                           hipHostGetDevicePointer(&d1, static_cast<uint8_t*>(host) + (n - 1), 0) == hipSuccess &&
                           static_cast<uint8_t*>(d1) - static_cast<uint8_t*>(d0) == (ptrdiff_t)(n - 1);
        if (whole) p = host;
        else { (void)hipGetLastError(); (void)hipHostUnregister(host); (void)hipGetLastError(); }
    }
    ~PinnedInPlace() { if (p) (void)hipHostUnregister(p); }
    PinnedInPlace(const PinnedInPlace&) = delete;
    PinnedInPlace& operator=(const PinnedInPlace&) = delete;
    explicit operator bool() const { return p != nullptr; }
};
// One copy between the caller's memory and the device on stream s, complete on return.  A megabyte and more: the caller's side is pinned in place for
// the copy's duration by US — never by the runtime's own pageable-copy path, which pins the caller's pages too and keeps the pin in a cache of its own
// (the last eight per queue) beyond the call.  A later hipHostRegister of the same buffer over a LARGER extent is then answered with the cached pin's
// pages and no more: the pages behind them are not mapped, and the first access faults the GPU (round 4: "memory access fault ... write access to a
// read-only page" on the last page of an output buffer — a staged decode had brought down a few bytes less into it than the pipelined call that
// followed registered; in a DENSITY_HIP_DEBUG build DENSITY_HIP_RAW_STAGED=1 brings the old copies back, tools/gpu_host_stream_sequence.py walks the sequence).
inline hipError_t copy_host_side_pinned(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const void* host = kind == hipMemcpyHostToDevice ? src : dst;
    if (n >= (1u << 20) && !debug_env("DENSITY_HIP_RAW_STAGED")) {      // (the switch — debug builds only — is the round-4 fault's reproducer, tools/gpu_host_stream_sequence.py)
        PinnedInPlace pin(host, n);
        if (pin) {
            const hipError_t e = hipMemcpyAsync(dst, src, n, kind, s), e2 = hipStreamSynchronize(s);
            return e != hipSuccess ? e : e2;
        }
    }
    const hipError_t e = hipMemcpyAsync(dst, src, n, kind, s), e2 = hipStreamSynchronize(s);
    return e != hipSuccess ? e : e2;
}
constexpr uint32_t kPipeMaxSlices = 48;
// the upload, download and kernel streams of the pipelined host-pointer calls and n_events events (api_host.hip)
bool pipe_streams(DeviceCtx* c, uint32_t n_events);
hipError_t pin_meta_ensure(DeviceCtx* c, size_t bytes);

constexpr size_t kSerialSlots = 16384;   // concurrent chunk streams of the functional Cheetah/Lion kernels (one lane each; 12 / 28 GiB of tables when all are in use)
constexpr size_t kSerialTableBudget = 8ull << 30;   // ... but never more than 8 GiB of tables (the count comes from an untrusted header on decode): Cheetah 10922 streams, Lion 4681
inline size_t serial_slots(int algo, size_t n_chunks) {
    if (algo == DENSITY_HIP_CHAMELEON) return 0;
    const size_t by_memory = kSerialTableBudget / serial_table_bytes(algo);
    size_t cap = kSerialSlots < by_memory ? kSerialSlots : by_memory;
    if (const char* e = debug_env("DENSITY_HIP_SERIAL_SLOTS")) { const size_t v = (size_t)atoll(e); if (v >= 1 && v < cap) cap = v; }   // (debug builds: streams in flight, tools/gpu_lion_slots.py)
    return n_chunks < cap ? n_chunks : cap;
}
inline size_t serial_tables(int algo, size_t n_chunks) { return algo == DENSITY_HIP_CHAMELEON ? 0 : align_up(serial_slots(algo, n_chunks) * serial_table_bytes(algo), kAlign); }

inline size_t zmap_bytes(int algo, size_t n_chunks) { return (algo == DENSITY_HIP_CHAMELEON && n_chunks <= kMaxPipelinedChunks) ? align_up((n_chunks ? n_chunks : 1) * kZmapWordsPerChunk * 4, kAlign) : 0; }

struct EncodePlan {
    size_t chunk, n_chunks, stride, off_err, off_sizes, off_offsets, off_slots, off_tables, off_zmap, off_stage, total;
};EncodePlan plan_encode(int algo, size_t n, size_t chunk);
struct DecodePlan {
    size_t off_err, off_sizes, off_offsets, off_produced, off_tables, off_zmap, off_pass, total, total_with_passes;
};// out_stride != 0 (the container's chunk size / a stream's output capacity): Cheetah's decode passes (decode_passes.hip) want a dword and
// a half per quad of scratch behind everything else; `total` is what the one-wave decoders need, `total_with_passes` what the passes need
DecodePlan plan_decode(int algo, size_t n_chunks, size_t out_stride = 0);

// algorithm dispatch
hipError_t codec_encode(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out,
                        uint64_t out_stride, uint64_t* d_sizes, uint8_t* d_index, uint8_t* d_tables, uint32_t* d_zmap, uint8_t* d_stage, uint32_t* d_err, hipStream_t s);
hipError_t codec_decode(int algo, const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out,
                        uint64_t out_stride, uint64_t out_total, bool exact, const uint8_t* d_index, uint64_t* d_produced, uint32_t* d_err,
                        uint8_t* d_tables, uint32_t* d_zmap, hipStream_t s, uint8_t* d_pass = nullptr);
const char* encode_kernel_name(int algo);
const char* decode_kernel_name(int algo);
size_t container_bound(int algo, size_t n, size_t chunk);
size_t container_bound_slotted(int algo, size_t n, size_t chunk);
int check_header(const density_hip_header_t& h, size_t container_size);

// device-side drivers of the container (api.hip; ctx already acquired; `ws` points at a workspace of sufficient size)
int run_encode_container(DeviceCtx* c, int algo, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, size_t chunk,
                         uint8_t* ws, hipStream_t s, density_hip_header_t* header_out, bool slotted = false, bool paged = false);
int run_decode_container(DeviceCtx* c, const uint8_t* d_in, size_t container_size, const density_hip_header_t& h, uint8_t* d_out,
                         size_t cap, uint8_t* ws, hipStream_t s, size_t* decoded_out, size_t ws_size = 0);
// ... of one reference stream (api_stream.hip)
int run_stream_encode(DeviceCtx* c, int algo, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, uint8_t* ws, hipStream_t s, size_t* size_out);
int run_stream_decode(DeviceCtx* c, int algo, const uint8_t* d_in, size_t n, uint8_t* d_out, size_t cap, uint8_t* ws, hipStream_t s, size_t* size_out);

}  // namespace api
}  // namespace density
