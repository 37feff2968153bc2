/*
 * density_hip.h — C ABI of libdensity_hip.so, the MI355X (gfx950) implementation of density's
 * 4-byte-word dictionary-hash encode/decode hot path.
 *
 * Section 1 is the drop-in boundary: the nine `extern "C"` symbols the reference crate (density-rs 0.16.6)
 * itself exports, with identical names, signatures and return convention.  A reference-side FFI (Rust
 * `extern "C"` block, cgo, ctypes ...) binds exactly these; see INTEGRATION.md.
 *
 * Section 2 is additive: the chunked container that makes the path data-parallel (each chunk is an independent
 * reference stream), device-pointer + stream variants, and profiling hooks.  Nothing in section 2 changes the
 * meaning of section 1.
 *
 * All citations are file:line in the reference tree (src/...).
 */
#ifndef DENSITY_HIP_H
#define DENSITY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------------------------
 * 1. Reference-compatible symbols (host pointers, single reference-format stream, bit-exact with the crate).
 *    (One stream is one dependency chain; long Chameleon streams are still encoded and decoded in parallel segments on
 *    the device, with byte-identical results: DESIGN.md 4.7.)
 *
 *    Return value: bytes written; 0 on any failure (the reference maps Err to 0 via unwrap_or(0) and otherwise
 *    panics on a short buffer or truncated input; this library returns 0 instead and never writes past
 *    output_size).  An empty input also returns 0, as in the reference.
 *    `output_size` must be >= {algo}_safe_encode_buffer_size(input_size) for encode and >= the original length
 *    for decode (the stream does not carry its decoded length; codec/codec.rs:72-80).
 *
 *    These calls stage through device memory (H2D, kernels, D2H) on a per-device internal stream and are
 *    thread-safe.  How one stream is spread over the device: chameleon_encode / _decode of a few MiB
 *    and more in ~256 parallel segments; cheetah_encode / lion_encode of 64 / 192 KiB and more in passes of ordered LDS
 *    exchanges, cheetah_decode of 64 KiB and more in decode passes (everything but its chain of contexts in parallel);
 *    lion_decode and short streams on one wave or one work-group.  The data-parallel path proper is the container API
 *    of section 2.
 *    SLOWER THAN THE CRATE on one CPU core (10 MB of prose, buffers on the device, profiles/r06_benches_density.txt): cheetah_decode
 *    ~0.32 GB/s against 1.4 (its chain of contexts is one team of four waves on one CU), lion_decode ~0.09 GB/s against 0.84 (two waves,
 *    tables in memory); cheetah_encode 3.2 against 1.0 and lion_encode 1.0 against 0.65 only just win.  One stream is one dependency
 *    chain: a caller with more than one stream's worth of data wants the container calls, which are what this library is for.
 * ---------------------------------------------------------------------------------------------------------- */

/* algorithms/chameleon/chameleon.rs:70-73 */
size_t chameleon_encode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size);
/* algorithms/chameleon/chameleon.rs:75-78 */
size_t chameleon_decode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size);
/* algorithms/chameleon/chameleon.rs:80-83 -> codec/codec.rs:18-21 */
size_t chameleon_safe_encode_buffer_size(size_t size);

/* algorithms/cheetah/cheetah.rs:105-108 */
size_t cheetah_encode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size);
/* algorithms/cheetah/cheetah.rs:110-113 */
size_t cheetah_decode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size);
/* algorithms/cheetah/cheetah.rs:115-118 */
size_t cheetah_safe_encode_buffer_size(size_t size);

/* algorithms/lion/lion.rs:193-196 */
size_t lion_encode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size);
/* algorithms/lion/lion.rs:198-201 */
size_t lion_decode(const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size);
/* algorithms/lion/lion.rs:203-206 */
size_t lion_safe_encode_buffer_size(size_t size);

/* ------------------------------------------------------------------------------------------------------------
 * 2. Additive extensions (not in the reference).
 * ---------------------------------------------------------------------------------------------------------- */

enum { DENSITY_HIP_CHAMELEON = 0, DENSITY_HIP_CHEETAH = 1, DENSITY_HIP_LION = 2 };

enum {
    DENSITY_HIP_OK = 0,
    DENSITY_HIP_ERR_ARGUMENT = 1,     /* bad algo / chunk size / null pointer */
    DENSITY_HIP_ERR_CAPACITY = 2,     /* output or workspace too small */
    DENSITY_HIP_ERR_FORMAT = 3,       /* container header or payload malformed / truncated */
    DENSITY_HIP_ERR_RUNTIME = 4,      /* HIP runtime error, no gfx950 device, failed self-test */
    DENSITY_HIP_ERR_UNSUPPORTED = 5   /* reserved: algorithm not available on the device path */
};

/*
 * Chunked container ("DHC1").  The input is cut into chunks of `chunk_size` bytes (a multiple of 256, i.e. of
 * every algorithm's block size); chunk i is encoded as an independent reference stream — fresh zero tables and a
 * fresh ProtectionState, exactly what {Algo}::encode(&input[i*C..]) returns (chameleon.rs:45-48) — so chunks
 * encode and decode in parallel.  Layout (little-endian):
 *     [0,32)                 density_hip_header_t
 *     [32, 32+4*n_chunks)    u32 encoded size of each chunk payload
 *     (16-byte aligned)      optional block index, one byte per 256-byte input block (flags & DENSITY_HIP_FLAG_BLOCK_INDEX)
 *     payload i starts at the next 16-byte boundary after payload i-1 (payload 0 after the tables)
 * With one chunk (chunk_size >= total_len) the single payload IS the reference stream of the whole input.
 *
 * Block index (Chameleon): the reference stream marks neither record boundaries nor raw-copy blocks — both sides re-derive
 * them by walking the records and running the same ProtectionState FSM (codec/codec.rs:35-37 vs :89-91), a serial
 * pointer chase.  The index stores what that walk would find: byte b describes input block b (numbered over the whole
 * input): bit 7 = raw-copy block; bits 0..6 = number of MAP flags of the block's signature, 0..64 (record length
 * = 8 + 256 - 2*n), or 0x7f for the chunk's ragged last block (< 256 bytes).  It is redundant metadata —
 * payloads stay plain reference streams — and a container without it (flags == 0, e.g. one assembled by a CPU producer)
 * decodes through the record-walking path.
 */
#define DENSITY_HIP_FLAG_BLOCK_INDEX 1u
/* Slotted container (device-resident form): payload i is NOT packed behind payload i-1 but stays in the worst-case slot the encoder
 * wrote it to, at payload_base + i * slot_stride, slot_stride = round_up({algo}_safe_encode_buffer_size(chunk_size), 256); the size
 * table says how much of each slot is stream; container_len is the end of the last payload.  The chunk streams themselves are the
 * same bytes.  This is what density_hip_encode_device_slotted() produces and what density_hip_decode_device() also accepts: the
 * gather into the packed form (one more pass over every encoded byte: write_buffer.rs:29-31's running total has no parallel
 * equivalent until the sizes exist) is left to the moment the container leaves the device — density_hip_pack_device(), or the
 * host-pointer density_hip_encode(), which always returns the packed form. */
#define DENSITY_HIP_FLAG_SLOTTED 2u
/* Paged container (round 5; Chameleon, with the block index): the wire form WITHOUT a stitch pass.  The encoder places the streams itself, in
 * pages of DENSITY_HIP_PAGE_BYTES taken from one counter as the chunks ask for them: a chunk's stream leaves a page when the records of its next
 * round of 16 blocks (4 KiB of input, at most 4224 bytes) would not end inside it, so page changes fall on multiples of 16 blocks and a page's
 * unused tail is at most a round (2 % of a page on text).  Layout behind the block index:
 *     (16-byte aligned)   page directory, per chunk 16 * (1 + P) bytes, P = density_hip_paged_pages_per_chunk(chunk_size):
 *                             {u32 n_pages, 0, 0, 0}, then per page of the chunk, in stream order,
 *                             {u32 page, u32 first input block of the chunk coded there, u32 bytes of stream in the page, 0}
 *     (256-byte aligned)  page 0, page 1, ...   (container_len = this base + DENSITY_HIP_PAGE_BYTES * pages in use)
 * Chunk i's reference stream = the first `bytes` of each of its pages, concatenated; the u32 size table holds the sum.  A CPU reader does that and
 * calls the crate; density_hip_decode_device() reads the pages in place. */
#define DENSITY_HIP_FLAG_PAGED 4u
#define DENSITY_HIP_PAGE_BYTES 65536u
#define DENSITY_HIP_MAGIC 0x31434844u /* "DHC1" */
#define DENSITY_HIP_DEFAULT_CHUNK (1u << 20)

typedef struct density_hip_header {
    uint32_t magic;          /* DENSITY_HIP_MAGIC */
    uint8_t  algo;           /* DENSITY_HIP_CHAMELEON ... */
    uint8_t  version;        /* 1 */
    uint16_t flags;          /* DENSITY_HIP_FLAG_* */
    uint32_t chunk_size;     /* bytes of input per chunk, multiple of 256 */
    uint32_t n_chunks;       /* ceil(total_len / chunk_size) */
    uint64_t total_len;      /* decoded length in bytes */
    uint64_t container_len;  /* total container length in bytes (header + table + padded payloads) */
} density_hip_header_t;

/* chunk_size 0 in the calls below means density_hip_auto_chunk_for(algo, input_size); for Chameleon (density_hip_auto_chunk): a chunk is one work-group
 * on one CU, small chunks restart the dictionary and cost ratio, every chunk start costs a table clear — so: 64 KiB up to 16 MiB of input, beyond
 * that the fewest whole waves of 256 chunks of at most 4 MiB with the input spread evenly over them in whole 4 KiB rounds (10 MB -> 64 KiB,
 * 100 MB -> 384 KiB = 255 chunks, 256 MiB -> 1 MiB, 1 GiB -> 4 MiB, 1.5 GiB -> 3 MiB = 512 chunks). */
size_t density_hip_auto_chunk(size_t input_size);
/* The same per algorithm, 64 KiB .. 1 MiB: Lion (one wave per chunk stream, memory-latency bound) takes the largest power of two that
 * still gives the device 700 streams, about three per CU (100 MB -> 128 KiB); Cheetah (decode passes: one chunk's chain of contexts per CU, in time proportional to the chunk)
 * one chunk per CU: the input over 256, rounded up to 4 KiB (100 MB -> 384 KiB). */
size_t density_hip_auto_chunk_for(int algo, size_t input_size);

/* Upper bound of the container size for `input_size` bytes (0 if the arguments are invalid). */
size_t density_hip_container_bound(int algo, size_t input_size, size_t chunk_size);

/* Host-pointer container codec (H2D, kernels, D2H). Return bytes written, 0 on failure.
 * Chameleon inputs worth three slices or more (32 MiB at least; a slice is a twelfth of the input, ten chunks at least) are pipelined: the
 * caller's buffers are pinned in place for the duration of the call (hipHostRegister), slices of chunks go up on one stream, through the
 * kernels on others, and down on a third: 46-52 GB/s at 256 MiB .. 1 GiB where the whole buffer staged in sequence gives 33-35.  The
 * container is byte for byte the same either way.  Where pinning fails the staged path is taken. */
size_t density_hip_encode(int algo, const uint8_t* input, size_t input_size, uint8_t* output, size_t output_size,
                          size_t chunk_size);
size_t density_hip_decode(const uint8_t* container, size_t container_size, uint8_t* output, size_t output_size);
/* Reads total_len from a host-resident container header (0 if malformed). */
size_t density_hip_decoded_size(const uint8_t* container, size_t container_size);

/* Device-resident container codec: pointers are device pointers on the current HIP device, `stream` is a
 * hipStream_t (NULL = the library's internal stream).  Work is enqueued on `stream`.
 *   - workspace: pass a device buffer of at least density_hip_{encode,decode}_workspace_size() bytes, or NULL to
 *     use the library's per-device cached workspace (then calls on different streams must not overlap).
 *     The encode workspace holds a worst-case slot per chunk (about 1.06 x the input) and, for Cheetah / Lion, a table slot per
 *     concurrent chunk stream (768 KiB / 1.75 MiB, at most 8 GiB) plus the scratch of the exchange passes (a dword per quad and
 *     the per-block masks: about 1.25 / 1.5 x the input).
 *   - header_out / decoded_size_out: optional HOST pointers; when non-NULL the call synchronises `stream` and
 *     fills them (and then also reports kernel-detected format errors).  When NULL the call is fully asynchronous.
 * Returns DENSITY_HIP_OK or an error code. */
size_t density_hip_encode_workspace_size(int algo, size_t input_size, size_t chunk_size);
size_t density_hip_decode_workspace_size(uint32_t n_chunks);
/* The same for a container of known shape: includes the scratch of Cheetah's decode passes (a dword and a half per quad: 1.5 x total_len
 * + 1/32; decode_passes.hip).  A workspace of only density_hip_decode_workspace_size() bytes still decodes — Cheetah then on one wave per
 * chunk stream. */
size_t density_hip_decode_workspace_size_for(int algo, size_t total_len, size_t chunk_size);
int density_hip_encode_device(int algo, const void* d_input, size_t input_size, void* d_output, size_t output_capacity,
                              size_t chunk_size, void* d_workspace, size_t workspace_size, void* stream,
                              density_hip_header_t* header_out);
/* The same, leaving every chunk stream in its slot inside the output (DENSITY_HIP_FLAG_SLOTTED): no stitch pass.  `output_capacity` must be at
 * least density_hip_container_bound_slotted().  With one chunk the result is the ordinary (packed) container. */
size_t density_hip_container_bound_slotted(int algo, size_t input_size, size_t chunk_size);
int density_hip_encode_device_slotted(int algo, const void* d_input, size_t input_size, void* d_output, size_t output_capacity,
                                      size_t chunk_size, void* d_workspace, size_t workspace_size, void* stream,
                                      density_hip_header_t* header_out);
/* The same as a PAGED container (DENSITY_HIP_FLAG_PAGED above): wire-ready without a stitch pass.  Chameleon inputs of two and more chunks of 1 MiB and
 * more and less than about 3 GiB; anything else comes out slotted, as from density_hip_encode_device_slotted() — the header's flags say which.
 * `output_capacity` must be at least density_hip_container_bound_paged() (the pages every chunk could need; text ends far below it). */
size_t density_hip_container_bound_paged(int algo, size_t input_size, size_t chunk_size);
size_t density_hip_paged_pages_per_chunk(size_t chunk_size);
int density_hip_encode_device_paged(int algo, const void* d_input, size_t input_size, void* d_output, size_t output_capacity,
                                    size_t chunk_size, void* d_workspace, size_t workspace_size, void* stream,
                                    density_hip_header_t* header_out);
/* Slotted (or packed) container -> packed container: byte for byte what density_hip_encode_device() writes for the same input.  Workspace as for
 * decode.  With header_out == NULL the call is asynchronous and reports nothing about the container it was given (a size table that does not
 * fit its slots or the container leaves the output unwritten): pass header_out to have it validated. */
int density_hip_pack_device(const void* d_container, size_t container_size, const density_hip_header_t* header, void* d_output,
                            size_t output_capacity, void* d_workspace, size_t workspace_size, void* stream, density_hip_header_t* header_out);
/* `header` may be NULL: it is then read back from the device (one small synchronous copy). */
int density_hip_decode_device(const void* d_container, size_t container_size, const density_hip_header_t* header,
                              void* d_output, size_t output_capacity, void* d_workspace, size_t workspace_size,
                              void* stream, size_t* decoded_size_out);

/* Device-resident single reference stream (the format of section 1, device pointers).  `size_out` is a HOST
 * pointer and must be non-NULL: the call synchronises `stream`. */
int density_hip_stream_encode_device(int algo, const void* d_input, size_t input_size, void* d_output,
                                     size_t output_capacity, void* stream, size_t* size_out);
int density_hip_stream_decode_device(int algo, const void* d_input, size_t input_size, void* d_output,
                                     size_t output_capacity, void* stream, size_t* size_out);

/* Profiling: when enabled, the device entry points bracket each kernel with HIP events on the launch stream.
 * density_hip_last_timings() synchronises the last event and returns the duration of every kernel launched on the
 * current device since the previous density_hip_last_timings() call (at most 8192 marks are kept), in launch order;
 * names[i] points to a static string.  Returns the number of entries written (<= capacity).  (The pipelined host-pointer container calls
 * run their slices on streams of their own and record no marks.) */
void density_hip_set_profiling(int enabled);
int density_hip_last_timings(float* milliseconds, const char** names, int capacity);

/* Test hook: how the reference-shaped Chameleon stream calls of a few MiB and more were served so far (process-wide counters):
 * out4[0] streams encoded in parallel segments, [1] passes those encodes took (1 per stream if every speculation held),
 * [2] streams decoded in parallel segments, [3] long streams decoded sequentially (mostly raw copies, or buffers the parallel path does not take). */
void density_hip_stream_stats(uint64_t* out4);
/* ... how many Cheetah decodes the decode passes (decode_passes.hip) have served so far (process-wide) ... */
uint64_t density_hip_decode_pass_count(void);
/* ... and of Cheetah container encodes under kernel variant bit 64: out2[0] chunks that went through the exchange passes, [1] how many
 * of them were handed back to the in-order kernel (raw-copy blocks, a ragged end). */
void density_hip_stage_stats(uint64_t* out2);

/* Test hook, bit mask: 1 = force the simple one-wavefront-per-chunk kernels, 2 = encode containers without the block index,
 * 4 = force the 16-wave role pipelines (chameleon.hip) instead of the default wave-rotation kernels (rotor.hip),
 * 8 = encode in batches with the stitch of one batch beside the encoding of the next, 16 = Cheetah / Lion on the
 * one-lane-per-stream kernels instead of the one-wave-per-stream kernels (serial_codec.hip), 32 = Cheetah containers on the
 * one-wave-per-stream encoder instead of the exchange passes (exchange_stages.hip), 64 = count the chunks the exchange passes
 * keep / hand back (density_hip_stage_stats; reads the verdicts back, so the encode call synchronises), 128 = Cheetah containers on the
 * one-wave-per-stream decoder instead of the decode passes (decode_passes.hip), 256 = the host-pointer container calls pipelined
 * whatever the size, a slice per chunk, 512 = never pipelined (below; the reference symbols on long streams too), 1024 = Cheetah's decode passes find
 * a chunk's records by the one-wave walk alone (no window kernels), 2048 = the other rotation encoder (8 chain + 8 emit waves), 4096 = Cheetah's decode
 * passes walk the contexts run by run on one wave (round 5's walk) instead of by a team of four waves, 8192 / 16384 = on ONE wave, 64 / 128 quads at a time,
 * 32768 = Lion's container decode on one wave per stream (round 4's) instead of two.
 * Payload bytes are identical in every variant. */
void density_hip_set_kernel_variant(int variant);

/* Runs the LDS ordering self-tests the kernels rely on (also run lazily before first use). 0 = the library can run.
 * density_hip_selftest_bits() returns the raw failure mask (0 = everything passed, -1 = no usable device): bits 0..7 plain 16-bit
 * LDS write order (fatal), bits 8..11 lane order of the ordered exchange ds_mskor_rtn_b32 (only the one-wavefront kernels run),
 * bits 12..13 lane-reversed rollback / token hand-off behind the exchanges (the 16-wave role pipelines run instead of the
 * wave-rotation kernels). */
int density_hip_selftest(void);
int density_hip_selftest_bits(void);
/*
 * Multi-GPU placement (the arithmetic of density_amd/parallel.py for callers below Python).  The path shards by chunks with no
 * data-path collective: rank g of G encodes the chunk range density_hip_shard_range() gives it into a container of its own; ONE all-gather of
 * three u64 per rank — {chunks, payload bytes, input bytes} of the local container, the caller's collective (RCCL ncclAllGather over xGMI) —
 * then tells every rank, through density_hip_global_layout(), where its size-table entries, its block-index slice and its payload region sit in the
 * global container (every shard but the last covers whole chunks, so index slices concatenate; every payload region but the last non-empty one
 * is padded to 16 bytes).  Pure host arithmetic: no device, no HIP call.  Both return DENSITY_HIP_OK or DENSITY_HIP_ERR_ARGUMENT.
 */
typedef struct density_hip_shard {
    uint64_t chunk_first, chunk_end;   /* chunks [first, end) of the global input: contiguous, balanced to within one chunk */
    uint64_t byte_first, byte_end;     /* the same range in input bytes (chunk-aligned; the last shard ends at total_len) */
} density_hip_shard_t;
int density_hip_shard_range(size_t total_len, size_t chunk_size, uint32_t rank, uint32_t world, density_hip_shard_t* out);
typedef struct density_hip_global_layout {
    uint64_t n_chunks, total_len;                       /* of the global container */
    uint64_t index_at, index_bytes, payload_at;         /* its block index (index_bytes == 0 without DENSITY_HIP_FLAG_BLOCK_INDEX) and payload area */
    uint64_t container_len;
    uint64_t chunk_offset, payload_offset, input_offset; /* of `rank`: first size-table entry; bytes from payload_at; input bytes before it */
    uint64_t payload_bytes_padded;                      /* this rank's payload region as it sits in the global container */
} density_hip_global_layout_t;
int density_hip_global_layout(const uint64_t* chunks, const uint64_t* payload_bytes, const uint64_t* input_bytes, uint32_t world, uint32_t rank,
                              uint32_t flags, density_hip_global_layout_t* out);

/*
 * Config 5's wire form: the MULTI-RANK container "DHCM" (round 6).  Every rank's own container — packed, slotted or PAGED, whatever the rank made: each is a
 * self-describing DHC1 blob of that rank's shard — travels as it stands; a 32-byte super-header and one 24-byte row per rank say where each blob lies:
 *
 *     [0,32)               density_hip_multi_header_t {magic "DHCM", version 1, algo, flags 0, n_ranks, chunk_size, total_len, container_len}
 *     [32, 32 + 24 R)      per rank {offset, length, input_bytes}: the blob's place from the start of the super-container (256-byte aligned), its length,
 *                          the input bytes it decodes to (rank r's output starts at the sum of the input_bytes before it)
 *     (256-byte aligned)   blob 0, blob 1, ...
 *
 * The only collective is still one all-gather of two u64 per rank {container length, input bytes} (ncclAllGather over xGMI); density_hip_multi_layout() then
 * gives every rank its row and the super-container's length — a rank can write its blob at its offset of a shared file, or send it there (parallel.py:
 * concat_multi_to_rank0).  A reader decodes blob by blob (density_hip_decode / _decode_device per row).  There is nothing in the reference to match (its
 * stream is one chain, codec/codec.rs:72-80).  Pure host arithmetic.
 */
#define DENSITY_HIP_MULTI_MAGIC 0x4D434844u /* "DHCM" little-endian */
typedef struct density_hip_multi_header {
    uint32_t magic;
    uint8_t version, algo;
    uint16_t flags;
    uint32_t n_ranks, chunk_size;
    uint64_t total_len, container_len;
} density_hip_multi_header_t;
typedef struct density_hip_multi_row { uint64_t offset, length, input_bytes; } density_hip_multi_row_t;
/* rows[r] for every rank and the header (algo / chunk_size as given) from the gathered {length, input_bytes}; returns DENSITY_HIP_OK or _ERR_ARGUMENT */
int density_hip_multi_layout(const uint64_t* lengths, const uint64_t* input_bytes, uint32_t n_ranks, int algo, size_t chunk_size,
                             density_hip_multi_header_t* header_out, density_hip_multi_row_t* rows_out);
/* Validates a super-container's front matter (magic, version, rows inside the container, in order, not overlapping, input bytes adding up) and copies
 * row `rank` out; container_size = bytes available.  DENSITY_HIP_OK / _ERR_FORMAT / _ERR_ARGUMENT. */
int density_hip_multi_row(const void* front, size_t front_size, size_t container_size, uint32_t rank, density_hip_multi_header_t* header_out, density_hip_multi_row_t* row_out);

/* Releases what the library holds on every device it has used (staging buffers, workspaces, streams, events); the next call sets them up
 * again.  Not needed for correctness — a process may simply exit — and must not run beside other calls into the library. */
void density_hip_shutdown(void);

/* Thread-local description of the last failure in this thread ("" if none). */
const char* density_hip_last_error(void);
/* "density_hip <version> (gfx950; reference: density-rs 0.16.6; kernels <id>)": <id> is a hash of the kernel sources the library was built from. */
const char* density_hip_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DENSITY_HIP_H */
