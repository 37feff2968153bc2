// kernels.hpp — host-side launchers of the gfx950 kernels (definitions in the .hip files of this directory).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/density_hip.h"

namespace density {

// Tuning and diagnostic switches come from the environment only in a DENSITY_HIP_DEBUG build (python -m density_amd.build --debug ->
// libdensity_hip_debug.so; tools/build_variant.sh): the shipped library has no environment interface.
inline const char* debug_env(const char* name) {
#ifdef DENSITY_HIP_DEBUG
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// ---- chameleon.hip ----
extern bool g_force_simple;   // density_hip_set_kernel_variant(1)
// One wavefront per chunk.  Chunk c reads in[c*chunk_bytes ...) and writes its reference stream at
// out + c*out_stride; sizes[c] receives the stream length.
// d_index (nullable): one byte per 256-byte input block, numbered over the whole input (see include/density_hip.h).
hipError_t launch_chameleon_encode(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks,
                                   uint8_t* d_out, uint64_t out_stride, uint64_t* d_sizes, uint8_t* d_index, uint32_t* d_zmap, uint32_t* d_err, hipStream_t stream);
// d_zmap (nullable -> one-wavefront kernels): kZmapWordsPerChunk words per chunk of scratch for the pipelined kernels' zero-entry maps.
// Chunk c reads the stream at in + offsets[c] (sizes[c] bytes) and writes out + c*out_stride.  With `exact`, a chunk
// that does not produce exactly min(out_stride, out_total - c*out_stride) bytes raises *d_err.
hipError_t launch_chameleon_decode(const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes,
                                   uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride, uint64_t out_total,
                                   bool exact, const uint8_t* d_index, uint32_t* d_zmap, uint64_t* d_produced, uint32_t* d_err, hipStream_t stream);
// the pipelined decoder keeps its zero-entry map in global memory: kZmapWordsPerChunk u32 per chunk, for at most
// kMaxPipelinedChunks chunks (more chunks than that, i.e. tiny chunks, run on the one-wavefront kernel)
constexpr uint32_t kZmapWordsPerChunk = 2048, kMaxPipelinedChunks = 16384;

// ---- rotor.hip (Chameleon wave-rotation kernels: the default encode / index-fed decode path) ----
// pages of a paged container: 64 KiB — a round of 16 blocks is at most 4224 bytes, so a page's unused tail is below 7 % and 2 % on average
constexpr uint32_t kPageShift = 16, kPageBytes = 1u << kPageShift;
// pages one chunk can need (a page is left when the next round's records do not fit: at most a round's worth unused per page), and the words of
// its directory
__host__ __device__ inline uint32_t pages_per_chunk(uint64_t worst_stream_bytes) { return (uint32_t)(worst_stream_bytes / (kPageBytes - 4352u)) + 2u; }
__host__ __device__ inline uint32_t page_dir_words(uint32_t pages) { return 4u * (pages + 1u); }
// what the paged DECODER takes (rotor.hip: kRotMaxBlocks blocks of index, kDecMaxPages directory entries in LDS): the encoder offers no more
constexpr uint64_t kPagedMaxChunk = 4ull << 20;
constexpr uint32_t kPagedMaxPages = 96;
bool rotor_encode_eligible(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks);
// the rotation encoder writing a PAGED container's pages (d_pages: page 0) and directory; d_page_counter zeroed by the caller
hipError_t launch_rotor_encode_paged(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_pages, uint32_t page_limit,
                                     uint32_t* d_page_counter, uint32_t* d_dir, uint32_t dir_words, uint64_t* d_sizes, uint8_t* d_index, uint32_t* d_err, hipStream_t stream);
hipError_t launch_rotor_encode(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                               uint64_t* d_sizes, uint8_t* d_index, uint32_t* d_err, hipStream_t stream);
// Whole-stream-exact encoding in segments (api.hip::run_stream_encode_segmented): a chunk may start from a given dictionary image
// (table + zero-entry map, kSegImageBytes) and FSM state instead of a fresh one, and reports where it ended.
constexpr uint64_t kSegImageBytes = 128ull * 1024 + 8ull * 1024;
struct SegArgs {
    const uint8_t* init_images = nullptr;   // per chunk kSegImageBytes, or nullptr: fresh tables
    const uint32_t* init_guard = nullptr;   // per chunk: packed FSM state (rotor.hip::pack_guard); bit 31: start in speculation (fast) mode
    uint8_t* final_images = nullptr;        // per chunk: the dictionary image after the chunk
    uint32_t* final_guard = nullptr;        // per chunk: the FSM state after the chunk's last whole block
    uint32_t* raw_blocks = nullptr;         // per chunk: number of raw-copy blocks (pre-zeroed by the caller)
    // paged container (round 5, include/density_hip.h DENSITY_HIP_FLAG_PAGED): the streams live in pages of kPageBytes taken from one counter
    uint32_t* page_counter = nullptr;       // encoder: the next free page of the output
    uint32_t* page_dir = nullptr;           // per chunk page_dir_words words: {n_pages, 0, 0, 0}, then per page {page, first block, bytes used, 0}
    uint32_t page_dir_words = 0;
    uint32_t page_limit = 0;                // pages the output has room for (encoder) / the container holds (decoder)
    uint32_t lastwriters_only = 0;          // decoder: a pass that is run for its final dictionary alone — MAP quads are not looked up (their
                                            // slots are mostly empty in a dictionary that starts empty, and every empty read is a zero-entry question)
};
hipError_t launch_rotor_encode_seg(const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                                   uint64_t* d_sizes, uint32_t* d_err, SegArgs seg, hipStream_t stream);
// per chunk (whole rounds only): the dictionary image a fresh table has after every block of the chunk was coded ("last writers")
hipError_t launch_rotor_lastwriters(const uint8_t* d_in, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_images, uint32_t* d_err, hipStream_t stream);
// start images of chunks first+1 .. first+count: base image, then the last-writer images of chunks first+1 .. laid over it one after the other
hipError_t launch_merge_images(const uint8_t* d_base, const uint8_t* d_lastwriters, uint8_t* d_start, uint32_t count, hipStream_t stream);
// offsets[first + i] = *d_carry + sizes[first] + .. + sizes[first + i - 1]; *d_carry moves to the end of the last one
hipError_t launch_scan_offsets(const uint64_t* d_sizes, uint32_t first, uint32_t count, uint64_t* d_carry, uint64_t* d_offsets, hipStream_t stream);
// byte-granular gather of chunk streams (d_src + i * src_stride, sizes[i]) to d_dst + offsets[i]
hipError_t launch_compact_bytes(const uint8_t* d_src, uint64_t src_stride, const uint64_t* d_sizes, const uint64_t* d_offsets, uint32_t n_chunks,
                                uint8_t* d_dst, hipStream_t stream);
// the index-fed decoder on segments of one stream: start dictionaries in, final dictionaries out (SegArgs: init_images / final_images)
hipError_t launch_rotor_decode_seg(const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out,
                                   uint64_t out_stride, uint64_t out_total, const uint8_t* d_index, uint32_t* d_zmap, uint64_t* d_produced, uint32_t* d_err,
                                   SegArgs seg, hipStream_t stream);
bool rotor_decode_eligible(const uint8_t* d_out, uint32_t n_chunks, uint64_t out_stride, uint64_t out_total, const uint8_t* d_index, const uint32_t* d_zmap);
hipError_t launch_rotor_decode(const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out,
                               uint64_t out_stride, uint64_t out_total, bool exact, const uint8_t* d_index, uint32_t* d_zmap,
                               uint64_t* d_produced, uint32_t* d_err, hipStream_t stream);
// LDS assumptions of the rotation kernels (ordered exchange lane order, token hand-off behind the exchanges, lane-reversed rollback)
hipError_t launch_rotor_selftest(uint32_t* d_fail, hipStream_t stream);
constexpr bool kRotorSplitDefault = false;   // which rotation encoder ships: the split one (8 chain + 8 emit waves, rotor.hip) or the 8-wave one
extern bool g_rotor_split;     // density_hip_set_kernel_variant(2048): the OTHER rotation encoder than kRotorSplitDefault (A/B runs, cross-checks)
extern bool g_force_pipeline;  // density_hip_set_kernel_variant(4): the 16-wave role pipelines of chameleon.hip instead
extern bool g_exchange_unsafe, g_rotor_unsafe;   // start-up self-test verdicts (api.hip::acquire_ctx)

// ---- serial_codec.hip (Cheetah: one wave per chunk stream; Lion, and Cheetah as a cross-check: one lane per stream; tables in global memory) ----
extern bool g_force_lane_codec;   // density_hip_set_kernel_variant(16)
extern bool g_lion_one_wave;      // density_hip_set_kernel_variant(32768): Lion's decode on one wave per stream instead of two
uint64_t serial_table_bytes(int algo);
hipError_t launch_serial_encode(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out,
                                uint64_t out_stride, uint64_t* d_sizes, uint8_t* d_tables, uint32_t n_slots, hipStream_t stream);
hipError_t launch_serial_decode(int algo, const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks,
                                uint8_t* d_out, uint64_t out_stride, uint64_t out_total, bool exact, uint64_t* d_produced, uint32_t* d_err,
                                uint8_t* d_tables, uint32_t n_slots, hipStream_t stream);

// The one-wave kernels (Cheetah, Lion) in the service of exchange_stages.hip: the chunks d_only marks, whole (the wave clears its tables itself);
// just the head of every chunk (head_bytes and more, until the blow-up protection is quiet; short or restless chunks are finished), tables (one slot per chunk: n_chunks x serial_table_bytes) and d_head_state (8 words per chunk) left
// behind; just the ragged end of the chunks d_tail_state marks (8 words per chunk), from the tables the passes wrote back
hipError_t launch_wave_encode_only(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                                   uint64_t* d_sizes, uint8_t* d_tables, uint32_t n_slots, const uint32_t* d_only, hipStream_t stream);
hipError_t launch_wave_encode_heads(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                                    uint64_t* d_sizes, uint8_t* d_tables, uint32_t* d_head_state, uint32_t head_bytes, uint32_t head_calm, hipStream_t stream);
hipError_t launch_wave_encode_tails(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                                    uint64_t* d_sizes, uint8_t* d_tables, const uint32_t* d_tail_state, hipStream_t stream);

// ---- exchange_stages.hip (Cheetah / Lion container encode, the default: three / seven passes of ordered LDS exchanges per chunk, then a size scan
// and the records) ----
extern bool g_force_wave_codec;   // density_hip_set_kernel_variant(32): the one-wave-per-stream encoders instead
extern bool g_stage_audit;        // density_hip_set_kernel_variant(64): count chunks kept / handed back (synchronises)
extern uint64_t g_stage_stats[2];
bool stage_encode_eligible(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks);
uint64_t stage_scratch_bytes(int algo, uint64_t total, uint32_t n_chunks);
hipError_t launch_stage_encode(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                               uint64_t* d_sizes, uint8_t* d_tables, uint32_t n_slots, uint8_t* d_scratch, uint32_t* d_err, hipStream_t stream);

// ---- decode_passes.hip (Cheetah container decode, the default: records parsed per chunk, dictionary and prediction tables as ordered LDS
// exchange passes over the whole chunk, the chain of contexts alone on one wave per chunk) ----
extern bool g_force_serial_decode;   // density_hip_set_kernel_variant(128): the one-wave-per-stream decoder instead
extern bool g_serial_parse;          // density_hip_set_kernel_variant(1024): Cheetah's decode passes find the records by the one-wave walk alone
extern int g_walk_blocks;            // density_hip_set_kernel_variant(8192 / 16384): the contexts walked by ONE wave, 64 / 128 quads at a time, instead of by a team of four
extern bool g_chain_walk;            // density_hip_set_kernel_variant(4096): Cheetah's contexts walked run by run on one wave (round 5)
bool decode_pass_eligible(int algo, const uint8_t* d_out, uint32_t n_chunks, uint64_t out_stride, uint64_t out_total);
uint64_t decode_pass_scratch_bytes(uint64_t out_stride, uint32_t n_chunks);
hipError_t launch_decode_passes(int algo, const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out,
                                uint64_t out_stride, uint64_t out_total, bool exact, uint64_t* d_produced, uint32_t* d_err, uint8_t* d_scratch,
                                hipStream_t stream);

// ---- stream_parse.hip: record boundaries of one calm Chameleon stream, in parallel ----
// d_info (16 words): 0 status (1 = a calm head was found), 1 first block behind the sequentially walked head, 2-3 its stream offset, 4 whole
// blocks of the stream, 5-6 stream offset where they end (the ragged end, if any, starts there), 7 first block of a pair of incompressible
// records behind the head (0xffffffff: none); in: 8 block / 9-10 stream offset the head walk starts at (0 / 0 for a whole stream; after a
// pair at block j: j / d_pos32[j], everything before being final).  d_index: one byte per whole block (MAP count; raw copies flagged),
// d_pos32[b]: stream offset of block b, d_chunk_offset[k]: stream offset of block k * chunk_blocks.
uint64_t stream_parse_workspace(uint64_t E);
hipError_t launch_stream_parse(const uint8_t* d_in, uint64_t E, uint64_t from_pos, uint8_t* d_ws, uint8_t* d_index, uint64_t index_cap, uint64_t* d_chunk_offset,
                               uint32_t chunk_blocks, uint32_t* d_pos32, uint32_t* d_info, hipStream_t stream);
// offsets / sizes of segments [first, first + count) of a parsed stream from the parse's d_chunk_offset (`end` != ~0: where the range's last segment ends)
hipError_t launch_seg_layout(const uint64_t* d_chunk_offset, uint32_t first, uint32_t count, uint64_t end, uint64_t* d_offsets, uint64_t* d_sizes, uint32_t* d_err, hipStream_t stream);

// ---- container.hip ----
// Exclusive scan of 16-byte-aligned chunk sizes -> payload offsets; writes the container header and the u32 size
// table (encode side).
// slot_stride != 0: a slotted container (DENSITY_HIP_FLAG_SLOTTED): payload i stays in its slot at payload_base + i * slot_stride
hipError_t launch_layout_encode(const uint64_t* d_sizes, uint32_t n_chunks, density_hip_header_t hdr, uint64_t payload_base, uint8_t* d_container,
                                uint64_t capacity, uint64_t* d_offsets, uint32_t* d_err, hipStream_t stream, uint64_t slot_stride = 0);
// the rotation decoder on a PAGED container: d_pages = page 0 (d_offsets: all zero), d_sizes the streams' lengths, the directory beside them
hipError_t launch_rotor_decode_paged(const uint8_t* d_pages, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks, uint8_t* d_out, uint64_t out_stride,
                                     uint64_t out_total, const uint8_t* d_index, const uint32_t* d_dir, uint32_t dir_words, uint32_t n_pages, uint32_t* d_zmap,
                                     uint64_t* d_produced, uint32_t* d_err, hipStream_t stream);
hipError_t launch_layout_encode_paged(const uint64_t* d_sizes, uint32_t n_chunks, density_hip_header_t hdr, uint64_t dir_base, uint64_t dir_end, uint64_t pages_base, uint8_t* d_container,
                                      uint64_t capacity, const uint32_t* d_page_counter, uint32_t* d_err, hipStream_t stream);
hipError_t launch_layout_decode_paged(const uint8_t* d_container, uint32_t n_chunks, uint64_t* d_sizes, uint64_t* d_offsets, hipStream_t stream);
// The same for a slice of the chunks (batched encode): offsets continue from *d_carry, which is left at the slice's end.
hipError_t launch_layout_encode_batch(const uint64_t* d_sizes, uint32_t first, uint32_t count, bool is_first, bool is_last, density_hip_header_t hdr,
                                      uint64_t payload_base, uint8_t* d_container, uint64_t capacity, uint64_t* d_offsets, uint64_t* d_carry, uint32_t* d_err,
                                      hipStream_t stream);
// Decode side: reads the u32 size table of a container, produces u64 sizes + offsets, validates against container_size.
hipError_t launch_layout_decode(const uint8_t* d_container, uint64_t container_size, uint32_t n_chunks, uint64_t payload_base,
                                uint64_t* d_sizes, uint64_t* d_offsets, uint32_t* d_err, hipStream_t stream, uint64_t slot_stride = 0);
// Gathers chunk streams from their worst-case slots into the packed container.
hipError_t launch_compact(const uint8_t* d_slots, uint64_t slot_stride, const uint64_t* d_sizes, const uint64_t* d_offsets,
                          uint32_t n_chunks, uint8_t* d_container, const uint32_t* d_err, hipStream_t stream, bool more_follow = false);
// (`more_follow`: the launch gathers one batch of a container and further streams follow its last one — the alignment gap behind that stream
// is then zero-filled like the gaps between the launch's own streams, so that a batched gather writes the same bytes as a single one)
// LDS same-address ordering self-test (ascending lane order within one ds instruction). *d_fail != 0 on violation.
hipError_t launch_selftest(uint32_t* d_fail, hipStream_t stream);

}  // namespace density
