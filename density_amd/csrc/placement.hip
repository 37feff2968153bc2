// placement.hip — multi-GPU placement arithmetic of the container path for callers below Python (include/density_hip.h:
// density_hip_shard_range, density_hip_global_layout).  Host code only: no kernel, no HIP call.  The path shards by chunks (SURVEY.md 8e;
// there is nothing to match in the reference, whose stream is one chain: codec/codec.rs:72-80); density_amd/parallel.py holds the same
// arithmetic for torch.distributed callers and tests/test_placement_abi.py holds the two against each other.
#include <cstdint>
#include <cstddef>

#include "../../include/density_hip.h"

namespace {
inline uint64_t align16(uint64_t v) { return (v + 15u) / 16u * 16u; }
}  // namespace

extern "C" {

int density_hip_shard_range(size_t total_len, size_t chunk_size, uint32_t rank, uint32_t world, density_hip_shard_t* out) {
    if (!out || world == 0 || rank >= world || chunk_size < 256 || chunk_size % 256 != 0) return DENSITY_HIP_ERR_ARGUMENT;
    const uint64_t n_chunks = ((uint64_t)total_len + chunk_size - 1) / chunk_size;
    // contiguous chunk ranges, balanced to within one chunk (128-bit products: n_chunks * world may pass 2^64 for absurd inputs only, but cheaply exact)
    const uint64_t c0 = (uint64_t)(((unsigned __int128)n_chunks * rank) / world);
    const uint64_t c1 = (uint64_t)(((unsigned __int128)n_chunks * (rank + 1ull)) / world);
    const unsigned __int128 b0 = (unsigned __int128)c0 * chunk_size, b1 = (unsigned __int128)c1 * chunk_size;
    out->chunk_first = c0;
    out->chunk_end = c1;
    out->byte_first = b0 < total_len ? (uint64_t)b0 : (uint64_t)total_len;
    out->byte_end = b1 < total_len ? (uint64_t)b1 : (uint64_t)total_len;
    return DENSITY_HIP_OK;
}

int density_hip_global_layout(const uint64_t* chunks, const uint64_t* payload_bytes, const uint64_t* input_bytes, uint32_t world, uint32_t rank,
                              uint32_t flags, density_hip_global_layout_t* out) {
    if (!chunks || !payload_bytes || !input_bytes || !out || world == 0 || rank >= world) return DENSITY_HIP_ERR_ARGUMENT;
    // every payload region but the last non-empty one is padded, so that the payload behind it starts 16-byte aligned
    int64_t last = -1;
    for (uint32_t r = 0; r < world; ++r) if (payload_bytes[r] > 0) last = (int64_t)r;
    uint64_t n_chunks = 0, total_len = 0, pay_total = 0;
    *out = density_hip_global_layout_t{};
    for (uint32_t r = 0; r < world; ++r) {
        const uint64_t padded = (int64_t)r == last ? payload_bytes[r] : align16(payload_bytes[r]);
        if (r == rank) { out->chunk_offset = n_chunks; out->payload_offset = pay_total; out->input_offset = total_len; out->payload_bytes_padded = padded; }
        n_chunks += chunks[r];
        total_len += input_bytes[r];
        pay_total += padded;
    }
    out->n_chunks = n_chunks;
    out->total_len = total_len;
    out->index_at = align16(sizeof(density_hip_header_t) + 4u * n_chunks);
    out->index_bytes = (flags & DENSITY_HIP_FLAG_BLOCK_INDEX) ? (total_len + 255u) / 256u : 0u;
    out->payload_at = align16(out->index_at + out->index_bytes);
    out->container_len = out->payload_at + pay_total;                              // (the last region is not padded at its end)
    return DENSITY_HIP_OK;
}

// ---- the multi-rank container "DHCM" (include/density_hip.h): every rank's blob as it stands behind a table of {offset, length, input bytes} ----
int density_hip_multi_layout(const uint64_t* lengths, const uint64_t* input_bytes, uint32_t n_ranks, int algo, size_t chunk_size,
                             density_hip_multi_header_t* header_out, density_hip_multi_row_t* rows_out) {
    if (!lengths || !input_bytes || !header_out || !rows_out || n_ranks == 0 || algo < DENSITY_HIP_CHAMELEON || algo > DENSITY_HIP_LION) return DENSITY_HIP_ERR_ARGUMENT;
    uint64_t at = (sizeof(density_hip_multi_header_t) + (uint64_t)n_ranks * sizeof(density_hip_multi_row_t) + 255u) / 256u * 256u, total = 0;
    for (uint32_t r = 0; r < n_ranks; ++r) {
        rows_out[r] = density_hip_multi_row_t{at, lengths[r], input_bytes[r]};
        total += input_bytes[r];
        at += lengths[r];
        if (r + 1 < n_ranks) at = (at + 255u) / 256u * 256u;                         // (the last blob is not padded at its end)
    }
    *header_out = density_hip_multi_header_t{DENSITY_HIP_MULTI_MAGIC, 1, (uint8_t)algo, 0, n_ranks, (uint32_t)chunk_size, total, at};
    return DENSITY_HIP_OK;
}

int density_hip_multi_row(const void* front, size_t front_size, size_t container_size, uint32_t rank, density_hip_multi_header_t* header_out, density_hip_multi_row_t* row_out) {
    if (!front || !header_out || !row_out) return DENSITY_HIP_ERR_ARGUMENT;
    if (front_size < sizeof(density_hip_multi_header_t)) return DENSITY_HIP_ERR_FORMAT;
    density_hip_multi_header_t h;
    __builtin_memcpy(&h, front, sizeof(h));
    if (h.magic != DENSITY_HIP_MULTI_MAGIC || h.version != 1 || h.flags != 0 || h.algo > DENSITY_HIP_LION || h.n_ranks == 0 || h.n_ranks > (1u << 20)) return DENSITY_HIP_ERR_FORMAT;
    const uint64_t rows_end = sizeof(h) + (uint64_t)h.n_ranks * sizeof(density_hip_multi_row_t);
    if (front_size < rows_end || h.container_len > container_size || h.container_len < rows_end) return DENSITY_HIP_ERR_FORMAT;
    if (rank >= h.n_ranks) return DENSITY_HIP_ERR_ARGUMENT;
    uint64_t at = rows_end, total = 0;
    for (uint32_t r = 0; r < h.n_ranks; ++r) {                                       // rows in order, inside the container, not overlapping; input bytes add up
        density_hip_multi_row_t row;
        __builtin_memcpy(&row, static_cast<const uint8_t*>(front) + sizeof(h) + (uint64_t)r * sizeof(row), sizeof(row));
        if (row.offset < at || row.offset % 256u != 0 || row.offset > h.container_len || row.length > h.container_len - row.offset) return DENSITY_HIP_ERR_FORMAT;
        if (row.input_bytes > h.total_len - total) return DENSITY_HIP_ERR_FORMAT;
        at = row.offset + row.length;
        total += row.input_bytes;
        if (r == rank) *row_out = row;
    }
    if (total != h.total_len) return DENSITY_HIP_ERR_FORMAT;
    *header_out = h;
    return DENSITY_HIP_OK;
}

}  // extern "C"
