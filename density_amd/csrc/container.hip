// container.hip — layout (size scan), stitch (compaction) and self-test kernels for gfx950.
//
// The encode kernels write every chunk's stream into a worst-case sized slot; chunk sizes are known only afterwards
// (write_buffer.rs:29-31 keeps a running total; in parallel that becomes an exclusive scan).  layout_* computes the
// payload offsets (16-byte aligned so the gather and the decoder's loads are aligned), compact gathers the streams.
#include "common.hpp"
#include "kernels.hpp"

namespace density {

namespace {

constexpr uint32_t kScanThreads = 1024;
constexpr uint32_t kHeaderBytes = 32;
static_assert(sizeof(density_hip_header_t) == kHeaderBytes, "container header is 32 bytes");

__device__ __forceinline__ uint64_t align16(uint64_t v) { return (v + 15ull) & ~15ull; }

// inclusive scan of one u64 per thread across a 1024-thread block; returns inclusive value, *total = block sum
__device__ __forceinline__ uint64_t block_inclusive_scan(uint64_t v, uint64_t* wave_sums /* [16] LDS */, uint64_t* total) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t lo = bperm(lane >= (uint32_t)d ? lane - d : lane, (uint32_t)v);
        const uint32_t hi = bperm(lane >= (uint32_t)d ? lane - d : lane, (uint32_t)(v >> 32));
        if (lane >= (uint32_t)d) v += ((uint64_t)hi << 32) | lo;
    }
    if (lane == 63) wave_sums[wave] = v;
    __syncthreads();
    uint64_t prefix = 0, sum = 0;
    for (uint32_t i = 0; i < kScanThreads / 64; ++i) {
        const uint64_t s = wave_sums[i];
        if (i < wave) prefix += s;
        sum += s;
    }
    __syncthreads();
    *total = sum;
    return v + prefix;
}

// slot_stride != 0: a SLOTTED container (DENSITY_HIP_FLAG_SLOTTED) — payload i sits in its worst-case slot at base + i * slot_stride, no scan
template <typename SizeT>
__device__ __forceinline__ void layout_common(const SizeT* __restrict__ sizes_in, uint32_t n, uint64_t base,
                                              uint64_t* __restrict__ sizes_out, uint32_t* __restrict__ table_out,
                                              uint64_t* __restrict__ offsets, uint64_t* end_out, uint64_t limit, uint32_t* __restrict__ err,
                                              uint64_t slot_stride = 0) {
    __shared__ uint64_t wave_sums[kScanThreads / 64];
    uint64_t carry = base, last_end = base;
    for (uint32_t t0 = 0; t0 < n; t0 += kScanThreads) {
        const uint32_t i = t0 + threadIdx.x;
        const uint64_t sz = i < n ? (uint64_t)sizes_in[i] : 0ull;
        uint64_t tile_total = 0;
        const uint64_t incl = slot_stride ? 0ull : block_inclusive_scan(align16(sz), wave_sums, &tile_total);
        if (i < n) {
            uint64_t off = slot_stride ? base + (uint64_t)i * slot_stride : carry + incl - align16(sz);
            uint64_t keep = sz;
            if (slot_stride && sz > slot_stride) { off = base; keep = 0; atomicOr(err, 4u); }   // a size table entry larger than its slot (never from this library): not followed
            if (off + sz > limit) {                     // a size table that runs past the container: the codec kernels must not follow it
                off = base; keep = 0;
                atomicOr(err, 4u);
            }
            offsets[i] = off;
            if (sizes_out) sizes_out[i] = keep;
            if (table_out) table_out[i] = (uint32_t)sz;
            if (i == n - 1) *end_out = off + sz;     // single writer
        }
        carry += tile_total;
    }
    (void)last_end;
}

__global__ __launch_bounds__(kScanThreads) void layout_encode_kernel(const uint64_t* __restrict__ sizes, uint32_t n,
                                                                     density_hip_header_t hdr, uint64_t base, uint8_t* __restrict__ container,
                                                                     uint64_t capacity, uint64_t* __restrict__ offsets,
                                                                     uint64_t* __restrict__ end_scratch, uint32_t* __restrict__ err, uint64_t slot_stride) {
    if (threadIdx.x == 0) *end_scratch = base;
    __syncthreads();
    layout_common<uint64_t>(sizes, n, base, nullptr, reinterpret_cast<uint32_t*>(container + kHeaderBytes), offsets, end_scratch, ~0ull, err, slot_stride);
    __syncthreads();
    if (threadIdx.x == 0) {
        hdr.container_len = *end_scratch;
        *reinterpret_cast<density_hip_header_t*>(container) = hdr;
        if (hdr.container_len > capacity) atomicOr(err, 2u);
    }
    // the gaps in front of the payloads — size table to block index, block index to the first stream — are part of the container: zeros
    const uint64_t table_end = kHeaderBytes + 4ull * n, ibase = (table_end + 15) / 16 * 16;
    const uint64_t iend = ibase + ((hdr.flags & DENSITY_HIP_FLAG_BLOCK_INDEX) ? (hdr.total_len + 255) / 256 : 0);
    if (threadIdx.x < ibase - table_end) container[table_end + threadIdx.x] = 0;
    if (threadIdx.x >= 32 && threadIdx.x - 32 < base - iend && base <= capacity) container[iend + threadIdx.x - 32] = 0;
}

// A slice [first, first + count) of the chunks (encode in batches: api.hip): offsets continue from *carry (the end of the previous
// batch's last payload; `base` for the first batch), the size-table entries of the slice are written, *carry moves on.  The last
// batch writes the header.
__global__ __launch_bounds__(kScanThreads) void layout_encode_batch_kernel(const uint64_t* __restrict__ sizes, uint32_t first, uint32_t count,
                                                                           uint32_t is_first, uint32_t is_last, density_hip_header_t hdr, uint64_t base,
                                                                           uint8_t* __restrict__ container, uint64_t capacity, uint64_t* __restrict__ offsets,
                                                                           uint64_t* __restrict__ carry, uint32_t* __restrict__ err) {
    __shared__ uint64_t end_scratch;
    const uint64_t start = is_first ? base : align16(*carry);
    if (threadIdx.x == 0) end_scratch = start;
    __syncthreads();
    layout_common<uint64_t>(sizes + first, count, start, nullptr, reinterpret_cast<uint32_t*>(container + kHeaderBytes) + first, offsets + first, &end_scratch, ~0ull, err);
    __syncthreads();
    if (threadIdx.x == 0) {
        *carry = end_scratch;
        if (is_last) {
            hdr.container_len = end_scratch;
            *reinterpret_cast<density_hip_header_t*>(container) = hdr;
        }
        if (end_scratch > capacity) atomicOr(err, 2u);
    }
    if (is_first) {                                              // (as in layout_encode_kernel: the gaps in front of the payloads are zeros)
        const uint64_t table_end = kHeaderBytes + 4ull * hdr.n_chunks, ibase = (table_end + 15) / 16 * 16;
        const uint64_t iend = ibase + ((hdr.flags & DENSITY_HIP_FLAG_BLOCK_INDEX) ? (hdr.total_len + 255) / 256 : 0);
        if (threadIdx.x < ibase - table_end) container[table_end + threadIdx.x] = 0;
        if (threadIdx.x >= 32 && threadIdx.x - 32 < base - iend && base <= capacity) container[iend + threadIdx.x - 32] = 0;
    }
}

// A PAGED container's front matter: size table, header (its length: the pages the encoder took from the counter), zeroed gaps.  The directory and the
// pages were written by the encode kernel itself.
__global__ __launch_bounds__(kScanThreads) void layout_encode_paged_kernel(const uint64_t* __restrict__ sizes, uint32_t n, density_hip_header_t hdr, uint64_t dir_base,
                                                                           uint64_t dir_end, uint64_t pages_base, uint8_t* __restrict__ container, uint64_t capacity,
                                                                           const uint32_t* __restrict__ page_counter, uint32_t* __restrict__ err) {
    uint32_t* table = reinterpret_cast<uint32_t*>(container + kHeaderBytes);
    for (uint32_t i = threadIdx.x; i < n; i += kScanThreads) table[i] = (uint32_t)sizes[i];
    if (threadIdx.x == 0) {
        hdr.container_len = pages_base + (uint64_t)*page_counter * kPageBytes;
        *reinterpret_cast<density_hip_header_t*>(container) = hdr;
        if (hdr.container_len > capacity) atomicOr(err, 2u);
    }
    const uint64_t table_end = kHeaderBytes + 4ull * n, ibase = (table_end + 15) / 16 * 16;
    const uint64_t iend = ibase + (hdr.total_len + 255) / 256;
    if (threadIdx.x < ibase - table_end) container[table_end + threadIdx.x] = 0;
    if (threadIdx.x >= 32 && threadIdx.x - 32 < dir_base - iend) container[iend + threadIdx.x - 32] = 0;
    for (uint64_t i = dir_end + threadIdx.x; i < pages_base; i += kScanThreads) container[i] = 0;
}
// the directory entries behind a chunk's last page are part of the wire bytes too: zeros, not what the buffer held.  A wave per chunk (in the layout kernel's
// one work-group this loop was 33 of its 39 µs — 4 % of the headline's round trip)
__global__ __launch_bounds__(256) void clear_directory_tails_kernel(uint32_t n, uint64_t dir_base, uint64_t dir_end, uint8_t* __restrict__ container) {
    const uint32_t c = blockIdx.x * 4u + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    if (c >= n) return;
    const uint32_t words = (uint32_t)((dir_end - dir_base) / 4 / n);
    uint32_t* dir = reinterpret_cast<uint32_t*>(container + dir_base) + (uint64_t)c * words;
    const uint32_t used = dir[0];                                                     // pages of chunk c (word 0 of its directory; never rewritten here)
    for (uint32_t w = 4u * (used + 1u) + lane; w < words; w += 64u) dir[w] = 0u;
}

__global__ __launch_bounds__(kScanThreads) void layout_decode_kernel(const uint8_t* __restrict__ container, uint64_t container_size,
                                                                     uint32_t n, uint64_t base, uint64_t* __restrict__ sizes,
                                                                     uint64_t* __restrict__ offsets, uint64_t* __restrict__ end_scratch,
                                                                     uint32_t* __restrict__ err, uint64_t slot_stride) {
    if (threadIdx.x == 0) *end_scratch = base;
    __syncthreads();
    layout_common<uint32_t>(reinterpret_cast<const uint32_t*>(container + kHeaderBytes), n, base, sizes, nullptr, offsets, end_scratch, container_size, err, slot_stride);
    __syncthreads();
    if (threadIdx.x == 0 && *end_scratch > container_size) atomicOr(err, 4u);   // truncated container
}

constexpr uint32_t kCopyThreads = 256;
constexpr uint32_t kCopyTile = kCopyThreads * 16u * 4u;   // 16 KiB per work-group

__global__ __launch_bounds__(kCopyThreads) void compact_kernel(const uint8_t* __restrict__ slots, uint64_t slot_stride,
                                                               const uint64_t* __restrict__ sizes, const uint64_t* __restrict__ offsets,
                                                               uint32_t tiles_per_chunk, uint8_t* __restrict__ container,
                                                               const uint32_t* __restrict__ err, uint32_t more_follow) {
    if (*err) return;                                          // layout overflowed the capacity: do not write
    const uint32_t chunk = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const uint64_t size = sizes[chunk];
    const uint64_t begin = (uint64_t)tile * kCopyTile;
    if (begin >= size) return;
    const uint8_t* s = slots + chunk * slot_stride;           // 16-byte aligned (stride and base are)
    uint8_t* d = container + offsets[chunk];                   // 16-byte aligned by layout
    const uint64_t full = size / 16;                           // whole uint4's
    const uint4* s4 = reinterpret_cast<const uint4*>(s);
    uint4* d4 = reinterpret_cast<uint4*>(d);
    const uint64_t i0 = begin / 16 + threadIdx.x;
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint64_t i = i0 + (uint64_t)j * kCopyThreads; if (i < full) v[j] = s4[i]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const uint64_t i = i0 + (uint64_t)j * kCopyThreads; if (i < full) d4[i] = v[j]; }
    // ragged tail (< 16 bytes) handled by the tile that contains it
    const uint64_t tail_at = full * 16;
    if (tail_at >= begin && tail_at < begin + kCopyTile) {
        const uint32_t r = (uint32_t)(size - tail_at);
        if (threadIdx.x < r) d[tail_at + threadIdx.x] = s[tail_at + threadIdx.x];
        // the gap up to the next stream's 16-byte boundary is part of the container: zeros, not whatever the buffer held
        // (`more_follow`: this launch gathers a batch and another batch's streams come behind its last one)
        else if (threadIdx.x < 16 && r != 0 && (chunk + 1 < gridDim.x / tiles_per_chunk || more_follow)) d[tail_at + threadIdx.x] = 0;
    }
}

// LDS ordering assumptions of chameleon.hip, checked on the device the library is running on.
__global__ __launch_bounds__(64) void selftest_kernel(uint32_t* __restrict__ fail) {
    __shared__ __attribute__((aligned(16))) uint16_t cells[256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) cells[i] = 0xffffu;
    __syncthreads();
    const uint32_t base = lds_addr(cells);
    uint32_t bad = 0, r0, r1, r2, r3;
    // (a) all lanes, one u16 cell: highest lane's write must survive; (b) lane pairs share a cell; (c) neighbours in
    // one dword do not clobber each other; (d) the read between two writes of one instruction stream sees the first.
    asm volatile(
        "ds_write_b16 %4, %8\n\t"
        "ds_read_u16 %0, %4\n\t"
        "ds_write_b16 %5, %8\n\t"
        "ds_read_u16 %1, %5\n\t"
        "ds_write_b16 %6, %8\n\t"
        "ds_read_u16 %2, %6\n\t"
        "ds_write_b16 %7, %8\n\t"
        "ds_read_u16 %3, %7\n\t"
        "ds_write_b16 %7, %9\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(base), "v"(base + 64u + 2u * (lane >> 1)), "v"(base + 192u + 2u * lane), "v"(base + 2u * 200u + 2u * (lane & 7u)),
          "v"(lane), "v"(lane + 100u)
        : "memory");
    if (r0 != 63u) bad |= 1u;
    if (r1 != (lane | 1u)) bad |= 2u;
    if (r2 != lane) bad |= 4u;
    if (r3 != (56u + (lane & 7u))) bad |= 8u;
    __syncthreads();
    if (cells[200 + (lane & 7u)] != 156u + (lane & 7u)) bad |= 16u;
    if (bad) atomicOr(fail, bad);
}

}  // namespace

hipError_t launch_layout_encode(const uint64_t* d_sizes, uint32_t n_chunks, density_hip_header_t hdr, uint64_t payload_base, uint8_t* d_container,
                                uint64_t capacity, uint64_t* d_offsets, uint32_t* d_err, hipStream_t stream, uint64_t slot_stride) {
    // d_offsets has n_chunks + 1 entries; the extra one is scratch for the end offset
    hipLaunchKernelGGL(layout_encode_kernel, dim3(1), dim3(kScanThreads), 0, stream, d_sizes, n_chunks, hdr, payload_base, d_container, capacity,
                       d_offsets, d_offsets + n_chunks, d_err, slot_stride);
    return hipGetLastError();
}

hipError_t launch_layout_encode_paged(const uint64_t* d_sizes, uint32_t n_chunks, density_hip_header_t hdr, uint64_t dir_base, uint64_t dir_end, uint64_t pages_base, uint8_t* d_container,
                                      uint64_t capacity, const uint32_t* d_page_counter, uint32_t* d_err, hipStream_t stream) {
    hipLaunchKernelGGL(layout_encode_paged_kernel, dim3(1), dim3(kScanThreads), 0, stream, d_sizes, n_chunks, hdr, dir_base, dir_end, pages_base, d_container, capacity, d_page_counter, d_err);
    if (n_chunks) hipLaunchKernelGGL(clear_directory_tails_kernel, dim3((n_chunks + 3) / 4), dim3(256), 0, stream, n_chunks, dir_base, dir_end, d_container);
    return hipGetLastError();
}

// PAGED container, decode side: the streams' lengths from the u32 table; every chunk reads from page 0 on (the kernel follows the directory)
__global__ __launch_bounds__(kScanThreads) void layout_decode_paged_kernel(const uint8_t* __restrict__ container, uint32_t n, uint64_t* __restrict__ sizes, uint64_t* __restrict__ offsets) {
    const uint32_t* table = reinterpret_cast<const uint32_t*>(container + kHeaderBytes);
    for (uint32_t i = threadIdx.x; i < n; i += kScanThreads) { sizes[i] = table[i]; offsets[i] = 0; }
}
hipError_t launch_layout_decode_paged(const uint8_t* d_container, uint32_t n_chunks, uint64_t* d_sizes, uint64_t* d_offsets, hipStream_t stream) {
    hipLaunchKernelGGL(layout_decode_paged_kernel, dim3(1), dim3(kScanThreads), 0, stream, d_container, n_chunks, d_sizes, d_offsets);
    return hipGetLastError();
}

hipError_t launch_layout_encode_batch(const uint64_t* d_sizes, uint32_t first, uint32_t count, bool is_first, bool is_last, density_hip_header_t hdr,
                                      uint64_t payload_base, uint8_t* d_container, uint64_t capacity, uint64_t* d_offsets, uint64_t* d_carry, uint32_t* d_err,
                                      hipStream_t stream) {
    hipLaunchKernelGGL(layout_encode_batch_kernel, dim3(1), dim3(kScanThreads), 0, stream, d_sizes, first, count, is_first ? 1u : 0u, is_last ? 1u : 0u, hdr,
                       payload_base, d_container, capacity, d_offsets, d_carry, d_err);
    return hipGetLastError();
}

hipError_t launch_layout_decode(const uint8_t* d_container, uint64_t container_size, uint32_t n_chunks, uint64_t payload_base, uint64_t* d_sizes,
                                uint64_t* d_offsets, uint32_t* d_err, hipStream_t stream, uint64_t slot_stride) {
    hipLaunchKernelGGL(layout_decode_kernel, dim3(1), dim3(kScanThreads), 0, stream, d_container, container_size, n_chunks, payload_base, d_sizes,
                       d_offsets, d_offsets + n_chunks, d_err, slot_stride);
    return hipGetLastError();
}

hipError_t launch_compact(const uint8_t* d_slots, uint64_t slot_stride, const uint64_t* d_sizes, const uint64_t* d_offsets,
                          uint32_t n_chunks, uint8_t* d_container, const uint32_t* d_err, hipStream_t stream, bool more_follow) {
    if (n_chunks == 0) return hipSuccess;
    const uint32_t tiles = (uint32_t)((slot_stride + kCopyTile - 1) / kCopyTile);
    const uint64_t blocks = (uint64_t)tiles * n_chunks;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(compact_kernel, dim3((uint32_t)blocks), dim3(kCopyThreads), 0, stream, d_slots, slot_stride, d_sizes, d_offsets, tiles,
                       d_container, d_err, more_follow ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_selftest(uint32_t* d_fail, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, stream, d_fail);
    return hipGetLastError();
}

}  // namespace density
