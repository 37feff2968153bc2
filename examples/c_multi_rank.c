/* Config 5's wire form from plain C — the host arithmetic only (no device, no collective: the gathered {container length, input bytes} of four ranks are
 * made up here): density_hip_shard_range() gives every rank its chunks, density_hip_multi_layout() the super-container "DHCM" over the ranks' blobs as they
 * stand, density_hip_multi_row() is what a reader calls on the front matter.  On a real node the lengths come out of ncclAllGather (16 bytes per rank) and
 * every rank writes its blob at rows[rank].offset.   gcc -Iinclude examples/c_multi_rank.c -Ldensity_amd -ldensity_hip */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "density_hip.h"

int main(void) {
    enum { WORLD = 4 };
    const size_t total = ((size_t)1 << 30) + 12345, chunk = 4u << 20;
    uint64_t lengths[WORLD], inputs[WORLD], sum = 0;
    for (uint32_t r = 0; r < WORLD; ++r) {
        density_hip_shard_t sh;
        if (density_hip_shard_range(total, chunk, r, WORLD, &sh) != DENSITY_HIP_OK) { printf("shard_range failed\n"); return 1; }
        const uint64_t first = sh.chunk_first, count = sh.chunk_end - sh.chunk_first, off = sh.byte_first, bytes = sh.byte_end - sh.byte_first;
        inputs[r] = bytes;
        lengths[r] = bytes * 5 / 8 + 4096 + 64 * r;                       /* what the rank's encoder would report: hdr.container_len */
        sum += bytes;
        printf("rank %u: chunks [%llu, %llu), input bytes %llu at %llu\n", r, (unsigned long long)first, (unsigned long long)(first + count),
               (unsigned long long)bytes, (unsigned long long)off);
    }
    if (sum != total) { printf("the shards do not cover the input\n"); return 1; }
    density_hip_multi_header_t h;
    density_hip_multi_row_t rows[WORLD];
    if (density_hip_multi_layout(lengths, inputs, WORLD, DENSITY_HIP_CHAMELEON, chunk, &h, rows) != DENSITY_HIP_OK) { printf("multi_layout failed\n"); return 1; }
    /* the front matter as rank 0 writes it at offset 0 */
    unsigned char front[sizeof h + sizeof rows];
    memcpy(front, &h, sizeof h);
    memcpy(front + sizeof h, rows, sizeof rows);
    uint64_t out_at = 0;
    for (uint32_t r = 0; r < WORLD; ++r) {
        density_hip_multi_header_t gh;
        density_hip_multi_row_t row;
        if (density_hip_multi_row(front, sizeof front, (size_t)h.container_len, r, &gh, &row) != DENSITY_HIP_OK) { printf("multi_row failed\n"); return 1; }
        if (row.offset % 256 || row.length != lengths[r] || row.input_bytes != inputs[r] || gh.total_len != total) { printf("row %u is not what was laid out\n", r); return 1; }
        printf("blob %u: [%llu, %llu) of the super-container decodes to output bytes [%llu, %llu)\n", r, (unsigned long long)row.offset,
               (unsigned long long)(row.offset + row.length), (unsigned long long)out_at, (unsigned long long)(out_at + row.input_bytes));
        out_at += row.input_bytes;
    }
    front[sizeof h] ^= 1;                                                 /* blob 0 no longer 256-byte aligned: a reader must refuse the front matter */
    density_hip_multi_header_t gh;
    density_hip_multi_row_t row;
    if (density_hip_multi_row(front, sizeof front, (size_t)h.container_len, 0, &gh, &row) != DENSITY_HIP_ERR_FORMAT) { printf("a crafted row was accepted\n"); return 1; }
    printf("super-container: %llu bytes for %llu input bytes in %u blobs\nmulti-rank layout ok\n", (unsigned long long)h.container_len, (unsigned long long)h.total_len, h.n_ranks);
    return 0;
}
