/* The drop-in boundary from plain C: the reference's own symbols (chameleon.rs:70-83 and friends) and the container API of
 * include/density_hip.h, linked against libdensity_hip.so.  Build and run (tests/test_gpu_c_example.py does exactly this):
 *   gcc -O2 -Iinclude examples/c_roundtrip.c -Ldensity_amd -ldensity_hip -o /tmp/c_roundtrip && LD_LIBRARY_PATH=density_amd /tmp/c_roundtrip */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "density_hip.h"

static void fill(uint8_t* p, size_t n) {                 /* compressible, not periodic: words from a small vocabulary */
    static const char* words[] = {"density ", "chameleon ", "cheetah ", "lion ", "the ", "of ", "and ", "hash ", "signature ", "block "};
    uint64_t s = 0x9E3779B97F4A7C15ull;
    size_t i = 0;
    while (i < n) {
        s ^= s >> 12; s ^= s << 25; s ^= s >> 27;
        const char* w = words[(s * 0x2545F4914F6CDD1Dull >> 33) % 10];
        size_t l = strlen(w);
        if (l > n - i) l = n - i;
        memcpy(p + i, w, l);
        i += l;
    }
}

int main(void) {
    const size_t n = 3u * 1000u * 1000u + 7u;
    uint8_t* in = malloc(n);
    uint8_t* back = malloc(n);
    fill(in, n);
    int failures = 0;

    /* 1. the reference's symbols: one stream, bit-exact with the crate */
    struct { const char* name; size_t (*enc)(const uint8_t*, size_t, uint8_t*, size_t); size_t (*dec)(const uint8_t*, size_t, uint8_t*, size_t); size_t (*safe)(size_t); } algos[] = {
        {"chameleon", chameleon_encode, chameleon_decode, chameleon_safe_encode_buffer_size},
        {"cheetah", cheetah_encode, cheetah_decode, cheetah_safe_encode_buffer_size},
        {"lion", lion_encode, lion_decode, lion_safe_encode_buffer_size},
    };
    for (int a = 0; a < 3; ++a) {
        const size_t cap = algos[a].safe(n);
        uint8_t* out = malloc(cap);
        const size_t e = algos[a].enc(in, n, out, cap);
        memset(back, 0, n);
        const size_t d = e ? algos[a].dec(out, e, back, n) : 0;
        const int ok = e && d == n && memcmp(in, back, n) == 0;
        printf("%s_encode / %s_decode: %zu -> %zu -> %zu bytes, ratio %.3f: %s\n", algos[a].name, algos[a].name, n, e, d, e ? (double)n / (double)e : 0.0, ok ? "ok" : "FAILED");
        if (!ok) { printf("  last error: %s\n", density_hip_last_error()); ++failures; }
        /* too small an output is an error value, not a crash (the reference panics there) */
        if (algos[a].enc(in, n, out, 100) != 0) { printf("  a 100-byte output buffer was accepted\n"); ++failures; }
        free(out);
    }

    /* 2. the chunked container (host buffers) */
    for (int algo = 0; algo < 3; ++algo) {
        const size_t chunk = 65536;
        const size_t bound = density_hip_container_bound(algo, n, chunk);
        uint8_t* out = malloc(bound);
        const size_t c = density_hip_encode(algo, in, n, out, bound, chunk);
        memset(back, 0, n);
        const size_t want = c ? density_hip_decoded_size(out, c) : 0;
        const size_t d = c ? density_hip_decode(out, c, back, n) : 0;
        const int ok = c && want == n && d == n && memcmp(in, back, n) == 0;
        printf("density_hip_encode / _decode (algorithm %d, %zu-byte chunks): %zu -> %zu -> %zu bytes: %s\n", algo, chunk, n, c, d, ok ? "ok" : "FAILED");
        if (!ok) { printf("  last error: %s\n", density_hip_last_error()); ++failures; }
        free(out);
    }
    free(in); free(back);
    printf("%s\n", failures ? "FAILED" : "all round trips ok");
    return failures ? 1 : 0;
}
