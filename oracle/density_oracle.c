/*
 * density_oracle.c — CPU restatement of density-rs 0.16.6 (g1mv/density) block codec.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP path in density_amd/csrc.
 * It is imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
 * library (libdensity_hip.so) never links, loads or calls it.
 *
 * Parity pin: the three golden vectors of the reference's own unit tests (src/lib.rs:28,50,72) are
 * reproduced byte-for-byte by tests/test_oracle_golden.py.  The Rust reference itself cannot be built in
 * this image (no rustc/cargo), so there is no oracle/_ref binary; see DESIGN.md "Oracle".
 *
 * Every function cites the reference file:line (relative to /root/reference/src) whose behaviour it restates.
 * Plain C11, no dependencies.  Build: see oracle/Makefile.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))
#define FORCE_INLINE static inline __attribute__((always_inline))

enum { ALGO_CHAMELEON = 0, ALGO_CHEETAH = 1, ALGO_LION = 2 };

/* hash: algorithms/chameleon/chameleon.rs:14-15,89 (same constant cheetah.rs:14-15, lion.rs:14-15) */
#define HASH_MULTIPLIER 0x9D6EF916u
FORCE_INLINE uint32_t hash16(uint32_t quad) { return (uint32_t)(quad * HASH_MULTIPLIER) >> 16; }

/* per-algorithm geometry: chameleon.rs:138-146, cheetah.rs:188-196, lion.rs:317-325 */
FORCE_INLINE size_t sig_bytes(int algo)   { return algo == ALGO_LION ? 6 : 8; }
FORCE_INLINE size_t flag_bits(int algo)   { return algo == ALGO_CHAMELEON ? 1 : (algo == ALGO_CHEETAH ? 2 : 3); }
FORCE_INLINE size_t block_bytes(int algo) { return 4 * (sig_bytes(algo) * 8) / flag_bits(algo); } /* 256 / 128 / 64 */

/* codec/codec.rs:18-21 */
FORCE_INLINE size_t safe_size(int algo, size_t n) {
    size_t b = block_bytes(algo), s = sig_bytes(algo);
    return n + (n / b) * s + ((n % b) ? s : 0);
}

/* ---- little-endian unaligned access (io/read_buffer.rs:32-44, io/write_buffer.rs:17-31) ---- */
FORCE_INLINE uint16_t ld16(const uint8_t* p) { uint16_t v; memcpy(&v, p, 2); return v; }
FORCE_INLINE uint32_t ld32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
FORCE_INLINE uint64_t ld64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }
FORCE_INLINE void st16(uint8_t* p, uint16_t v) { memcpy(p, &v, 2); }
FORCE_INLINE void st32(uint8_t* p, uint32_t v) { memcpy(p, &v, 4); }
#if __BYTE_ORDER__ != __ORDER_LITTLE_ENDIAN__
#error "oracle assumes a little-endian host (the stream format itself is little-endian)"
#endif

/* ---- blow-up protection FSM: codec/protection_state.rs:1-47 ---- */
typedef struct { uint8_t penalty, penalty_start; uint8_t prev_incompressible; uint64_t counter; } guard_t;
FORCE_INLINE void guard_init(guard_t* g) { g->penalty = 0; g->penalty_start = 1; g->prev_incompressible = 0; g->counter = 0; }
/* protection_state.rs:19-27 */
FORCE_INLINE int guard_block_is_copy(guard_t* g) {
    if ((g->counter & 0xf) == 0 && g->penalty_start > 1) g->penalty_start >>= 1;
    g->counter++;
    return g->penalty > 0;
}
/* protection_state.rs:30-35 */
FORCE_INLINE void guard_decay(guard_t* g) { if (--g->penalty == 0) g->penalty_start++; }
/* protection_state.rs:38-47 */
FORCE_INLINE void guard_update(guard_t* g, int incompressible) {
    if (incompressible && g->prev_incompressible) g->penalty = g->penalty_start;
    g->prev_incompressible = (uint8_t)(incompressible != 0);
}

/* ---- table state: chameleon.rs:30-43, cheetah.rs:25-55, lion.rs:29-72 (all zero-initialised) ---- */
typedef struct { uint32_t a, b; } pair_t;
typedef struct { uint32_t n[5]; } pred5_t;
typedef struct {
    uint32_t* dict;    /* chameleon: 64Ki x u32 */
    pair_t*   dict2;   /* cheetah/lion: 64Ki x {a,b} */
    uint32_t* pred;    /* cheetah: 64Ki x next */
    pred5_t*  pred5;   /* lion: 64Ki x {a..e} */
    uint32_t  last_hash;
} state_t;

/* (the all-cores runs at the end of this file keep a thread's tables from chunk to chunk — zeroed, not re-allocated: a few hundred threads mapping and
 * unmapping 0.75-1.75 MiB per chunk spend their time in the address-space lock, not in the codec.  Single calls allocate and free, like the reference's
 * Vec per call.) */
static __thread int tls_keep_tables = 0;
static __thread state_t tls_tables[3];
static int state_alloc(state_t* st, int algo) {
    memset(st, 0, sizeof *st);
    if (tls_keep_tables) {
        state_t* k = &tls_tables[algo];
        if (algo == ALGO_CHAMELEON) { if (!k->dict) k->dict = malloc(65536 * sizeof(uint32_t)); if (!k->dict) return 0; memset(k->dict, 0, 65536 * sizeof(uint32_t)); }
        else {
            if (!k->dict2) k->dict2 = malloc(65536 * sizeof(pair_t));
            if (algo == ALGO_CHEETAH) { if (!k->pred) k->pred = malloc(65536 * sizeof(uint32_t)); } else if (!k->pred5) k->pred5 = malloc(65536 * sizeof(pred5_t));
            if (!k->dict2 || !(algo == ALGO_CHEETAH ? (void*)k->pred : (void*)k->pred5)) return 0;
            memset(k->dict2, 0, 65536 * sizeof(pair_t));
            if (algo == ALGO_CHEETAH) memset(k->pred, 0, 65536 * sizeof(uint32_t)); else memset(k->pred5, 0, 65536 * sizeof(pred5_t));
        }
        *st = *k; st->last_hash = 0;
        return 1;
    }
    if (algo == ALGO_CHAMELEON) { st->dict = calloc(65536, sizeof(uint32_t)); return st->dict != NULL; }
    st->dict2 = calloc(65536, sizeof(pair_t));
    if (algo == ALGO_CHEETAH) st->pred = calloc(65536, sizeof(uint32_t));
    else st->pred5 = calloc(65536, sizeof(pred5_t));
    return st->dict2 && (st->pred || st->pred5);
}
static void state_free(state_t* st) { if (tls_keep_tables) return; free(st->dict); free(st->dict2); free(st->pred); free(st->pred5); }

/* optional statistics for tests (not part of the reference) */
typedef struct { uint64_t copy_blocks, coded_blocks, flags[8]; } oracle_stats_t;

/* ---- encode_quad: emits one flag (returned) and 0/2/4 item bytes at *op ---- */

/* chameleon.rs:88-100 */
FORCE_INLINE uint32_t enc_quad_chameleon(state_t* st, uint32_t q, uint8_t** op) {
    uint32_t h = hash16(q);
    if (st->dict[h] != q) { st->dict[h] = q; st32(*op, q); *op += 4; return 0; }
    st16(*op, (uint16_t)h); *op += 2; return 1;
}

/* cheetah.rs:123-149 */
FORCE_INLINE uint32_t enc_quad_cheetah(state_t* st, uint32_t q, uint8_t** op) {
    uint32_t h = hash16(q), flag;
    uint32_t* guess = &st->pred[st->last_hash];
    if (*guess == q) { flag = 3; }
    else {
        pair_t* e = &st->dict2[h];
        if (e->a == q) { flag = 1; st16(*op, (uint16_t)h); *op += 2; }
        else {
            if (e->b == q) { flag = 2; st16(*op, (uint16_t)h); *op += 2; }
            else           { flag = 0; st32(*op, q); *op += 4; }
            e->b = e->a; e->a = q;
        }
        *guess = q;
    }
    st->last_hash = h;
    return flag;
}

/* lion.rs:50-57 */
FORCE_INLINE void pred5_push_front(pred5_t* p, uint32_t q) { p->n[4] = p->n[3]; p->n[3] = p->n[2]; p->n[2] = p->n[1]; p->n[1] = p->n[0]; p->n[0] = q; }
/* move entry k (1..4) to the front, shifting 0..k-1 down: lion.rs:240-262 (encode), :135-186 (decode) */
FORCE_INLINE void pred5_promote(pred5_t* p, int k, uint32_t q) { for (int i = k; i > 0; --i) p->n[i] = p->n[i - 1]; p->n[0] = q; }

/* lion.rs:211-270 */
FORCE_INLINE uint32_t enc_quad_lion(state_t* st, uint32_t q, uint8_t** op) {
    uint32_t h = hash16(q), flag;
    pred5_t* p = &st->pred5[st->last_hash];
    if      (p->n[0] == q) { flag = 1; }
    else if (p->n[1] == q) { flag = 2; pred5_promote(p, 1, q); }
    else if (p->n[2] == q) { flag = 3; pred5_promote(p, 2, q); }
    else if (p->n[3] == q) { flag = 4; pred5_promote(p, 3, q); }
    else if (p->n[4] == q) { flag = 5; pred5_push_front(p, q); }      /* E: full shift, lion.rs:240-243 */
    else {
        pair_t* e = &st->dict2[h];
        if (e->a == q) { flag = 6; st16(*op, (uint16_t)h); *op += 2; }
        else {
            if (e->b == q) { flag = 7; st16(*op, (uint16_t)h); *op += 2; }
            else           { flag = 0; st32(*op, q); *op += 4; }
            e->b = e->a; e->a = q;
        }
        pred5_push_front(p, q);
    }
    st->last_hash = h;
    return flag;
}

FORCE_INLINE uint32_t enc_quad(int algo, state_t* st, uint32_t q, uint8_t** op) {
    return algo == ALGO_CHAMELEON ? enc_quad_chameleon(st, q, op) : algo == ALGO_CHEETAH ? enc_quad_cheetah(st, q, op) : enc_quad_lion(st, q, op);
}

/* ---- Codec::encode / encode_block: codec/codec.rs:34-80; signature ink codec.rs:24-26 / lion.rs:334-337 ---- */
FORCE_INLINE size_t encode_stream(int algo, const uint8_t* in, size_t n, uint8_t* out, size_t cap, oracle_stats_t* stats) {
    const size_t B = block_bytes(algo), S = sig_bytes(algo), F = flag_bits(algo);
    uint8_t* scratch = NULL;
    uint8_t* dst = out;
    size_t need = safe_size(algo, n);
    /* the reference panics when `output` is too small (write_buffer.rs:19); we encode into scratch and
       report 0 unless the real output fits (src/lib.rs tests use an output smaller than the safe size) */
    if (cap < need + 8) { scratch = malloc(need + 8); if (!scratch) return 0; dst = scratch; }
    state_t st; if (!state_alloc(&st, algo)) { free(scratch); return 0; }
    guard_t g; guard_init(&g);
    uint8_t* op = dst;
    for (size_t pos = 0; pos < n; pos += B) {
        size_t len = n - pos < B ? n - pos : B;
        const uint8_t* blk = in + pos;
        if (guard_block_is_copy(&g)) {                           /* codec.rs:35-37 */
            memcpy(op, blk, len); op += len; guard_decay(&g);
            if (stats) stats->copy_blocks++;
            continue;
        }
        uint8_t* rec = op; op += S;                               /* codec.rs:38-41 */
        uint64_t sig = 0; unsigned shift = 0;
        size_t nq = len / 4;
        for (size_t k = 0; k < nq; ++k) {                         /* codec.rs:42-57 (u128 unroll == quad order) */
            uint32_t flag = enc_quad(algo, &st, ld32(blk + 4 * k), &op);
            sig |= (uint64_t)flag << shift; shift += (unsigned)F; /* io/write_signature.rs:14-17 */
            if (stats) stats->flags[flag]++;
        }
        size_t tail = len & 3;                                    /* codec.rs:58-61: raw, no flag bits */
        memcpy(op, blk + 4 * nq, tail); op += tail;
        for (size_t i = 0; i < S; ++i) rec[i] = (uint8_t)(sig >> (8 * i));
        guard_update(&g, (size_t)(op - rec) >= B);                /* codec.rs:68 */
        if (stats) stats->coded_blocks++;
    }
    size_t produced = (size_t)(op - dst);
    state_free(&st);
    if (scratch) {
        /* Chameleon/Cheetah ink 8 bytes at the signature slot (write_buffer.rs:24-26), always inside `produced`
           because a coded record is never shorter than its signature */
        if (produced <= cap) memcpy(out, scratch, produced); else produced = 0;
        free(scratch);
    }
    return produced;
}

/* ---- decoders ---- */
typedef struct { const uint8_t* p; const uint8_t* end; } rd_t;
FORCE_INLINE size_t rd_left(const rd_t* r) { return (size_t)(r->end - r->p); }

/* chameleon.rs:56-68 */
FORCE_INLINE uint32_t dec_chameleon(state_t* st, uint32_t flag, rd_t* r) {
    if (flag == 0) { uint32_t q = ld32(r->p); r->p += 4; st->dict[hash16(q)] = q; return q; }
    uint32_t h = ld16(r->p); r->p += 2; return st->dict[h];
}
/* cheetah.rs:68-103,154-163 */
FORCE_INLINE uint32_t dec_cheetah(state_t* st, uint32_t flag, rd_t* r) {
    uint32_t q, h;
    if (flag == 3) { q = st->pred[st->last_hash]; h = hash16(q); }
    else {
        pair_t* e;
        if (flag == 0) { q = ld32(r->p); r->p += 4; h = hash16(q); e = &st->dict2[h]; e->b = e->a; e->a = q; }
        else {
            h = ld16(r->p); r->p += 2; e = &st->dict2[h];
            if (flag == 1) q = e->a; else { q = e->b; e->b = e->a; e->a = q; }
        }
        st->pred[st->last_hash] = q;
    }
    st->last_hash = h;
    return q;
}
/* lion.rs:85-186,275-290 */
FORCE_INLINE uint32_t dec_lion(state_t* st, uint32_t flag, rd_t* r) {
    uint32_t q, h;
    pred5_t* p = &st->pred5[st->last_hash];
    if (flag >= 1 && flag <= 5) {
        int k = (int)flag - 1; q = p->n[k]; h = hash16(q);
        if (k > 0) pred5_promote(p, k, q);
    } else {
        pair_t* e;
        if (flag == 0) { q = ld32(r->p); r->p += 4; h = hash16(q); e = &st->dict2[h]; e->b = e->a; e->a = q; }
        else {
            h = ld16(r->p); r->p += 2; e = &st->dict2[h];
            if (flag == 6) q = e->a; else { q = e->b; e->b = e->a; e->a = q; }
        }
        pred5_push_front(p, q);
    }
    st->last_hash = h;
    return q;
}
FORCE_INLINE uint32_t dec_item(int algo, state_t* st, uint32_t flag, rd_t* r) {
    return algo == ALGO_CHAMELEON ? dec_chameleon(st, flag, r) : algo == ALGO_CHEETAH ? dec_cheetah(st, flag, r) : dec_lion(st, flag, r);
}
FORCE_INLINE size_t item_bytes(int algo, uint32_t flag) {
    if (flag == 0) return 4;
    if (algo == ALGO_CHAMELEON) return 2;
    if (algo == ALGO_CHEETAH) return flag == 3 ? 0 : 2;
    return flag >= 6 ? 2 : 0;
}

/* codec.rs:28-31 (8-byte signature); lion.rs:340-351 (6 significant bytes; short read near the end) */
FORCE_INLINE int read_sig(int algo, rd_t* r, uint64_t* sig) {
    if (algo == ALGO_LION) {
        if (rd_left(r) < 6) return 0;
        uint64_t v = 0; for (int i = 0; i < 6; ++i) v |= (uint64_t)r->p[i] << (8 * i);
        r->p += 6; *sig = v; return 1;
    }
    if (rd_left(r) < 8) return 0;
    *sig = ld64(r->p); r->p += 8; return 1;
}

/* Codec::decode: codec/codec.rs:82-126.  Returns 0 where the reference would panic (truncated input,
   output too small). */
FORCE_INLINE size_t decode_stream(int algo, const uint8_t* in, size_t n, uint8_t* out, size_t cap) {
    const size_t B = block_bytes(algo), S = sig_bytes(algo), F = flag_bits(algo);
    const uint64_t FM = ((uint64_t)1 << F) - 1;
    const size_t quads_per_block = B / 4;
    state_t st; if (!state_alloc(&st, algo)) return 0;
    guard_t g; guard_init(&g);
    rd_t r = { in, in + n };
    uint8_t* op = out; uint8_t* oend = out + cap;
    int ok = 1;
    /* fast loop, codec.rs:88-100: a whole record is known to be present */
    while (rd_left(&r) >= S + B) {
        if ((size_t)(oend - op) < B) { ok = 0; goto done; }
        if (guard_block_is_copy(&g)) { memcpy(op, r.p, B); r.p += B; op += B; guard_decay(&g); continue; }
        const uint8_t* mark = r.p; uint64_t sig;
        read_sig(algo, &r, &sig);
        for (size_t k = 0; k < quads_per_block; ++k) {          /* chameleon decodes 2 per unit: same order */
            uint32_t q = dec_item(algo, &st, (uint32_t)(sig & FM), &r); sig >>= F;
            st32(op, q); op += 4;
        }
        guard_update(&g, (size_t)(r.p - mark) >= B);
    }
    /* tail loop, codec.rs:102-123 with decode_partial_unit (chameleon.rs:117-135, cheetah.rs:165-185, lion.rs:292-314) */
    while (rd_left(&r) > 0) {
        if (guard_block_is_copy(&g)) {
            size_t take = rd_left(&r) > B ? B : rd_left(&r);
            if ((size_t)(oend - op) < take) { ok = 0; goto done; }
            memcpy(op, r.p, take); op += take; r.p += take;
            if (rd_left(&r) == 0) break;                          /* codec.rs:107-109: last (possibly short) raw block */
            guard_decay(&g);
            continue;
        }
        const uint8_t* mark = r.p; uint64_t sig;
        if (!read_sig(algo, &r, &sig)) { ok = 0; goto done; }
        int stop = 0;
        for (size_t k = 0; k < quads_per_block; ++k) {
            uint32_t flag = (uint32_t)(sig & FM); sig >>= F;
            size_t left = rd_left(&r);
            if (flag == 0 && left < 4) {                          /* implicit PLAIN at end of data */
                if ((size_t)(oend - op) < left) { ok = 0; goto done; }
                memcpy(op, r.p, left); op += left; r.p += left; stop = 1; break;
            }
            if (left < item_bytes(algo, flag)) { ok = 0; goto done; }   /* reference: slice panic */
            if ((size_t)(oend - op) < 4) { ok = 0; goto done; }
            uint32_t q = dec_item(algo, &st, flag, &r);
            st32(op, q); op += 4;
        }
        if (stop) break;
        guard_update(&g, (size_t)(r.p - mark) >= B);
    }
done:
    state_free(&st);
    return ok ? (size_t)(op - out) : 0;
}

/* ---- exported symbols (specialised per algorithm so the compiler folds the geometry) ---- */
#define EXPORT_ALGO(name, ALGO)                                                                                   \
    ORACLE_API size_t oracle_##name##_encode(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {             \
        return encode_stream(ALGO, in, n, out, cap, NULL); }                                                      \
    ORACLE_API size_t oracle_##name##_encode_stats(const uint8_t* in, size_t n, uint8_t* out, size_t cap, oracle_stats_t* s) { \
        memset(s, 0, sizeof *s); return encode_stream(ALGO, in, n, out, cap, s); }                                \
    ORACLE_API size_t oracle_##name##_decode(const uint8_t* in, size_t n, uint8_t* out, size_t cap) {             \
        return decode_stream(ALGO, in, n, out, cap); }                                                            \
    ORACLE_API size_t oracle_##name##_safe_encode_buffer_size(size_t n) { return safe_size(ALGO, n); }

EXPORT_ALGO(chameleon, ALGO_CHAMELEON)
EXPORT_ALGO(cheetah, ALGO_CHEETAH)
EXPORT_ALGO(lion, ALGO_LION)

/* ---- the chunks of a buffer on all host cores (bench.py's all-cores baseline: SURVEY.md 8d "N-thread run over the same chunks") ----
 * Chunk i of `in` (chunk bytes, the last one shorter) is one independent reference stream — what the container holds —, written to / read from
 * out + i * stride; one chunk per task (OpenMP, dynamic schedule); sizes[i] = the stream's length.  Returns the number of chunks that failed. */
ORACLE_API int oracle_encode_chunks_mt(int algo, const uint8_t* in, size_t n, size_t chunk, uint8_t* out, size_t stride, uint64_t* sizes, int threads) {
    const long n_chunks = (long)((n + chunk - 1) / chunk);
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : bad)
    for (long i = 0; i < n_chunks; ++i) {
        tls_keep_tables = 1;
        const size_t len = (size_t)(i + 1) * chunk <= n ? chunk : n - (size_t)i * chunk;
        const uint8_t* p = in + (size_t)i * chunk; uint8_t* o = out + (size_t)i * stride;
        const size_t e = algo == ALGO_CHAMELEON ? encode_stream(ALGO_CHAMELEON, p, len, o, stride, NULL)
                       : algo == ALGO_CHEETAH ? encode_stream(ALGO_CHEETAH, p, len, o, stride, NULL) : encode_stream(ALGO_LION, p, len, o, stride, NULL);
        sizes[i] = e;
        if (e == 0 && len != 0) ++bad;
        tls_keep_tables = 0;
    }
    return bad;
}
ORACLE_API int oracle_decode_chunks_mt(int algo, const uint8_t* in, size_t stride, const uint64_t* sizes, uint8_t* out, size_t n, size_t chunk, int threads) {
    const long n_chunks = (long)((n + chunk - 1) / chunk);
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : bad)
    for (long i = 0; i < n_chunks; ++i) {
        tls_keep_tables = 1;
        const size_t len = (size_t)(i + 1) * chunk <= n ? chunk : n - (size_t)i * chunk;
        const uint8_t* p = in + (size_t)i * stride; uint8_t* o = out + (size_t)i * chunk;
        const size_t d = algo == ALGO_CHAMELEON ? decode_stream(ALGO_CHAMELEON, p, (size_t)sizes[i], o, len)
                       : algo == ALGO_CHEETAH ? decode_stream(ALGO_CHEETAH, p, (size_t)sizes[i], o, len) : decode_stream(ALGO_LION, p, (size_t)sizes[i], o, len);
        if (d != len) ++bad;
        tls_keep_tables = 0;
    }
    return bad;
}
