// serial_codec.hip — Cheetah and Lion on the device, functional path.
//
// STATUS: correctness-first.  One LANE per chunk stream, the reference's scalar algorithm per lane, dictionary and
// predictor tables in global memory (Cheetah 768 KiB, Lion 1.75 MiB per stream: cheetah.rs:25-55, lion.rs:29-72 — neither
// fits LDS, and the predictor holds arbitrary quads, so the 16-bit packing of chameleon.hip does not apply to it).
// Parallelism is across chunks only; memory accesses are per-lane scattered.  This exists so that every algorithm of the
// path runs on the GPU bit-exactly behind the same C ABI; the wave-parallel, L2-resident design that replaces it is
// described in DESIGN.md §9.  The code is written against the format, with the reference lines it
// must agree with cited per function.
#include "common.hpp"
#include "kernels.hpp"

namespace density {

namespace {

struct Pair { uint32_t a, b; };

template <int ALGO> struct Geo;
template <> struct Geo<DENSITY_HIP_CHEETAH> {                 // cheetah.rs:17-23,188-196
    static constexpr uint32_t kFlagBits = 2, kSig = 8, kBlock = 128, kPredWords = 1;
};
template <> struct Geo<DENSITY_HIP_LION> {                    // lion.rs:17-27,317-325
    static constexpr uint32_t kFlagBits = 3, kSig = 6, kBlock = 64, kPredWords = 5;
};

template <int ALGO>
struct Tables {
    Pair* dict;          // 64 Ki x {a, b}
    uint32_t* pred;      // 64 Ki x kPredWords
    uint32_t last_hash;
    __device__ void clear() {
        uint4* p = reinterpret_cast<uint4*>(dict);
        const uint4 z = make_uint4(0, 0, 0, 0);
        for (uint32_t i = 0; i < 65536u * sizeof(Pair) / 16; ++i) p[i] = z;
        uint4* r = reinterpret_cast<uint4*>(pred);
        for (uint32_t i = 0; i < 65536u * Geo<ALGO>::kPredWords * 4u / 16; ++i) r[i] = z;
        last_hash = 0;
    }
};

__device__ __forceinline__ uint32_t hash16(uint32_t q) { return (q * kHashMul) >> 16; }

// ---- per-quad steps.  `item` receives the bytes to emit (0, 2 or 4), the return value is the flag. ----

// cheetah.rs:123-149
__device__ __forceinline__ uint32_t enc_quad(Tables<DENSITY_HIP_CHEETAH>& t, uint32_t q, uint32_t& item, uint32_t& item_len) {
    const uint32_t h = hash16(q);
    uint32_t* guess = t.pred + t.last_hash;
    Pair* e = t.dict + h;
    // both table reads are issued before either is needed (the dictionary entry is unused for a predicted quad): one memory
    // round trip per quad instead of two
    const uint32_t predicted = *guess;
    const Pair cur = *e;
    uint32_t flag;
    if (predicted == q) { flag = 3; item_len = 0; }
    else {
        if (cur.a == q) { flag = 1; item = h; item_len = 2; }
        else {
            if (cur.b == q) { flag = 2; item = h; item_len = 2; } else { flag = 0; item = q; item_len = 4; }
            *e = Pair{q, cur.a};
        }
        *guess = q;
    }
    t.last_hash = h;
    return flag;
}

// lion.rs:50-57,211-270
__device__ __forceinline__ uint32_t enc_quad(Tables<DENSITY_HIP_LION>& t, uint32_t q, uint32_t& item, uint32_t& item_len) {
    const uint32_t h = hash16(q);
    uint32_t* p = t.pred + 5u * t.last_hash;
    Pair* e = t.dict + h;
    uint32_t n0 = p[0], n1 = p[1], n2 = p[2], n3 = p[3], n4 = p[4];
    const Pair cur = *e;                                      // issued with the predictions: one memory round trip per quad
    uint32_t flag;
    item_len = 0;
    if (n0 == q) { flag = 1; }
    else if (n1 == q) { flag = 2; p[1] = n0; p[0] = q; }
    else if (n2 == q) { flag = 3; p[2] = n1; p[1] = n0; p[0] = q; }
    else if (n3 == q) { flag = 4; p[3] = n2; p[2] = n1; p[1] = n0; p[0] = q; }
    else {
        if (n4 == q) { flag = 5; }
        else {
            if (cur.a == q) { flag = 6; item = h; item_len = 2; }
            else {
                if (cur.b == q) { flag = 7; item = h; item_len = 2; } else { flag = 0; item = q; item_len = 4; }
                *e = Pair{q, cur.a};
            }
        }
        p[4] = n3; p[3] = n2; p[2] = n1; p[1] = n0; p[0] = q;     // shift_predictions
    }
    t.last_hash = h;
    return flag;
}

__device__ __forceinline__ uint32_t item_bytes(Tables<DENSITY_HIP_CHEETAH>&, uint32_t flag) { return flag == 0 ? 4u : (flag == 3 ? 0u : 2u); }
__device__ __forceinline__ uint32_t item_bytes(Tables<DENSITY_HIP_LION>&, uint32_t flag) { return flag == 0 ? 4u : (flag >= 6 ? 2u : 0u); }

// cheetah.rs:68-103,154-163 — `p` points at the item (if any)
__device__ __forceinline__ uint32_t dec_quad(Tables<DENSITY_HIP_CHEETAH>& t, uint32_t flag, const uint8_t* p) {
    uint32_t q, h;
    if (flag == 3) { q = t.pred[t.last_hash]; h = hash16(q); }
    else {
        if (flag == 0) { q = ld32u(p); h = hash16(q); Pair* e = t.dict + h; *e = Pair{q, e->a}; }
        else {
            h = ld16u(p);
            Pair* e = t.dict + h;
            const Pair cur = *e;
            if (flag == 1) q = cur.a; else { q = cur.b; *e = Pair{q, cur.a}; }
        }
        t.pred[t.last_hash] = q;
    }
    t.last_hash = h;
    return q;
}

// lion.rs:85-186,275-290
__device__ __forceinline__ uint32_t dec_quad(Tables<DENSITY_HIP_LION>& t, uint32_t flag, const uint8_t* p5) {
    uint32_t q, h;
    uint32_t* p = t.pred + 5u * t.last_hash;
    if (flag >= 1 && flag <= 5) {
        q = p[flag - 1];
        for (uint32_t k = flag - 1; k > 0; --k) p[k] = p[k - 1];     // move to front (A: nothing moves)
        if (flag > 1) p[0] = q;
        h = hash16(q);
    } else {
        if (flag == 0) { q = ld32u(p5); h = hash16(q); Pair* e = t.dict + h; *e = Pair{q, e->a}; }
        else {
            h = ld16u(p5);
            Pair* e = t.dict + h;
            const Pair cur = *e;
            if (flag == 6) q = cur.a; else { q = cur.b; *e = Pair{q, cur.a}; }
        }
        p[4] = p[3]; p[3] = p[2]; p[2] = p[1]; p[1] = p[0]; p[0] = q;
    }
    t.last_hash = h;
    return q;
}

template <int ALGO>
__device__ __forceinline__ void store_sig(uint8_t* rec, uint64_t sig) {           // codec.rs:24-26, lion.rs:334-337
    for (uint32_t i = 0; i < Geo<ALGO>::kSig; ++i) rec[i] = (uint8_t)(sig >> (8 * i));
}

// codec.rs:28-31; lion.rs:340-351 (6 significant bytes)
template <int ALGO>
__device__ __forceinline__ uint64_t load_sig(const uint8_t* p) {
    uint64_t v = 0;
    for (uint32_t i = 0; i < Geo<ALGO>::kSig; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
}

// ---- Codec::encode / encode_block, codec.rs:34-80 ----
template <int ALGO>
__global__ __launch_bounds__(64) void serial_encode_chunks(const uint8_t* __restrict__ in, uint64_t total, uint64_t chunk_bytes,
                                                           uint32_t n_chunks, uint8_t* __restrict__ out, uint64_t out_stride,
                                                           uint64_t* __restrict__ sizes, uint8_t* __restrict__ tables, uint32_t n_slots) {
    // `tables` arrives zeroed (the launcher clears the slots in use with one memset), so a lane clears its own tables
    // only before its second and later chunks
    using G = Geo<ALGO>;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    constexpr uint64_t kTableBytes = 65536ull * (sizeof(Pair) + 4ull * G::kPredWords);
    Tables<ALGO> t;
    t.dict = reinterpret_cast<Pair*>(tables + slot * kTableBytes);
    t.pred = reinterpret_cast<uint32_t*>(tables + slot * kTableBytes + 65536ull * sizeof(Pair));
    for (uint64_t chunk = slot; chunk < n_chunks; chunk += n_slots) {
        const uint8_t* src = in + chunk * chunk_bytes;
        const uint64_t len = (total - chunk * chunk_bytes) < chunk_bytes ? (total - chunk * chunk_bytes) : chunk_bytes;
        uint8_t* dst = out + chunk * out_stride;
        if (chunk != slot) t.clear();
        t.last_hash = 0;
        Guard guard;
        uint64_t opos = 0;
        for (uint64_t pos = 0; pos < len; pos += G::kBlock) {
            const uint32_t blen = (len - pos) < G::kBlock ? (uint32_t)(len - pos) : G::kBlock;
            const uint8_t* blk = src + pos;
            if (guard.block_is_copy()) {                                      // codec.rs:35-37
                for (uint32_t i = 0; i < blen; ++i) dst[opos + i] = blk[i];
                opos += blen;
                guard.decay();
                continue;
            }
            uint8_t* rec = dst + opos;
            uint64_t o = G::kSig, sig = 0;
            const uint32_t nq = blen >> 2;
            for (uint32_t k = 0; k < nq; ++k) {                               // codec.rs:42-57
                uint32_t item = 0, ilen = 0;
                const uint32_t flag = enc_quad(t, ld32u(blk + 4u * k), item, ilen);
                sig |= (uint64_t)flag << (G::kFlagBits * k);                  // io/write_signature.rs:14-17
                if (ilen == 2) st16u(rec + o, item); else if (ilen == 4) st32u(rec + o, item);
                o += ilen;
            }
            for (uint32_t i = 4u * nq; i < blen; ++i) rec[o++] = blk[i];     // codec.rs:58-61
            store_sig<ALGO>(rec, sig);
            guard.update(o >= G::kBlock);                                     // codec.rs:68
            opos += o;
        }
        sizes[chunk] = opos;
    }
}

// ---- Codec::decode, codec.rs:82-126 (fast and tail loops unified: per-unit checks are exact for both) ----
template <int ALGO>
__global__ __launch_bounds__(64) void serial_decode_chunks(const uint8_t* __restrict__ in, const uint64_t* __restrict__ offsets,
                                                           const uint64_t* __restrict__ sizes, uint32_t n_chunks,
                                                           uint8_t* __restrict__ out, uint64_t out_stride, uint64_t out_total,
                                                           uint32_t exact, uint64_t* __restrict__ produced, uint32_t* __restrict__ err,
                                                           uint8_t* __restrict__ tables, uint32_t n_slots) {
    using G = Geo<ALGO>;
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n_slots) return;
    constexpr uint64_t kTableBytes = 65536ull * (sizeof(Pair) + 4ull * G::kPredWords);
    constexpr uint64_t kMask = (1ull << G::kFlagBits) - 1;
    Tables<ALGO> t;
    t.dict = reinterpret_cast<Pair*>(tables + slot * kTableBytes);
    t.pred = reinterpret_cast<uint32_t*>(tables + slot * kTableBytes + 65536ull * sizeof(Pair));
    for (uint64_t chunk = slot; chunk < n_chunks; chunk += n_slots) {
        const uint8_t* src = in + offsets[chunk];
        const uint64_t elen = sizes[chunk];
        uint8_t* dst = out + chunk * out_stride;
        const uint64_t room_all = out_total - chunk * out_stride;
        const uint64_t cap = room_all < out_stride ? room_all : out_stride;
        if (chunk != slot) t.clear();
        t.last_hash = 0;
        Guard guard;
        uint64_t ipos = 0, opos = 0;
        bool bad = false, done = false;
        while (ipos < elen && !bad && !done) {
            const uint64_t rem = elen - ipos;
            if (guard.block_is_copy()) {                                      // codec.rs:89-91,103-110
                const uint32_t take = rem > G::kBlock ? G::kBlock : (uint32_t)rem;
                if (opos + take > cap) { bad = true; break; }
                for (uint32_t i = 0; i < take; ++i) dst[opos + i] = src[ipos + i];
                ipos += take; opos += take;
                if (rem <= G::kBlock) break;
                guard.decay();
                continue;
            }
            if (rem < G::kSig) { bad = true; break; }                         // read_signature would panic
            const uint64_t mark = ipos;
            uint64_t sig = load_sig<ALGO>(src + ipos);
            ipos += G::kSig;
            for (uint32_t k = 0; k < G::kBlock / 4; ++k) {
                const uint32_t flag = (uint32_t)(sig & kMask);
                sig >>= G::kFlagBits;
                const uint64_t left = elen - ipos;
                if (flag == 0 && left < 4) {                                  // cheetah.rs:168-176, lion.rs:295-303: end of data
                    if (opos + left > cap) { bad = true; break; }
                    for (uint32_t i = 0; i < left; ++i) dst[opos + i] = src[ipos + i];
                    opos += left; ipos += left; done = true;
                    break;
                }
                const uint32_t need = item_bytes(t, flag);
                if (left < need || opos + 4 > cap) { bad = true; break; }
                const uint32_t q = dec_quad(t, flag, src + ipos);
                ipos += need;
                st32u(dst + opos, q);
                opos += 4;
            }
            if (!done && !bad) guard.update(ipos - mark >= G::kBlock);        // codec.rs:98,122
        }
        if (exact && !bad && opos != cap) bad = true;
        produced[chunk] = opos;
        if (bad) atomicOr(err, 1u);
    }
}

}  // namespace

uint64_t serial_table_bytes(int algo) { return 65536ull * (sizeof(Pair) + 4ull * (algo == DENSITY_HIP_LION ? 5 : 1)); }

hipError_t launch_serial_encode(int algo, const uint8_t* d_in, uint64_t total, uint64_t chunk_bytes, uint32_t n_chunks, uint8_t* d_out,
                                uint64_t out_stride, uint64_t* d_sizes, uint8_t* d_tables, uint32_t n_slots, hipStream_t stream) {
    if (n_chunks == 0) return hipSuccess;
    const uint32_t blocks = (n_slots + 63) / 64;
    hipError_t e = hipMemsetAsync(d_tables, 0, (size_t)n_slots * serial_table_bytes(algo), stream);
    if (e != hipSuccess) return e;
    if (algo == DENSITY_HIP_CHEETAH)
        hipLaunchKernelGGL(serial_encode_chunks<DENSITY_HIP_CHEETAH>, dim3(blocks), dim3(64), 0, stream, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots);
    else
        hipLaunchKernelGGL(serial_encode_chunks<DENSITY_HIP_LION>, dim3(blocks), dim3(64), 0, stream, d_in, total, chunk_bytes, n_chunks, d_out, out_stride, d_sizes, d_tables, n_slots);
    return hipGetLastError();
}

hipError_t launch_serial_decode(int algo, const uint8_t* d_in, const uint64_t* d_offsets, const uint64_t* d_sizes, uint32_t n_chunks,
                                uint8_t* d_out, uint64_t out_stride, uint64_t out_total, bool exact, uint64_t* d_produced, uint32_t* d_err,
                                uint8_t* d_tables, uint32_t n_slots, hipStream_t stream) {
    if (n_chunks == 0) return hipSuccess;
    const uint32_t blocks = (n_slots + 63) / 64;
    hipError_t e = hipMemsetAsync(d_tables, 0, (size_t)n_slots * serial_table_bytes(algo), stream);
    if (e != hipSuccess) return e;
    if (algo == DENSITY_HIP_CHEETAH)
        hipLaunchKernelGGL(serial_decode_chunks<DENSITY_HIP_CHEETAH>, dim3(blocks), dim3(64), 0, stream, d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact ? 1u : 0u, d_produced, d_err, d_tables, n_slots);
    else
        hipLaunchKernelGGL(serial_decode_chunks<DENSITY_HIP_LION>, dim3(blocks), dim3(64), 0, stream, d_in, d_offsets, d_sizes, n_chunks, d_out, out_stride, out_total, exact ? 1u : 0u, d_produced, d_err, d_tables, n_slots);
    return hipGetLastError();
}

}  // namespace density
