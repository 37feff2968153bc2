"""Seeded synthetic inputs shared by tests and bench.py.  numpy only."""
import numpy as np

_M64 = (1 << 64) - 1
SALT_MUL = 0xB5AD   # density_amd/csrc/chameleon_dev.hpp::kSaltMul (the GPU table's per-slot salt: internal, but the adversarial inputs below aim at it)


def xs_bytes(seed, n):
    """xorshift64* byte stream exactly as SURVEY.md Appendix C defines it (pure Python; small n)."""
    s, out = seed, bytearray()
    while len(out) < n:
        s ^= s >> 12
        s ^= (s << 25) & _M64
        s ^= s >> 27
        out += ((s * 0x2545F4914F6CDD1D) & _M64).to_bytes(8, "little")
    return bytes(out[:n])


def random_bytes(n, seed=1):
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8)


def prose(n, seed=0x9E3779B97F4A7C15, vocab=4096, zipf_s=1.0):
    """Pseudo-English: Zipf(s) words from a `vocab`-word lowercase vocabulary, single spaces, '. ' every 8-20 words.

    Vectorised stand-in for the generator sketched in SURVEY.md §8(d) (numpy PCG64 instead of xorshift64*, so the
    bytes are defined by THIS function and its seed; dickens/enwik8 are not available in this environment).
    """
    rng = np.random.default_rng(seed & _M64)
    lens = rng.integers(2, 11, size=vocab)
    words = [bytes(rng.integers(97, 123, size=l, dtype=np.uint8)) for l in lens]
    p = 1.0 / np.arange(1, vocab + 1) ** zipf_s
    p /= p.sum()
    out = bytearray()
    while len(out) < n:
        k = max(1024, min(1 << 20, (n - len(out)) // 5 + 64))
        idx = rng.choice(vocab, size=k, p=p)
        gaps = rng.integers(8, 21, size=k // 8 + 2).cumsum()
        stops = set(int(g) for g in gaps if g < k)
        parts = []
        for j, w in enumerate(idx):
            parts.append(words[w])
            parts.append(b". " if j in stops else b" ")
        out += b"".join(parts)
    return np.frombuffer(bytes(out[:n]), dtype=np.uint8).copy()


def rep_text(n, period=1_000_003, seed=0x9E3779B97F4A7C15):
    """`rep-text`: a prime-length pseudo-English period tiled to n bytes (config 2 of BASELINE.json at n = 2**30)."""
    base = prose(min(period, n), seed)
    reps = -(-n // base.size)
    return np.tile(base, reps)[:n].copy()


def same_hash_quads(n_quads, seed=3):
    """Distinct quads that all hash to ONE dictionary slot (intra-wave hazard worst case).

    hash(q) = (q * M mod 2^32) >> 16 with M = 2*M', M' odd: q = (t * inv(M')) mod 2^31 has product 2t, so every
    t < 2^15 lands in slot 0, and adding 2^31 does not change the product.
    """
    inv = pow(0x9D6EF916 >> 1, -1, 1 << 31)
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 1 << 15, size=n_quads, dtype=np.uint64)
    top = rng.integers(0, 2, size=n_quads, dtype=np.uint64) << np.uint64(31)
    q = ((t * np.uint64(inv)) & np.uint64(0x7FFFFFFF)) | top
    return q.astype("<u4").view(np.uint8).copy()


def mixed(n, seed=5):
    """text / random / zeros / text segments: drives the copy-mode FSM in and out."""
    rng = np.random.default_rng(seed)
    parts, left = [], n
    kinds = 0
    while left > 0:
        k = int(min(left, rng.integers(300, 9000)))
        kind = kinds % 4
        if kind == 0:
            parts.append(prose(k, seed + kinds))
        elif kind == 1:
            parts.append(rng.integers(0, 256, size=k, dtype=np.uint8))
        elif kind == 2:
            parts.append(np.zeros(k, dtype=np.uint8))
        else:
            parts.append(rng.integers(0, 4, size=k, dtype=np.uint8))
        left -= k
        kinds += 1
    return np.concatenate(parts)[:n].copy()


def low_zero_quads(n_quads, seed=4):
    """Quads whose 16-bit dictionary entry packs to 0 outside slot 0 (low 15 bits and top bit clear, e.g. the bytes
    00 00 xx yy / 00 80 xx yy with yy < 0x80): the one value that aliases a never-written slot in the GPU table
    (density_amd/csrc/chameleon.hip header).  Mixed with zero quads and ordinary values, with repeats."""
    rng = np.random.default_rng(seed)
    pool = (rng.integers(0, 1 << 16, size=48, dtype=np.uint64) << np.uint64(15)).astype(np.uint64)
    pool = np.concatenate([pool, np.zeros(4, np.uint64), rng.integers(0, 1 << 32, size=12, dtype=np.uint64)])
    q = pool[rng.integers(0, pool.size, size=n_quads)]
    return q.astype("<u4").view(np.uint8).copy()


def salted_zero_quads(n_quads, seed=7):
    """Quads whose STORED dictionary entry is 0 outside slot 0 in the GPU table (packed entry == slot_salt(slot), see
    density_amd/csrc/chameleon_dev.hpp): the one value that aliases a never-written slot and goes through the zero-entry map.
    Mixed with ordinary quads and repeats so that hits, misses and overwrites of such slots all occur."""
    inv = pow(0x9D6EF916 >> 1, -1, 1 << 31)
    rng = np.random.default_rng(seed)
    hs = rng.integers(1, 1 << 16, size=40, dtype=np.uint64)
    salt = (hs * np.uint64(SALT_MUL)) & np.uint64(0xFFFF)         # chameleon_dev.hpp::slot_salt
    pfull = (hs << np.uint64(16)) | (salt & np.uint64(0xFFFE))
    special = (((pfull >> np.uint64(1)) * np.uint64(inv)) & np.uint64(0x7FFFFFFF)) | ((salt & np.uint64(1)) << np.uint64(31))
    # sanity: they hash to their slot
    assert np.all(((special * np.uint64(0x9D6EF916)) & np.uint64(0xFFFFFFFF)) >> np.uint64(16) == hs)
    others = rng.integers(0, 1 << 32, size=24, dtype=np.uint64)
    # quads colliding with the special slots but with other entries
    coll = (((((hs[:16] << np.uint64(16)) | np.uint64(0x1234)) >> np.uint64(1)) * np.uint64(inv)) & np.uint64(0x7FFFFFFF))
    pool = np.concatenate([special, others, coll, np.zeros(2, np.uint64)])
    q = pool[rng.integers(0, pool.size, size=n_quads)]
    return q.astype("<u4").view(np.uint8).copy()


def binaryish(n, seed=6):
    """Little-endian 32-bit records with small values and zero padding: lots of 00 00 xx 00 style quads."""
    rng = np.random.default_rng(seed)
    vals = rng.choice(np.array([0, 1, 2, 65536, 32768, 98304, 0x10000 * 7, 0x8000 * 5, 255, 256], dtype=np.uint32), size=n // 4 + 1)
    noise = rng.integers(0, 1 << 32, size=n // 4 + 1, dtype=np.uint64).astype(np.uint32)
    pick = rng.random(n // 4 + 1) < 0.1
    return np.where(pick, noise, vals).astype("<u4").view(np.uint8)[:n].copy()


def by_kind(kind, n, seed=1):
    if kind == "prose":
        return prose(n, seed)
    if kind == "random":
        return random_bytes(n, seed)
    if kind == "zeros":
        return np.zeros(n, dtype=np.uint8)
    if kind == "mixed":
        return mixed(n, seed)
    if kind == "samehash":
        return same_hash_quads(n // 4 + 1, seed)[:n].copy()
    if kind == "lowzero":
        return low_zero_quads(n // 4 + 1, seed)[:n].copy()
    if kind == "saltzero":
        return salted_zero_quads(n // 4 + 1, seed)[:n].copy()
    if kind == "binaryish":
        return binaryish(n, seed)
    if kind == "rep":
        return rep_text(n, period=100_003, seed=seed)
    raise ValueError(kind)
