"""CPU-side checks of the product library: it loads, exports every symbol include/density_hip.h declares, and the
pure-arithmetic entry points (no device needed) agree with the reference formulae."""
import os
import re

from density_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "density_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", text, flags=re.M)
    return sorted(set(n for n in names if n.startswith(("chameleon_", "cheetah_", "lion_", "density_hip_"))))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    declared = _declared_symbols()
    assert len(declared) >= 9 + 14
    for name in declared:
        assert hasattr(L, name), name
    assert set(declared) == set(_lib.SYMBOLS), (set(declared) ^ set(_lib.SYMBOLS))


def test_reference_nine_symbols_present():
    """chameleon.rs:70-83, cheetah.rs:105-118, lion.rs:193-206"""
    L = _lib.lib()
    for a in ("chameleon", "cheetah", "lion"):
        for f in ("encode", "decode", "safe_encode_buffer_size"):
            assert hasattr(L, f"{a}_{f}")


def test_safe_encode_buffer_size_matches_reference_formula():
    """codec/codec.rs:18-21 — host arithmetic, runs without a GPU"""
    L = _lib.lib()
    geom = {"chameleon": (256, 8), "cheetah": (128, 8), "lion": (64, 6)}
    for a, (B, S) in geom.items():
        fn = getattr(L, f"{a}_safe_encode_buffer_size")
        for n in [0, 1, 3, 4, B - 1, B, B + 1, 10 * B, 10 * B + 5, 1 << 30, (1 << 33) + 17]:
            assert fn(n) == n + (n // B) * S + (S if n % B else 0)
    assert L.chameleon_safe_encode_buffer_size(1 << 30) == 1_107_296_256


def test_container_bound_arithmetic():
    L = _lib.lib()
    # one chunk: header 32 + table 4 -> block index at 48 (4 entries for 1000 bytes) -> payload at 64
    assert L.density_hip_container_bound(0, 1000, 1 << 20) == 64 + L.chameleon_safe_encode_buffer_size(1000)
    assert L.density_hip_container_bound(0, 0, 0) == 32
    assert L.density_hip_container_bound(0, 10, 100) == 0      # chunk must be a multiple of 256
    assert L.density_hip_container_bound(7, 10, 256) == 0
    n, c = 5 * 65536 + 123, 65536
    b = L.density_hip_container_bound(0, n, c)
    per = (L.chameleon_safe_encode_buffer_size(c) + 15) // 16 * 16
    index_at = (32 + 4 * 6 + 15) // 16 * 16
    payload_at = (index_at + (n + 255) // 256 + 15) // 16 * 16
    assert b == payload_at + 5 * per + L.chameleon_safe_encode_buffer_size(123)


def test_auto_chunk_policy():
    """density_hip_auto_chunk_for (host arithmetic): Chameleon in whole waves of 256 chunks of at most 4 MiB, in whole 4 KiB rounds, 64 KiB up to
    16 MiB of input; Cheetah one chunk per CU in 4 KiB trips between 64 KiB and 1 MiB; Lion the largest power of two with 700 streams."""
    L = _lib.lib()
    want = {10_192_446: 65536, 16 << 20: 65536, 100_000_000: 393_216, 256 << 20: 1 << 20, 1 << 30: 4 << 20, 3 << 29: 3 << 20, 2 << 30: 4 << 20}
    for n, c in want.items():
        assert L.density_hip_auto_chunk(n) == c == L.density_hip_auto_chunk_for(0, n), n
    for n in [1, 5_000_000, 20_000_000, 77_777_777, 999_999_999, (1 << 30) + 1, 5 << 30]:
        c = L.density_hip_auto_chunk(n)
        chunks = -(-n // c)
        assert 65536 <= c <= 4 << 20 and c % 4096 == 0
        if n > 16 << 20:
            waves = -(-n // (256 * (4 << 20)))
            assert 256 * (waves - 1) < chunks <= 256 * waves, (n, c, chunks)          # the last wave of work-groups is (almost) full
            assert chunks >= 0.93 * 256 * waves, (n, c, chunks)                      # (rounding a chunk up to 4 KiB costs a few chunks where chunks are small)
    assert L.density_hip_auto_chunk_for(1, 100_000_000) == 393_216
    assert L.density_hip_auto_chunk_for(2, 100_000_000) == 131072
    assert L.density_hip_auto_chunk_for(7, 100) == 0
