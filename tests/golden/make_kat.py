"""Regenerates tests/golden/kat.json from the C oracle (run from the repo root: python tests/golden/make_kat.py).

The three `reference` entries are NOT generated: they are the byte vectors asserted by the reference's own unit
tests (src/lib.rs:19,28,50,72) and are what pins the oracle.  The `derived` entries are oracle outputs recorded
after the oracle reproduced those three vectors and the independent table in SURVEY.md Appendix C; they pin later
refactors of the oracle and give the GPU tests fixed-size fixtures.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle  # noqa: E402
import datagen  # noqa: E402

REFERENCE_INPUT = b"test" * 31 + b"t"   # src/lib.rs:19
REFERENCE = {
    "chameleon": [0xfe, 0xff, 0xff, 0x7f, 0, 0, 0, 0, 116, 101, 115, 116] + [112, 251] * 30 + [116],   # src/lib.rs:28
    "cheetah": [244, 255, 255, 255, 255, 255, 255, 63, 116, 101, 115, 116, 112, 251, 116],              # src/lib.rs:50
    "lion": [112, 146, 36, 73, 146, 36, 116, 101, 115, 116, 112, 251, 73, 146, 36, 73, 146, 4, 116],    # src/lib.rs:72
}


def derived_inputs():
    xs = datagen.xs_bytes
    return {
        "empty": b"",
        "zeros1024": bytes(1024),
        "abcd300xyz": b"abcd" * 300 + b"xyz",
        "xs1_4099": xs(1, 4099),
        "words65536": b" ".join(b"w%03d" % (b % 200) for b in xs(2, 13200))[:65536],
        "prose100k": datagen.prose(100_000, seed=11).tobytes(),
        "mixed200k": datagen.mixed(200_000, seed=12).tobytes(),
        "samehash16k": datagen.same_hash_quads(4096, seed=13).tobytes(),
    }


def main():
    kat = {"reference_input_hex": REFERENCE_INPUT.hex(), "reference": {k: bytes(v).hex() for k, v in REFERENCE.items()}, "derived": {}}
    for name, data in derived_inputs().items():
        row = {"len": len(data), "sha256_input": hashlib.sha256(data).hexdigest()}
        for a in pyoracle.ALGOS:
            enc, st = pyoracle.encode_stats(a, data)
            row[a] = {"len": len(enc), "sha256": hashlib.sha256(enc).hexdigest(), "copy_blocks": st["copy_blocks"]}
        kat["derived"][name] = row
    with open(os.path.join(ROOT, "tests", "golden", "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    print(json.dumps(kat["derived"], indent=1)[:2000])


if __name__ == "__main__":
    main()
