"""bench.py's CPU legs without a GPU: the all-cores row (the oracle over the chunks of a buffer, OpenMP inside the oracle library) — its fields, its
chunked ratio against per-chunk oracle calls, the bit-exact comparison with "GPU" payloads (here: the oracle's own, and a corrupted one that must be
caught) — and the per-direction counter traffic that is only quoted for the workload and the library it was taken on."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench          # noqa: E402
import datagen        # noqa: E402
from oracle import pyoracle   # noqa: E402


@pytest.mark.parametrize("algo,chunk", [("chameleon", 1 << 18), ("cheetah", 98304), ("lion", 65536)])
def test_all_cores_row(algo, chunk):
    data = datagen.by_kind("mixed", (1 << 20) + 12345, seed=3)
    streams = [pyoracle.encode(algo, data[i:i + chunk]) for i in range(0, data.size, chunk)]
    row = bench.cpu_all_cores(data, chunk, algo, gpu_payloads=streams)
    assert "error" not in row, row
    assert row["n_chunks"] == len(streams) and row["gpu_chunks_compared_bit_exact"] == len(streams)
    assert row["threads"] >= 1 and row["value"] > 0 and row["slowest"] <= row["value"] <= row["fastest"] and row["samples"] == 7
    assert abs(row["ratio_chunked"] - data.size / sum(len(s) for s in streams)) < 1e-3
    wrong = list(streams)
    wrong[1] = wrong[1][:-1] + bytes([wrong[1][-1] ^ 1])
    with pytest.raises(AssertionError):
        bench.cpu_all_cores(data, chunk, algo, gpu_payloads=wrong)


def test_direction_traffic_is_only_quoted_for_its_workload():
    from density_amd import _lib
    chunk = int(_lib.lib().density_hip_auto_chunk_for(_lib.ALGO_IDS["cheetah"], 100_000_000))
    assert bench.direction_traffic("cheetah", 100_000_000, chunk + 256) == {}          # another chunk size
    assert bench.direction_traffic("cheetah", 99_999_744, chunk) == {}                 # another size
    assert bench.direction_traffic("chameleon", 100_000_000, chunk) == {}              # not a configuration the passes cover
    got = bench.direction_traffic("cheetah", 100_000_000, chunk)
    import glob, re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_directions.json")))
    lib_id = re.search(r"kernels ([0-9a-f]+)", _lib.lib().density_hip_version().decode()).group(1)
    if files and json.load(open(files[-1])).get("kernels_id") == lib_id:
        assert set(got) == {"encode", "decode"} and all(v[0] > 100_000_000 for v in got.values())
    else:
        assert got == {}                                                               # counters of another kernel generation are not this library's
