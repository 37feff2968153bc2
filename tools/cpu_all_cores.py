"""The all-cores CPU rows of the bench line on their own (no GPU work): the oracle over the chunks of configs 2 / 3 / 4, one chunk per OpenMP task:
    python tools/cpu_all_cores.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen, bench
prose = datagen.prose(100_000_000, seed=0xD1B54A32D192ED03)
for algo, data, chunk in (("chameleon", datagen.rep_text(256 << 20), 4 << 20), ("cheetah", prose, 393216), ("lion", prose, 131072)):
    r = bench.cpu_all_cores(data, chunk, algo)
    print(algo, {k: r.get(k) for k in ("value", "encode_MBps", "decode_MBps", "threads", "n_chunks", "ratio_chunked", "error")}, flush=True)
