#!/usr/bin/env python
"""benches/density.py — the reference's bench harness (benches/density.rs:11-136, benches/utils.rs:4-18) for this library.

Same shape as `cargo bench`: the file is taken from the FILE environment variable (default ./benches/data/dickens.txt; when neither
exists — no corpus ships with this repository — the named synthetic stand-in `synth-prose-10M`, tests/datagen.py, is used and said
so), the compression ratio is printed per algorithm, and every benchmark takes 25 samples (divan `sample_count = 25`), reported as
fastest | slowest | median | mean with throughput = uncompressed bytes / time in both directions (divan BytesCount of the input slice).

Three columns of rows per algorithm:
  gpu/stream     {Algo}::encode / ::decode through the reference's own C symbols on HOST buffers (one reference stream, H2D + kernel + D2H)
  gpu/container  the chunked container on DEVICE-resident buffers (kernel time only: the data-parallel path)
  cpu/oracle     the C restatement of the Rust reference, one thread — the stand-in for the reference's own row (no rustc in this image)

    FILE=path python benches/density.py [--samples 25] [--algos chameleon,cheetah,lion] [--chunk BYTES]
"""
import argparse
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DEFAULT_FILE_PATH = "./benches/data/dickens.txt"     # benches/utils.rs:4


def file_bytes():
    path = os.environ.get("FILE", DEFAULT_FILE_PATH)
    if os.path.exists(path):
        data = np.fromfile(path, dtype=np.uint8)
        print(f"Using file \x1b[1m{path}\x1b[0m (\x1b[37m{data.size} bytes\x1b[0m)")
        return data
    import datagen
    data = datagen.prose(10_192_446, seed=0x9E3779B97F4A7C15)       # dickens' size
    print(f"Using file \x1b[1msynth-prose-10M\x1b[0m (\x1b[37m{data.size} bytes\x1b[0m)  [{path} not present: synthetic stand-in, tests/datagen.py::prose]")
    return data


def fmt_time(s):
    return f"{s * 1e3:.4g} ms" if s >= 1e-3 else f"{s * 1e6:.4g} µs"


def fmt_rate(b, s):
    r = b / s
    return f"{r / 1e9:.4g} GB/s" if r >= 1e9 else f"{r / 1e6:.4g} MB/s"


def report(name, n, samples, ratio=None, last=False):
    fastest, slowest = min(samples), max(samples)
    median, mean = statistics.median(samples), statistics.fmean(samples)
    tag = f"({ratio:.3f}x)" if ratio else ""
    head = "╰─" if last else "├─"
    cont = " " if last else "│"
    print(f"│  {head} {name:<18}{tag:>9}   {fmt_time(fastest):<13} │ {fmt_time(slowest):<13} │ {fmt_time(median):<13} │ {fmt_time(mean):<13} │ {len(samples):<7} │ {len(samples)}")
    print(f"│  {cont} {'':<27}   {fmt_rate(n, fastest):<13} │ {fmt_rate(n, slowest):<13} │ {fmt_rate(n, median):<13} │ {fmt_rate(n, mean):<13} │         │")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=25)                  # benches/density.rs:11 sample_count = 25
    ap.add_argument("--algos", default="chameleon,cheetah,lion")
    ap.add_argument("--chunk", type=int, default=0, help="container chunk size (0 = the library default)")
    ap.add_argument("--no-gpu", action="store_true", help="CPU rows only")
    args = ap.parse_args()
    data = file_bytes()
    n = data.size
    from oracle import pyoracle
    gpu = not args.no_gpu
    if gpu:
        import torch
        from density_amd import BY_NAME, container
        x = torch.from_numpy(data).cuda()
        tstream = torch.cuda.Stream()             # (a real stream: handle 0 would mean "the library's internal stream" to density_hip_*_device)
        torch.cuda.set_stream(tstream)
        stream = tstream.cuda_stream
    print(f"{'density':<34} fastest       │ slowest       │ median        │ mean          │ samples │ iters")
    for algo in args.algos.split(","):
        print(f"├─ {algo:<45} │               │               │               │         │")
        # --- cpu/oracle (whole stream, 1 thread) ---
        cap = pyoracle.safe_encode_buffer_size(algo, n)
        enc = np.empty(cap, dtype=np.uint8)
        dec = np.empty(n, dtype=np.uint8)
        esize = pyoracle.encode_into(algo, data.ctypes.data, n, enc.ctypes.data, cap)
        te, td = [], []
        for _ in range(args.samples):
            t0 = time.perf_counter(); pyoracle.encode_into(algo, data.ctypes.data, n, enc.ctypes.data, cap); te.append(time.perf_counter() - t0)
        for _ in range(args.samples):
            t0 = time.perf_counter(); got = pyoracle.decode_into(algo, enc.ctypes.data, esize, dec.ctypes.data, n); td.append(time.perf_counter() - t0)
        assert got == n and np.array_equal(dec, data)
        report("cpu/oracle compress", n, te, n / esize)
        report("cpu/oracle decompr.", n, td, last=not gpu)
        if not gpu:
            continue
        # --- gpu/stream: the reference's symbols on host buffers ---
        C = BY_NAME[algo]
        genc = np.empty(C.safe_encode_buffer_size(n), dtype=np.uint8)
        gdec = np.empty(n, dtype=np.uint8)
        m = C.encode(data, genc)
        assert genc[:m].tobytes() == enc[:esize].tobytes(), "GPU stream differs from the oracle's"
        te, td = [], []
        for _ in range(args.samples):
            t0 = time.perf_counter(); C.encode(data, genc); te.append(time.perf_counter() - t0)
        for _ in range(args.samples):
            t0 = time.perf_counter(); k = C.decode(genc[:m], gdec); td.append(time.perf_counter() - t0)
        assert k == n and np.array_equal(gdec, data)
        report("gpu/stream compress", n, te, n / m)
        report("gpu/stream decompr.", n, td)
        # --- gpu/container: device-resident, events around the calls ---
        chunk = args.chunk
        capc = container.container_bound(algo, n, chunk)
        cont = torch.empty(capc, dtype=torch.uint8, device="cuda")
        back = torch.empty(n, dtype=torch.uint8, device="cuda")
        hdr = container.encode_device(algo, x.data_ptr(), n, cont.data_ptr(), capc, chunk, stream=stream)
        assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=stream) == n and torch.equal(back, x)
        te, td = [], []
        for _ in range(args.samples):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); container.encode_device(algo, x.data_ptr(), n, cont.data_ptr(), capc, chunk, stream=stream, want_header=False); e1.record()
            torch.cuda.synchronize(); te.append(e0.elapsed_time(e1) * 1e-3)
        for _ in range(args.samples):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=stream, sync=False); e1.record()
            torch.cuda.synchronize(); td.append(e0.elapsed_time(e1) * 1e-3)
        report(f"gpu/container compress", n, te, n / hdr.container_len)
        report(f"gpu/container decompr.", n, td, last=True)
    print("(gpu/container: chunk", args.chunk or "default", "bytes; ratios: whole stream for cpu/oracle and gpu/stream, chunked container incl. tables for gpu/container)")


if __name__ == "__main__":
    main()
