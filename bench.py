#!/usr/bin/env python
"""bench.py — Chameleon encode+decode round trip on device-resident buffers (BASELINE.json metric).

One "step" = one container encode followed by one container decode of this rank's synthetic buffer (inputs already
resident in HBM).  `value` = uncompressed bytes processed by ALL ranks / wall time of the K timed steps, in MB/s
(= N / (t_enc + t_dec), the round-trip throughput of SURVEY.md §8d, matching divan's BytesCount of the uncompressed
slice in both directions, benches/density.rs:29,48).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size BYTES] [--chunk BYTES] [--algo chameleon|cheetah|lion]

N > 1: one rank per GPU over RCCL.  Launched by the driver through torch.distributed.run, or — when WORLD_SIZE is not set —
by this script itself, which re-executes under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N`.  The path
shards by chunks with no data-path collective, so scaling is weak: every rank encodes/decodes its own buffer.
`--no-gpu` is a dry mode for the CPU test of the launcher and of the distributed bookkeeping (gloo, tiny oracle-built
containers, no timing claims).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

# the all-cores CPU rows: one OpenMP thread per core, where it was started (read by the OpenMP runtime when oracle/libdensity_oracle.so loads)
os.environ.setdefault("OMP_PROC_BIND", "close")
os.environ.setdefault("OMP_PLACES", "cores")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
METRIC = "MB/s encode+decode (round-trip) per GPU + compression ratio, dickens/enwik8"


def cpu_all_cores(src, chunk, algo, gpu_payloads=None):
    """All host cores, one chunk per task (OpenMP inside the oracle library: oracle_encode_chunks_mt): what a chunked CPU build of the same container does
    (SURVEY.md 8d "N-thread run over the same chunks").  Every chunk it encodes is compared with the GPU's stream of that chunk when `gpu_payloads` is given."""
    from oracle import pyoracle
    try:
        n = src.size
        dec = np.empty(n, dtype=np.uint8)
        ncpu = os.cpu_count() or 1
        nchunks = (n + chunk - 1) // chunk
        # one thread per PHYSICAL core at most (os.cpu_count() counts SMT siblings): measured on the GPU box's host, 256 logical cores — Cheetah 0.39 MiB chunks
        # 1 / 8 / 32 / 64 / 128 threads: 0.77 / 5.9 / 21.7 / 39.3 / 57.4 GB/s encode, and 0.5 GB/s on 255 (every logical core taken: the team's spinning waiters
        # and the interpreter's own threads starve the workers)
        threads = max(1, min(ncpu // 2, nchunks))
        ccap = (pyoracle.safe_encode_buffer_size(algo, chunk) + 63) // 64 * 64
        encs = np.empty((nchunks, ccap), dtype=np.uint8)
        sizes = np.zeros(nchunks, dtype=np.uint64)
        enc = lambda: pyoracle.encode_chunks_mt(algo, src.ctypes.data, n, chunk, encs.ctypes.data, ccap, sizes.ctypes.data, threads)
        dcd = lambda: pyoracle.decode_chunks_mt(algo, encs.ctypes.data, ccap, sizes.ctypes.data, dec.ctypes.data, n, chunk, threads)
        assert enc() == 0 and dcd() == 0                                                      # warm (pages of the buffers, the thread team)
        te, td = [], []
        for _ in range(7):
            time.sleep(0.3)                                                                   # (a pause between the bursts: the GPU box's host throttles all-core bursts that follow one another — round 6 saw a median of 1.1 GB/s beside a fastest of 29)
            t0 = time.perf_counter(); enc(); t1 = time.perf_counter(); dcd(); t2 = time.perf_counter()
            te.append(t1 - t0); td.append(t2 - t1)
        assert np.array_equal(dec, src)
        # The MEDIAN of seven, threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores: set at the top of this file, before the OpenMP runtime starts), with the
        # fastest and the slowest beside it — round 5 printed the fastest of five, and on a host that throttles bursts of all its cores (the same call: 157 GB/s,
        # then 11) a minimum says what the box can do once, not what it does.
        rt = sorted(a + b for a, b in zip(te, td))
        e, d = sorted(te)[3], sorted(td)[3]
        out = {"value": round(n / rt[3] / 1e6, 1), "unit": "MB/s", "cores": ncpu, "threads": threads, "encode_MBps": round(n / e / 1e6, 1), "decode_MBps": round(n / d / 1e6, 1),
               "fastest": round(n / rt[0] / 1e6, 1), "slowest": round(n / rt[-1] / 1e6, 1), "samples": 7,
               "ratio_chunked": round(n / int(sizes.sum()), 4), "chunk": chunk, "n_chunks": nchunks, "timing": "median of 7 round trips 0.3 s apart (fastest / slowest beside it), threads pinned to cores",
               "sample": f"{n} B in {nchunks} chunks of {chunk} B, one chunk per task on {threads} OpenMP threads ({ncpu} host cores), C restatement (oracle/density_oracle.c)"}
        if gpu_payloads is not None:
            m = min(nchunks, len(gpu_payloads))
            bad = [i for i in range(m) if bytes(encs[i][:int(sizes[i])]) != gpu_payloads[i]]
            assert not bad, f"GPU chunk streams differ from the oracle: chunks {bad[:8]}"
            out["gpu_chunks_compared_bit_exact"] = m
        return out
    except AssertionError:
        raise
    except Exception as ex:  # pragma: no cover
        return {"error": str(ex)}


def cpu_baseline(host, chunk, sample_bytes, algo="chameleon", reps=25, gpu_payloads=None):
    """Times the CPU oracle (C restatement of the Rust reference, single thread like the reference's bench) on a bounded
    sample of the same workload.  Checker/baseline only — never part of the measured GPU path.  `gpu_payloads`: the GPU's chunk
    streams of the same buffer; every chunk the all-cores leg encodes is compared with them (full-coverage parity at bench size)."""
    from oracle import pyoracle
    n = min(sample_bytes, host.size)
    src = np.ascontiguousarray(host[:n])
    cap = pyoracle.safe_encode_buffer_size(algo, n)
    enc = np.empty(cap, dtype=np.uint8)
    dec = np.empty(n, dtype=np.uint8)
    te, td = [], []
    esize = 0
    pyoracle.encode_into(algo, src.ctypes.data, min(n, 1 << 20), enc.ctypes.data, cap)      # (first touch of the library and of the buffers' pages)
    for _ in range(reps):                                                                 # the reference bench's protocol: sample_count = 25 (benches/density.rs:11), median and fastest
        t0 = time.perf_counter()
        esize = pyoracle.encode_into(algo, src.ctypes.data, n, enc.ctypes.data, cap)
        t1 = time.perf_counter()
        got = pyoracle.decode_into(algo, enc.ctypes.data, esize, dec.ctypes.data, n)
        t2 = time.perf_counter()
        assert got == n
        te.append(t1 - t0); td.append(t2 - t1)
    assert np.array_equal(dec, src)
    med_e, med_d, best_e, best_d = sorted(te)[reps // 2], sorted(td)[reps // 2], min(te), min(td)
    out = {"value": round(n / (med_e + med_d) / 1e6, 1), "unit": "MB/s", "cores": 1, "kind": "port",
           "sample": f"first {n >> 20} MiB of the rank-0 buffer, whole-stream {algo} encode+decode, {reps} samples (median; fastest beside it), "
                     f"C restatement of density-rs 0.16.6 (oracle/density_oracle.c), 1 thread",
           "samples": reps, "timing": "median",
           "encode_MBps": round(n / med_e / 1e6, 1), "decode_MBps": round(n / med_d / 1e6, 1),
           "fastest": {"value": round(n / (best_e + best_d) / 1e6, 1), "encode_MBps": round(n / best_e / 1e6, 1), "decode_MBps": round(n / best_d / 1e6, 1)},
           "ratio_whole_stream": round(n / esize, 4)}
    # ... and all cores over the chunks of the WHOLE buffer (every chunk the GPU made is compared: full-coverage parity at bench size)
    out["all_cores"] = cpu_all_cores(np.ascontiguousarray(host), chunk, algo, gpu_payloads)
    return out


def host_api_rates(algo, host, chunk, sample_bytes):
    """PCIe-inclusive rates of the host-pointer entry points on a bounded sample (never `value`): the reference's own symbols
    (chameleon_encode / _decode: ONE reference stream = one work-group, include/density_hip.h section 1) and the container API."""
    import time as _t
    from density_amd import BY_NAME, container
    n = min(sample_bytes, host.size)
    src = np.ascontiguousarray(host[:n])
    codec = BY_NAME[algo]
    out = {"sample_bytes": int(n)}
    enc = np.empty(codec.safe_encode_buffer_size(n), dtype=np.uint8)
    dec = np.empty(n, dtype=np.uint8)
    m = codec.encode(src, enc)                               # warm both directions (staging buffers, scratch, the output's pages, self-test)
    codec.decode(enc[:m], dec)
    te, td = [], []
    for _ in range(3):
        t0 = _t.perf_counter(); m = codec.encode(src, enc); t1 = _t.perf_counter()
        k = codec.decode(enc[:m], dec); t2 = _t.perf_counter()
        te.append(t1 - t0); td.append(t2 - t1)
    assert k == n and np.array_equal(dec, src)
    t_e, t_d = sorted(te)[1], sorted(td)[1]
    out["reference_symbols"] = {"encode_MBps": round(n / t_e / 1e6, 1), "decode_MBps": round(n / t_d / 1e6, 1),
                                "round_trip_MBps": round(n / (t_e + t_d) / 1e6, 1), "timing": "median of 3 warm calls",
                                "note": "ONE reference stream; H2D + kernels + D2H.  Chameleon encode of >= 4 MiB runs in parallel segments and is still the "
                                        "reference's stream byte for byte; decode of >= 2 MiB runs in parallel segments too (unless it is mostly raw copies); "
                                        "from 32 MiB (encode) / 16 MiB of stream (decode) on the transfers run in slices beside the kernels"}
    if algo == "chameleon":
        # the same single stream with the buffers already on the device (density_hip_stream_encode_device): what the segments buy
        import ctypes
        import torch
        from density_amd import _lib
        lib = _lib.lib()
        d_in = torch.from_numpy(src).cuda()
        d_out = torch.empty(enc.size + 64, dtype=torch.uint8, device="cuda")
        size = ctypes.c_size_t(0)
        call = lambda: lib.density_hip_stream_encode_device(0, ctypes.c_void_p(d_in.data_ptr()), n, ctypes.c_void_p(d_out.data_ptr()), d_out.numel(), None, ctypes.byref(size))
        assert call() == 0
        torch.cuda.synchronize(); t0 = _t.perf_counter()
        for _ in range(3):
            assert call() == 0
        torch.cuda.synchronize(); t1 = _t.perf_counter()
        same = bytes(d_out[:size.value].cpu().numpy()) == bytes(enc[:m])
        out["reference_symbols"]["device_resident_encode_MBps"] = round(3 * n / (t1 - t0) / 1e6, 1)
        out["reference_symbols"]["device_resident_encode_is_the_same_stream"] = bool(same)
        d_back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
        back = ctypes.c_size_t(0)
        dcall = lambda: lib.density_hip_stream_decode_device(0, ctypes.c_void_p(d_out.data_ptr()), size.value, ctypes.c_void_p(d_back.data_ptr()), n, None, ctypes.byref(back))
        assert dcall() == 0
        torch.cuda.synchronize(); t0 = _t.perf_counter()
        for _ in range(3):
            assert dcall() == 0
        torch.cuda.synchronize(); t1 = _t.perf_counter()
        out["reference_symbols"]["device_resident_decode_MBps"] = round(3 * n / (t1 - t0) / 1e6, 1)
        out["reference_symbols"]["device_resident_decode_is_the_input"] = bool(back.value == n and torch.equal(d_back[:n], d_in))
    def container_leg(data, ck):
        m_ = data.size
        cont = np.empty(container.container_bound(algo, m_, ck), dtype=np.uint8)
        back_ = np.empty(m_, dtype=np.uint8)
        cn = container.encode(algo, data, cont, ck)          # warm both directions
        container.decode(cont[:cn], back_)
        te, td = [], []
        for _ in range(3):
            t0 = _t.perf_counter(); cn = container.encode(algo, data, cont, ck); t1 = _t.perf_counter()
            k_ = container.decode(cont[:cn], back_); t2 = _t.perf_counter()
            te.append(t1 - t0); td.append(t2 - t1)
        assert k_ == m_ and np.array_equal(back_, data)
        t_e, t_d = sorted(te)[1], sorted(td)[1]
        return {"bytes": int(m_), "encode_MBps": round(m_ / t_e / 1e6, 1), "decode_MBps": round(m_ / t_d / 1e6, 1),
                "round_trip_MBps": round(m_ / (t_e + t_d) / 1e6, 1), "chunk": int(container.parse_header(cont[:32]).chunk_size), "timing": "median of 3 warm calls"}
    # The container calls on the sample at the chunk the library picks for a buffer of that size (what a caller of density_hip_encode(.., 0)
    # gets), and on a larger buffer: Chameleon inputs worth three slices go up, through the kernels and down in slices on separate streams
    # with the caller's buffers pinned in place (api.hip: *_container_pipelined); smaller ones are staged whole.
    out["container"] = dict(container_leg(src, 0), note="chunked container at the automatic chunk; H2D + kernels + D2H, pipelined in slices where the input is worth three")
    big = np.ascontiguousarray(host[:min(host.size, 4 * n)])
    if big.size > n:
        out["container_large"] = dict(container_leg(big, chunk), note="the same at four times the sample, the headline run's chunk size")
    return out


def size_sweep(container, algo, x, sizes, steps=5):
    """Round-trip throughput of smaller inputs with the automatic chunk size (density_hip_auto_chunk): a chunk is one work-group, so inputs with
    fewer chunks than CUs cannot fill the device; labelled cache-resident vs HBM-bound as SURVEY.md §8d asks."""
    import torch
    from density_amd import _lib
    out = []
    stream = torch.cuda.current_stream().cuda_stream
    for n in sizes:
        if n > x.numel():
            continue
        chunk = int(_lib.lib().density_hip_auto_chunk_for({"chameleon": 0, "cheetah": 1, "lion": 2}[algo], n))
        cap = container.container_bound(algo, n, chunk)
        cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
        back = torch.empty(n, dtype=torch.uint8, device="cuda")
        hdr = container.encode_device(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=stream)
        assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=stream) == n and torch.equal(back, x[:n])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            container.encode_device(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=stream, want_header=False)
            container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=stream, sync=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
        out.append({"bytes": n, "auto_chunk": chunk, "n_chunks": int(hdr.n_chunks), "round_trip_MBps": round(n / dt / 1e6, 1),
                    "compression_ratio": round(n / hdr.container_len, 4),
                    "residency": "HBM-bound" if 2 * n > (256 << 20) else "cache-resident (256 MiB Infinity Cache)"})
    return out



def hostile_data(kind, n):
    """SURVEY.md 8d's correctness inputs, at bench speed: zeros; xorshift64* random (tests/test_gpu_shipped_configs.py's generator); `mixed` = runs of
    300..9000 bytes cycling through text / random / zeros / two-bit noise like tests/datagen.py::mixed, cut from pools instead of generated run by run."""
    import datagen
    if kind == "zeros":
        return np.zeros(n, dtype=np.uint8)
    if kind == "random":
        from test_gpu_shipped_configs import xorshift_bytes
        return xorshift_bytes(n)
    rng = np.random.default_rng(5)
    text = datagen.prose(8 << 20, seed=77)
    noise = rng.integers(0, 256, size=8 << 20, dtype=np.uint8)
    low = rng.integers(0, 4, size=8 << 20, dtype=np.uint8)
    out = np.zeros(n, dtype=np.uint8)
    lens = rng.integers(300, 9000, size=n // 300 + 8)
    at, k = 0, 0
    while at < n:
        ln = int(min(lens[k], n - at))
        pool = (text, noise, None, low)[k % 4]
        if pool is not None:
            o = int(rng.integers(0, pool.size - ln))
            out[at:at + ln] = pool[o:o + ln]
        at += ln; k += 1
    return out


def roofline_entry(name, alg_bytes, ms, traffic=None, traffic_source=None):
    ach = alg_bytes / (ms * 1e-3) / 1e9 if ms else 0.0
    return {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 5),
            "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": int(alg_bytes), "kernel_avg_ms": round(ms, 4)}


def direction_traffic(algo, n, chunk):
    """Counter traffic per encode / decode CALL of configs 3 / 4 from the committed passes (probes/profile_directions.sh -> profiles/rNN_pmc_directions.json:
    all kernels and fills of a direction summed) — quoted only for the workload they were taken on (100 MB of prose at the automatic chunk) and the library
    whose kernels id they record."""
    try:
        import glob as _glob, re as _re
        from density_amd import _lib
        cand = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_directions.json")))
        if not cand or n != 100_000_000 or chunk != int(_lib.lib().density_hip_auto_chunk_for(_lib.ALGO_IDS[algo], n)):
            return {}
        pm = json.load(open(cand[-1]))
        m = _re.search(r"kernels ([0-9a-f]+)", _lib.lib().density_hip_version().decode())
        if algo not in pm["configs"] or not m or pm.get("kernels_id") != m.group(1):
            return {}
        src = os.path.relpath(cand[-1], ROOT)
        return {d: (int(pm["configs"][algo][d]["hbm_bytes_corrected"]), src) for d in ("encode", "decode")}
    except Exception:
        return {}


def settle(step, ms):
    """Runs `step` untimed for about `ms` milliseconds of wall clock and returns how many steps that took.  After an idle phase (here: the
    oracle comparisons on the host just before the timing) an MI355X takes ~25-40 ms of continuous work to come back to its sustained clock —
    tools/gpu_ramp.py: the first 5 steps of the headline workload run 13 % slower than the steady state, steps 5..25 5 % slower, everything
    after that within 0.05 % for seconds.  The timed steps are meant to show that steady state, so the ramp is run through first;
    `settle_ms` / `settle_steps` in the bench line say how much of it there was."""
    import torch
    if ms <= 0:
        return 0
    t0, n = time.perf_counter(), 0
    while (time.perf_counter() - t0) * 1e3 < ms:
        step(); n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return n


def other_config(container, algo, label, host, steps=3, warmup=1, cpu_sample=32 << 20, chunk=0, settle_ms=0.0, all_cores=False):
    """One of BASELINE's other configurations, measured like the headline workload (device-resident container encode + decode, HIP events
    around every kernel) with a bounded CPU sample beside it.  Returns a dict for the `other_configs` list of the bench line."""
    import torch
    from density_amd import _lib
    from oracle import pyoracle
    n = host.size
    chunk = chunk or int(_lib.lib().density_hip_auto_chunk_for(_lib.ALGO_IDS[algo], n))
    x = torch.from_numpy(host).cuda()
    cap = max(container.container_bound_slotted(algo, n, chunk), container.container_bound_paged(algo, n, chunk) if algo == "chameleon" else 0)
    cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    hdr = container.encode_device(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
    assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s) == n and torch.equal(back, x), f"{algo}: round trip mismatch"
    E = int(hdr.container_len)                                                         # the packed container: the algorithmic E
    _, payloads = container.chunk_payloads(cont[:E].cpu().numpy())
    # Timed: the PACKED container — encode, size scan, stitch, decode — i.e. the wire-ready form whose size `compression_ratio` states (round 6: until then these
    # rows timed the slotted form and printed the packed ratio beside it).  The slotted form (no stitch, not wire-ready) is timed for a few steps beside it.
    # Chameleon shapes the PAGED form serves (two chunks and more of 1 MiB .. 4 MiB) are timed on it instead — wire-ready as the encode kernel leaves it, no stitch.
    back.zero_(); torch.cuda.synchronize()                 # (a null caller stream means the library's own stream: not ordered behind torch's memset)
    enc_fn, form, Et = container.encode_device, "packed", E
    if algo == "chameleon":
        hdr_p = container.encode_device_paged(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
        if hdr_p.flags & container.FLAG_PAGED:
            enc_fn, form, hdr, Et = container.encode_device_paged, "paged", hdr_p, int(hdr_p.container_len)
            _, pp = container.chunk_payloads(cont[:Et].cpu().numpy())
            assert pp == payloads, "paged chunk streams differ from the packed form's"

    def step():
        enc_fn(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
        container.decode_device(cont.data_ptr(), Et, back.data_ptr(), n, header=hdr, stream=s, sync=False)

    def timed(fn, k):
        torch.cuda.synchronize()
        container.set_profiling(True)
        container.last_timings()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = container.last_timings()
        container.set_profiling(False)
        return dt, tm

    settle(step, settle_ms)
    for _ in range(warmup):
        step()
    dt, timings = timed(step, steps)
    assert torch.equal(back, x), f"{algo}: round trip mismatch after the timed steps"
    tot = {}
    for name, ms in timings:
        tot[name] = tot.get(name, 0.0) + ms / steps
    # beside it: the slotted form of the same chunk streams
    hdr_s = container.encode_device_slotted(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
    Ec = int(hdr_s.container_len)

    def step_slotted():
        container.encode_device_slotted(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
        container.decode_device(cont.data_ptr(), Ec, back.data_ptr(), n, header=hdr_s, stream=s, sync=False)
    back.zero_(); torch.cuda.synchronize()
    step_slotted()
    dt_s, _ = timed(step_slotted, steps)
    assert torch.equal(back, x), f"{algo}: round trip mismatch (slotted)"
    dec_names = ("layout_decode", f"{algo}_decode_chunks")
    t_dec = sum(v for k, v in tot.items() if k in dec_names)
    t_enc = sum(v for k, v in tot.items() if k not in dec_names)
    # CPU beside it: the oracle, one thread, on a bounded sample; every chunk of the sample compared with the GPU's payloads
    m = min(cpu_sample, n) // chunk * chunk or min(cpu_sample, n)
    src = np.ascontiguousarray(host[:m])
    ccap = pyoracle.safe_encode_buffer_size(algo, m)
    enc = np.empty(ccap, dtype=np.uint8)
    dec = np.empty(m, dtype=np.uint8)
    c0 = time.perf_counter(); es = pyoracle.encode_into(algo, src.ctypes.data, m, enc.ctypes.data, ccap)
    c1 = time.perf_counter(); got = pyoracle.decode_into(algo, enc.ctypes.data, es, dec.ctypes.data, m); c2 = time.perf_counter()
    assert got == m and np.array_equal(dec, src)
    nchk = max(m // chunk, 1)
    bad = [i for i in range(min(nchk, len(payloads))) if payloads[i] != pyoracle.encode(algo, host[i * chunk:(i + 1) * chunk])]
    assert not bad, f"{algo}: GPU chunk streams differ from the oracle: {bad[:8]}"
    # ... and on all host cores over the SAME chunks (the whole buffer, every chunk held against the GPU's stream of it)
    cores = cpu_all_cores(np.ascontiguousarray(host), chunk, algo, gpu_payloads=payloads) if all_cores else None
    return {"config": label, "algorithm": algo, "bytes": int(n), "chunk_bytes": int(chunk), "n_chunks": int(hdr.n_chunks),
            "value": round(n * steps / dt / 1e6, 1), "unit": "MB/s", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 4),
            "compression_ratio": round(n / Et, 4), "encoded_bytes": Et, "encoded_bytes_packed": E,
            "container_form": ("paged (wire-ready as the encode kernel leaves it; compression_ratio is of these bytes, the roofline's E is the packed size)" if form == "paged" else
                               "packed (wire-ready: encode + size scan + stitch + decode are all inside the timed steps; compression_ratio is of these bytes)"),
            "value_slotted": round(n * steps / dt_s / 1e6, 1),
            "encode_ms": round(t_enc, 4), "decode_ms": round(t_dec, 4), "kernel_ms": {k: round(v, 4) for k, v in tot.items()},
            "residency": "HBM-bound" if 2 * n > (256 << 20) else "cache-resident (fits the 256 MiB Infinity Cache)",
            "roofline": {"encode": roofline_entry(f"{algo} encode (all kernels of the direction)", n + E, t_enc, *direction_traffic(algo, n, chunk).get("encode", (None, None))),
                         "decode": roofline_entry(f"{algo} decode (all kernels of the direction)", n + E, t_dec, *direction_traffic(algo, n, chunk).get("decode", (None, None)))},
            "cpu_baseline": {"value": round(m / (c2 - c0) / 1e6, 1), "unit": "MB/s", "cores": 1, "kind": "port",
                             "encode_MBps": round(m / (c1 - c0) / 1e6, 1), "decode_MBps": round(m / (c2 - c1) / 1e6, 1),
                             "ratio_whole_stream": round(m / es, 4),
                             "sample": f"first {m >> 20} MiB, whole-stream {algo} encode+decode, 1 thread, C restatement of density-rs 0.16.6",
                             "gpu_chunks_compared_bit_exact": int(min(nchk, len(payloads))), "all_cores": cores}}


def strict_stream_leg(host, x, steps=5, label=None):
    """The reference's own call shape on the headline buffer: ONE Chameleon stream over the whole input (chameleon.rs:45-53), encoded and
    decoded in parallel segments on the device, buffers device-resident.  Byte-identity with the reference's stream is the
    GPU suite's to show at full size (tests/test_gpu_chameleon.py::test_config2_full_size_strict_stream...); here: decode == input, and the
    stream of the first 64 MiB — a prefix of the whole stream, blocks being coded in order — equals the oracle's."""
    import ctypes
    import torch
    from density_amd import Chameleon, _lib
    from oracle import pyoracle
    lib = _lib.lib()
    n = host.size
    d_out = torch.empty(Chameleon.safe_encode_buffer_size(n) + 64, dtype=torch.uint8, device="cuda")
    d_back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    size, back = ctypes.c_size_t(0), ctypes.c_size_t(0)
    enc = lambda: lib.density_hip_stream_encode_device(0, ctypes.c_void_p(x.data_ptr()), n, ctypes.c_void_p(d_out.data_ptr()), d_out.numel(), None, ctypes.byref(size))
    assert enc() == 0
    dec = lambda: lib.density_hip_stream_decode_device(0, ctypes.c_void_p(d_out.data_ptr()), size.value, ctypes.c_void_p(d_back.data_ptr()), n, None, ctypes.byref(back))
    assert dec() == 0 and back.value == n and torch.equal(d_back[:n], x), "strict stream: decode != input"
    m = min(64 << 20, n) // 256 * 256
    want = pyoracle.encode("chameleon", host[:m])
    # (the whole stream's records of the first m bytes: the oracle's stream of that prefix, unless the prefix ends inside the stream's last record)
    k = len(want) if m < n else size.value
    assert bytes(d_out[:k].cpu().numpy()) == want[:k], "strict stream: prefix differs from the oracle's stream"
    # (the GPU has idled during the oracle call above: a dozen untimed round trips bring its clocks back before anything is timed; then the
    # median of `steps` individually timed calls per direction)
    for _ in range(12):                                                               # (~40 ms: tools/gpu_ramp.py)
        assert enc() == 0 and dec() == 0
    te, td = [], []
    for _ in range(steps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        assert enc() == 0
        torch.cuda.synchronize(); t1 = time.perf_counter()
        assert dec() == 0
        torch.cuda.synchronize(); t2 = time.perf_counter()
        te.append(t1 - t0); td.append(t2 - t1)
    E = int(size.value)
    e_ms, d_ms = sorted(te)[len(te) // 2] * 1e3, sorted(td)[len(td) // 2] * 1e3
    return {"config": label or "2 (strict): ONE reference stream over the whole buffer, chameleon_encode / chameleon_decode shape, device-resident",
            "bytes": int(n), "encoded_bytes": E, "compression_ratio": round(n / E, 4), "encode_ms": round(e_ms, 4), "decode_ms": round(d_ms, 4),
            "value": round(n / ((e_ms + d_ms) * 1e-3) / 1e6, 1), "unit": "MB/s", "timing": "wall clock around each call (host orchestration of the passes included), median of 5",
            "oracle_prefix_compared_bytes": int(k), "decode_is_the_input": True,
            "roofline": {"encode": roofline_entry("strict stream encode (all passes)", n + E, e_ms), "decode": roofline_entry("strict stream decode (all passes)", n + E, d_ms)}}

def self_launch(args):
    """--gpus N without a torchrun environment: start the N ranks ourselves (one node, 127.0.0.1 rendezvous)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def dry_run(args, world, rank):
    """--no-gpu: the launcher and the distributed bookkeeping on CPU tensors under gloo (tests/test_parallel_gloo.py).  Local containers
    come from the oracle (test infrastructure); nothing is timed."""
    import torch
    import torch.distributed as dist
    import datagen
    from density_amd import parallel
    from test_parallel_gloo import cpu_container
    n, chunk = min(args.size, 64 << 10), 4096
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        assert dist.get_world_size() == world
    data = datagen.mixed(n, seed=100 + rank)
    local = torch.frombuffer(bytearray(cpu_container(data, chunk)), dtype=torch.uint8)
    hdr, table, index, payload = parallel.parse_local(local)
    n_gpus = 1
    glob = {"container_len": hdr["container_len"]}
    if world > 1:
        lay = parallel.exchange_layout(hdr["n_chunks"], payload.numel(), n, torch.device("cpu"))
        glob = parallel.global_layout(lay, chunk, hdr["flags"])
        merged = parallel.concat_to_rank0(local, chunk)
        assert (merged is not None) == (rank == 0)
        n_gpus = dist.get_world_size()
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": 0.0, "unit": "MB/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
                          "dry_run": True, "scaling": "weak", "global_container_bytes": int(glob["container_len"])}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--settle-ms", type=float, default=100.0,
                    help="untimed continuous running before the warm-up steps, so that the GPU is back at its sustained clock after the host-side checks (0: none)")
    ap.add_argument("--size", type=int, default=1 << 30, help="bytes per GPU (default 1 GiB = BASELINE config 2)")
    ap.add_argument("--chunk", type=int, default=0, help="container chunk size; 0 = the library's automatic choice (density_hip_auto_chunk: 4 MiB at 1 GiB)")
    ap.add_argument("--cpu-sample", type=int, default=256 << 20)
    ap.add_argument("--host-sample", type=int, default=64 << 20, help="bytes for the PCIe-inclusive host-API rates")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline and the host-API legs")
    ap.add_argument("--no-sweep", action="store_true", help="skip the size sweep (profiling runs: only the headline workload's launches)")
    ap.add_argument("--no-extra", action="store_true", help="skip the other_configs legs (configs 3/4, strict stream): only the headline workload")
    ap.add_argument("--no-gpu", action="store_true", help="dry mode: launcher + distributed bookkeeping on CPU/gloo (tests)")
    ap.add_argument("--packed", action="store_true", help="time the packed container (encode + stitch pass + decode) instead of the paged one")
    ap.add_argument("--slotted", action="store_true", help="time the slotted container (rounds 3-4's `value`: no stitch, not wire-ready) instead of the paged one")
    ap.add_argument("--concat", action="store_true", help="also time the optional gather-to-rank-0 stitch (N > 1)")
    ap.add_argument("--variant", type=int, default=0, help="kernel variant bit mask (density_hip_set_kernel_variant): 0 = default")
    ap.add_argument("--algo", default="chameleon", choices=["chameleon", "cheetah", "lion"],
                    help="chameleon is the headline workload; cheetah/lion: use a smaller --size")
    ap.add_argument("--data", default="rep-text", choices=["rep-text", "prose"],
                    help="rep-text = BASELINE config 2 (period 1000003 B); prose = non-periodic synth-prose (configs 3/4 stand-in)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks"
    if args.no_gpu:
        return dry_run(args, world, rank)

    import torch
    import torch.distributed as dist
    import datagen
    from density_amd import container
    from oracle import pyoracle

    # One rank per GPU over RCCL whenever a launcher set the rendezvous up — including a world of ONE (WORLD_SIZE=1 in the environment:
    # the whole process-group branch — init, the size all-gather, the optional concat — then runs on a single GPU, which is how
    # tests/test_gpu_rccl.py exercises it on the 1-GPU box); a plain `python bench.py` is a single process without a group.
    use_pg = world > 1 or (os.environ.get("WORLD_SIZE") == "1" and "MASTER_PORT" in os.environ)
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, "RCCL world size differs from --gpus"
    else:
        torch.cuda.set_device(0)
    n_gpus = dist.get_world_size() if use_pg else 1

    n = args.size
    from density_amd import _lib
    chunk = args.chunk or int(_lib.lib().density_hip_auto_chunk_for({"chameleon": 0, "cheetah": 1, "lion": 2}[args.algo], n))
    container.set_kernel_variant(args.variant)
    # config 2 / 5 of BASELINE.json: rep-text, per-shard seed = seed + rank; configs 3/4 stand-in: non-periodic prose
    if args.data == "rep-text":
        host = datagen.rep_text(n, seed=0x9E3779B97F4A7C15 + rank)
    else:
        host = datagen.prose(n, seed=0xD1B54A32D192ED03 + rank)
    x = torch.from_numpy(host).cuda()
    algo = args.algo
    cap = max(container.container_bound_slotted(algo, n, chunk), container.container_bound_paged(algo, n, chunk))
    cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    ws_size = max(int(density_ws(container, n, chunk, args.algo)), 1)
    ws = torch.empty(ws_size, dtype=torch.uint8, device="cuda")
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    # The timed container is the PAGED form (include/density_hip.h DENSITY_HIP_FLAG_PAGED; round 5): the encoder places every chunk stream itself, in
    # 64 KiB pages taken from one counter, so what it leaves in HBM IS the wire blob (header + tables + page directory + pages in use) — two codec
    # launches, no stitch pass — and the decoder reads the pages in place.  `value`, `encoded_bytes` and `compression_ratio` describe THOSE bytes.
    # --slotted times rounds 3-4's form (worst-case slots, not wire-ready), --packed the packed form (encode + stitch + decode); both are also measured
    # for a few steps beside the headline (`slotted_container`, `packed_container`).
    form = "packed" if args.packed else ("slotted" if args.slotted else "paged")
    enc_dev = {"packed": container.encode_device, "slotted": container.encode_device_slotted, "paged": container.encode_device_paged}[form]

    # correctness before any timing: decode(encode(x)) == x, and chunk streams equal to the oracle's (all of the CPU sample's chunks are
    # compared in cpu_baseline; here a spread of chunks so that --no-cpu runs are checked too)
    hdr_p = container.encode_device(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, workspace=(ws.data_ptr(), ws_size))
    got = container.decode_device(cont.data_ptr(), hdr_p.container_len, back.data_ptr(), n, header=hdr_p, stream=s, workspace=(ws.data_ptr(), ws_size))
    assert got == n and torch.equal(back, x), "round trip mismatch"
    raw = cont[:hdr_p.container_len].cpu().numpy()
    _, payloads = container.chunk_payloads(raw)
    for i in sorted(set([0, hdr_p.n_chunks // 3, hdr_p.n_chunks - 1])):
        assert payloads[i] == pyoracle.encode(algo, host[i * chunk:(i + 1) * chunk]), f"chunk {i} differs from the oracle"
    E = int(hdr_p.container_len)                       # the PACKED container: the algorithmic E of the roofline (the fewest bytes the streams + tables can take)
    hdr, Ec = hdr_p, E
    packed_ref = cont[:E].clone()
    if form != "packed":
        # the timed form: decodes to the input, and its chunk streams — read the way a CPU reader of the container reads them (slots / page
        # directory: container.chunk_payloads) — are byte for byte the packed container's, i.e. the oracle's
        back.zero_(); torch.cuda.synchronize()         # (a null caller stream means the library's own stream: not ordered behind torch's memset)
        hdr = enc_dev(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, workspace=(ws.data_ptr(), ws_size))
        Ec = int(hdr.container_len)
        if form == "paged" and not (hdr.flags & container.FLAG_PAGED):
            form = "slotted"                            # shapes the paged form is not for come out slotted (the header says so)
        got = container.decode_device(cont.data_ptr(), Ec, back.data_ptr(), n, header=hdr, stream=s, workspace=(ws.data_ptr(), ws_size))
        assert got == n and torch.equal(back, x), f"round trip mismatch ({form} container)"
        _, payloads_f = container.chunk_payloads(cont[:Ec].cpu().numpy())
        assert payloads_f == payloads, f"the {form} container's chunk streams differ from the packed container's"
        del payloads_f
        if form == "slotted":
            repacked = torch.empty(cap, dtype=torch.uint8, device="cuda")
            hr = container.pack_device(cont.data_ptr(), Ec, repacked.data_ptr(), cap, header=hdr, stream=s, workspace=(ws.data_ptr(), ws_size))
            assert hr.container_len == E and torch.equal(repacked[:E], packed_ref), "pack(slotted) differs from the packed container"
            del repacked
    # the bytes `value` is about: what has to leave the device for the container to be decodable elsewhere
    wire_bytes = Ec if form in ("paged", "packed") else E
    del raw

    def step():
        enc_dev(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, workspace=(ws.data_ptr(), ws_size), want_header=False)
        container.decode_device(cont.data_ptr(), Ec, back.data_ptr(), n, header=hdr, stream=s, workspace=(ws.data_ptr(), ws_size), sync=False)

    settle_steps = settle(step, args.settle_ms)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    container.set_profiling(True)      # HIP events around every kernel, on the launch stream, inside the timed region
    container.last_timings()           # drain
    if use_pg:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_pg:
        dist.barrier()
    dt_local = dt = time.perf_counter() - t0
    timings = container.last_timings()
    container.set_profiling(False)
    per_rank_ms = [dt_local / args.steps * 1e3]
    if use_pg:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        all_t = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(all_t, t)
        per_rank_ms = [float(v.item()) / args.steps * 1e3 for v in all_t]
        dt = max(float(v.item()) for v in all_t)
    assert torch.equal(back, x), "round trip mismatch after timed steps"
    # the other container forms through the same round trip, a few steps each, for the record (never `value`)
    def side_form(fname):
        fenc = {"packed": container.encode_device, "slotted": container.encode_device_slotted, "paged": container.encode_device_paged}[fname]
        fh = fenc(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, workspace=(ws.data_ptr(), ws_size))
        flen = int(fh.container_len)

        def fstep():
            fenc(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, workspace=(ws.data_ptr(), ws_size), want_header=False)
            container.decode_device(cont.data_ptr(), flen, back.data_ptr(), n, header=fh, stream=s, workspace=(ws.data_ptr(), ws_size), sync=False)
        settle(fstep, args.settle_ms / 2)
        fstep(); torch.cuda.synchronize()
        container.set_profiling(True); container.last_timings()
        tp0 = time.perf_counter()
        for _ in range(5):
            fstep()
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - tp0) / 5
        pk = {}
        for name, ms in container.last_timings():
            pk[name] = pk.get(name, 0.0) + ms / 5
        container.set_profiling(False)
        assert torch.equal(back, x)
        notes = {"packed": "encode + stitch (compact: 2E bytes, none of them algorithmic) + decode of the packed container: the densest wire form",
                 "slotted": "rounds 3-4's `value`: chunk streams left in worst-case slots (1.03 x N of address space), not wire-ready until density_hip_pack_device",
                 "paged": "streams in 64 KiB pages from one counter: wire-ready without a stitch"}
        return {"value": round(n_gpus * n / dtp / 1e6, 1), "unit": "MB/s", "ms_per_step": round(dtp * 1e3, 4), "kernel_ms": {k: round(v, 4) for k, v in pk.items()},
                "whole_path_hbm_frac": round(2.0 * (n + E) / dtp / 1e9 / HBM_PEAK_GBS, 5), "container_bytes": flen,
                "wire_ready": fname != "slotted", "note": notes[fname]}
    side = {}
    if not args.no_extra:
        for fname in ("packed", "slotted", "paged"):
            if fname != form and not (fname == "paged" and form == "slotted" and not args.slotted):
                side[fname] = side_form(fname)
    packed_cmp = side.get("packed")

    # the path's only collective: all-gather of per-shard (chunks, payload bytes) -> offsets in the global container.  The bookkeeping below is done
    # on the PACKED form of this rank's container (parallel.py's layout arithmetic is the packed one; a paged shard is wire-ready as it stands and
    # would travel as one blob per rank)
    from density_amd import parallel
    local_cont = packed_ref
    hdr_l, table_l, index_l, payload_l = parallel.parse_local(local_cont)
    if use_pg:
        torch.cuda.synchronize(); tg0 = time.perf_counter()
        lay = parallel.exchange_layout(hdr_l["n_chunks"], payload_l.numel(), n, x.device)
        torch.cuda.synchronize(); gather_ms = (time.perf_counter() - tg0) * 1e3
        glob = parallel.global_layout(lay, chunk, hdr_l["flags"])
    else:
        gather_ms, glob = 0.0, {"container_len": E, "n_chunks": int(hdr.n_chunks), "total_len": n}
    # config 5's wire form (round 6): the ranks' TIMED blobs as they stand behind a super-header and a row per rank ("DHCM", include/density_hip.h) — one more
    # all-gather of two u64 per rank; nothing is re-encoded or moved here
    algo_id = {"chameleon": 0, "cheetah": 1, "lion": 2}[algo]
    if use_pg:
        torch.cuda.synchronize(); tm0 = time.perf_counter()
        _, mrows, mlen = parallel.exchange_multi_layout(Ec, n, x.device, algo_id, chunk)
        torch.cuda.synchronize(); multi_gather_ms = (time.perf_counter() - tm0) * 1e3
    else:
        (_, mrows, mlen), multi_gather_ms = parallel.multi_layout([Ec], [n], algo_id, chunk), 0.0
    concat_ms, concat_checked = None, None
    if use_pg and args.concat:
        torch.cuda.synchronize(); dist.barrier(); tc0 = time.perf_counter()
        merged = parallel.concat_to_rank0(local_cont, chunk)
        torch.cuda.synchronize(); dist.barrier(); concat_ms = (time.perf_counter() - tc0) * 1e3
        if rank == 0:
            # the stitched global container decodes on this GPU to the ranks' inputs one after the other; with one rank it IS the local container
            assert merged.numel() == glob["container_len"]
            if n_gpus == 1:
                assert torch.equal(merged, local_cont), "world-size-1 concat differs from the local container"
            mh = container.parse_header(bytes(merged[:32].cpu().numpy()))
            gback = torch.empty(int(mh.total_len), dtype=torch.uint8, device="cuda")
            assert container.decode_device(merged.data_ptr(), merged.numel(), gback.data_ptr(), gback.numel(), header=mh, stream=s) == mh.total_len
            assert torch.equal(gback[:n], x), "rank 0's part of the stitched container does not decode to its input"
            concat_checked = True
            del gback
        del merged

    if rank == 0:
        per = {}
        for name, ms in timings:
            per.setdefault(name, []).append(ms)
        avg = {k: sum(v) / len(v) for k, v in per.items()}                 # per launch
        tot = {k: sum(v) / args.steps for k, v in per.items()}            # per step
        launches = {k: len(v) / args.steps for k, v in per.items()}
        # algorithmic bytes per step: encode reads N writes E, decode reads E writes N; the stitch pass
        # (layout + compact) moves no algorithmic bytes — it is overhead that lowers the whole-path fraction.
        alg = {f"{algo}_encode_chunks": n + E, f"{algo}_decode_chunks": n + E}
        dom = max(alg, key=lambda k: tot.get(k, 0.0))
        alg_launch = alg[dom] / max(launches.get(dom, 1.0), 1.0)
        ach = alg_launch / (avg[dom] * 1e-3) / 1e9 if avg.get(dom) else 0.0
        # HBM traffic of that kernel from the committed PMC passes (rocprofv3 cannot run inside this process); only quoted
        # when the profile was taken on this exact workload and kernel generation
        traffic, traffic_src = None, None
        import re as _re
        _m = _re.search(r"kernels ([0-9a-f]+)", _lib.lib().density_hip_version().decode())
        kernels_id = _m.group(1) if _m else None
        try:
            import glob as _glob
            cand = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")))
            if cand and n == 1 << 30 and chunk == 4 << 20 and args.variant == 0 and algo == "chameleon" and args.data == "rep-text":
                pm = json.load(open(cand[-1]))
                key = dom.replace("_chunks", "")
                match = [k for k in pm["kernels"] if key in k and pm["kernels"][k].get("default_path")]
                # ... and on THIS kernel generation: the summary records the library's kernels id (density_hip_version()); counters of other
                # kernels are not quoted as this run's traffic
                if pm.get("kernels_id") != kernels_id:
                    traffic_src = f"{os.path.relpath(cand[-1], ROOT)} is of kernels {pm.get('kernels_id')}, this library is {kernels_id}: not quoted"
                    match = []
                if match:
                    traffic = int(pm["kernels"][match[0]]["hbm_bytes_corrected"])   # per launch, like `achieved`
                    traffic_src = os.path.relpath(cand[-1], ROOT)
        except Exception:
            pass
        ms_step = dt / args.steps * 1e3
        t_enc = sum(tot.get(k, 0.0) for k in (f"{algo}_encode_chunks", "layout_encode", "compact", "stitch_tail"))
        t_dec = sum(tot.get(k, 0.0) for k in ("layout_decode", f"{algo}_decode_chunks"))
        label = "rep-text (BASELINE config 2: synthetic repeating text, period 1000003 B)" if args.data == "rep-text" else \
                "synth-prose (non-periodic; stand-in for enwik8, BASELINE configs 3/4)"
        residency = "HBM-bound (buffers exceed the 256 MiB Infinity Cache)" if 2 * n > (256 << 20) else "cache-resident (fits the 256 MiB Infinity Cache)"
        result = {
            "metric": METRIC,
            "value": round(n_gpus * n * args.steps / dt / 1e6, 1),
            "unit": "MB/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "settle_ms": args.settle_ms, "settle_steps": settle_steps,
            "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{algo} {label}, {n >> 20} MiB per GPU, device-resident container encode+decode, chunk {chunk >> 10} KiB, {residency}",
                       "algorithm": algo, "bytes_per_gpu": n, "chunk_bytes": chunk, "n_chunks": int(hdr.n_chunks),
                       "parallelism": f"chunk-sharded x{n_gpus}, no data-path collective"},
            "compression_ratio": round(n / wire_bytes, 4),
            "encoded_bytes": int(wire_bytes),
            "compression_ratio_packed": round(n / E, 4), "encoded_bytes_packed": E,
            "value_packed": (packed_cmp["value"] if packed_cmp else None),
            "whole_path_hbm_frac_packed": (packed_cmp["whole_path_hbm_frac"] if packed_cmp else None),
            "value_slotted": (side["slotted"]["value"] if "slotted" in side else None),
            "value_definition": {"paged": "`value`: the round trip through the PAGED device-resident container — two codec launches, no stitch pass, and what the encoder leaves "
                                          "in HBM is the wire blob (`encoded_bytes` = its length, `compression_ratio` = N over it: the same bytes `value` is timed on).  "
                                          "`value_packed`: the same through the PACKED container (encode + stitch + decode; `encoded_bytes_packed`, the densest form); "
                                          "`value_slotted`: rounds 3-4's `value` (slots of the worst case, not wire-ready).  The roofline's algorithmic E is the packed size.",
                                 "slotted": "`value`: the round trip through the SLOTTED device-resident container (two codec launches, no stitch; its encoded form spans "
                                            "1.03 x N of address space until packed: `encoded_bytes` states the packed size)",
                                 "packed": "`value`: the round trip through the PACKED container (encode + stitch + decode)"}[form],
            "kernels_id": kernels_id,
            "container_form": form,
            "packed_container": packed_cmp,
            "slotted_container": side.get("slotted"),
            "paged_container": side.get("paged"),
            "encode_ms": round(t_enc, 4), "decode_ms": round(t_dec, 4),
            "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms],
            "kernel_ms": {k: round(v, 4) for k, v in tot.items()},
            "kernel_launches_per_step": {k: round(v, 2) for k, v in launches.items()},
            "whole_path_hbm_frac": round((2.0 * (n + E)) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": int(alg_launch), "kernel_avg_ms": round(avg[dom], 4),
                         "launches_per_step": round(launches.get(dom, 1.0), 2),
                         # the whole timed step against the same roofline: 2 (N + E) algorithmic bytes over ms_per_step, for the timed form and for the others
                         "whole_path": {"form": form, "frac": round((2.0 * (n + E)) / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "value": round(n_gpus * n * args.steps / dt / 1e6, 1),
                                        "frac_packed": (packed_cmp["whole_path_hbm_frac"] if packed_cmp else None), "value_packed": (packed_cmp["value"] if packed_cmp else None),
                                        "frac_slotted": (side["slotted"]["whole_path_hbm_frac"] if "slotted" in side else None),
                                        "value_slotted": (side["slotted"]["value"] if "slotted" in side else None)}},
            "multi_gpu": {"process_group": ("nccl (RCCL)" if use_pg else None), "size_gather_ms": round(gather_ms, 3),
                          "concat_to_rank0_ms": (round(concat_ms, 3) if concat_ms is not None else None), "concat_decodes_to_input": concat_checked,
                          "global_container_bytes": int(glob["container_len"]),
                          "multi_container": {"format": "DHCM: the ranks' timed blobs as they stand behind a 32-byte header and a 24-byte row per rank", "bytes": int(mlen),
                                              "ranks": len(mrows), "gather_ms": round(multi_gather_ms, 3)}},
        }
        if n_gpus == 1 and not args.no_sweep:
            result["size_sweep"] = size_sweep(container, algo, x, [10_000_000, 100_000_000, n])
        if not args.no_cpu and n_gpus == 1:                                              # (rank 0 at N = 1 only: at N > 1 the other ranks would sit in the final barrier meanwhile)
            result["cpu_baseline"] = cpu_baseline(host, chunk, args.cpu_sample, algo, gpu_payloads=payloads)
            result["host_api"] = host_api_rates(algo, host, chunk, args.host_sample)
        if n_gpus == 1 and not args.no_extra and algo == "chameleon" and args.data == "rep-text":
            # SURVEY.md 8d's hostile inputs at their stated size through the same device container path (never `value`): all-zero (every quad hits:
            # chameleon.rs:88-100), xorshift-random (the blow-up protection turns most blocks into raw copies: protection_state.rs:19-47) and a
            # patchwork of text / random / zeros / low-entropy runs that drives the FSM in and out — with per-direction roofline entries
            kinds = []
            for kind in ("zeros", "random", "mixed"):
                data = hostile_data(kind, 256 << 20)
                kinds.append(other_config(container, "chameleon", f"data kind '{kind}': 256 MiB, Chameleon container at the automatic chunk (SURVEY.md 8d 'also run')",
                                          data, steps=5, warmup=1, cpu_sample=16 << 20, settle_ms=args.settle_ms / 2))
                kinds[-1]["data_kind"] = kind
                del data
            result["data_kinds"] = kinds
            # ... and where the driver's `parsed.roofline` shows them: per kind the slower direction's fraction of the same roofline
            result["roofline"]["data_kinds"] = {k["data_kind"]: {"encode_frac": k["roofline"]["encode"]["frac"], "decode_frac": k["roofline"]["decode"]["frac"],
                                                                 "round_trip_MBps": k["value"], "container_form": k["container_form"].split(" ")[0]} for k in kinds}
            # BASELINE's other configurations in the same run (never `value`): configs 3 / 4 on the enwik8 stand-in, and the headline buffer as
            # ONE reference stream (the reference's own call shape)
            extra = [strict_stream_leg(host, x)]
            del cont, back
            # config 1's size (dickens: 10,192,446 B; absent, like every corpus: the stand-in benches/density.py uses): as a container at the
            # automatic chunk, and as ONE reference stream — the shape the reference's own bench has, with its whole-stream ratio
            small = datagen.prose(10_192_446, seed=0x9E3779B97F4A7C15)
            extra.append(other_config(container, "chameleon", "1: Chameleon on synth-prose-10M (dickens stand-in: 10,192,446 B of non-periodic synthetic prose), container at the automatic chunk",
                                      small, steps=10, warmup=2, cpu_sample=small.size, settle_ms=args.settle_ms / 2))
            xs = torch.from_numpy(small).cuda()
            extra.append(strict_stream_leg(small, xs, label="1 (strict): the same 10,192,446 B as ONE reference stream (chameleon_encode / chameleon_decode shape), device-resident"))
            del xs
            prose = datagen.prose(100_000_000, seed=0xD1B54A32D192ED03)
            for a, lbl in (("cheetah", "3: Cheetah on synth-prose-100M (enwik8 stand-in: 100,000,000 B of non-periodic synthetic prose)"),
                           ("lion", "4: Lion on synth-prose-100M (enwik8 stand-in: 100,000,000 B of non-periodic synthetic prose)")):
                extra.append(other_config(container, a, lbl, prose, settle_ms=args.settle_ms, all_cores=True))
            result["other_configs"] = extra
        print(json.dumps(result))
    if use_pg:
        dist.barrier()
        dist.destroy_process_group()


def density_ws(container, n, chunk, algo="chameleon"):
    from density_amd import _lib
    return max(_lib.lib().density_hip_encode_workspace_size(_lib.ALGO_IDS[algo], n, chunk), _lib.lib().density_hip_decode_workspace_size_for(_lib.ALGO_IDS[algo], n, chunk))


if __name__ == "__main__":
    main()
