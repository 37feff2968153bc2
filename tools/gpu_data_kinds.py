"""Chameleon container on SURVEY.md 8d's data kinds (+ the headline text at the same size), device-resident, kernel times by HIP events, the first
chunks of every kind compared with the oracle's streams and decode == input:
    python tools/gpu_data_kinds.py [MiB=256] [kinds=text,zeros,random,mixed] [steps=5]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
import bench
from density_amd import container, _lib
if any(k.startswith("DENSITY_HIP_") for k in os.environ): _lib.use_debug_build()   # (switches are read by the debug build only)
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["text", "zeros", "random", "mixed"]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
for kind in kinds:
    host = datagen.rep_text(mib << 20) if kind == "text" else bench.hostile_data(kind, mib << 20)
    r = bench.other_config(container, "chameleon", kind, host, steps=steps, warmup=2, cpu_sample=8 << 20, settle_ms=50.0)
    print(f"{kind:>7}: encode {r['encode_ms']:.4f} ms  decode {r['decode_ms']:.4f} ms  round trip {r['value'] / 1e3:.1f} GB/s  ratio {r['compression_ratio']:.3f}  "
          f"frac enc {r['roofline']['encode']['frac']:.3f} dec {r['roofline']['decode']['frac']:.3f}  chunks == oracle: {r['cpu_baseline']['gpu_chunks_compared_bit_exact']}", flush=True)
