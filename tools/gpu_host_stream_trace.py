"""Trace of the pipelined reference symbols: the truncated-stream cases one by one, then timings of a 64 MiB round trip."""
import faulthandler, sys, time
import numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import datagen
from density_amd import Chameleon, container
from density_amd.codec import DecodeError
from oracle import pyoracle
faulthandler.enable()

data = datagen.rep_text(64 << 20, period=1_000_003)
out = np.zeros(Chameleon.safe_encode_buffer_size(data.size), dtype=np.uint8)
back = np.zeros(data.size, dtype=np.uint8)
for i in range(3):
    t0 = time.perf_counter(); m = Chameleon.encode(data, out); t1 = time.perf_counter(); k = Chameleon.decode(out[:m], back); t2 = time.perf_counter()
    print(f"round {i}: encode {data.size / (t1 - t0) / 1e9:.1f} GB/s, decode {data.size / (t2 - t1) / 1e9:.1f} GB/s", flush=True)
assert k == data.size and np.array_equal(back, data)
print("staged:", flush=True)
container.set_kernel_variant(512)
for i in range(2):
    t0 = time.perf_counter(); m = Chameleon.encode(data, out); t1 = time.perf_counter(); k = Chameleon.decode(out[:m], back); t2 = time.perf_counter()
    print(f"round {i}: encode {data.size / (t1 - t0) / 1e9:.1f} GB/s, decode {data.size / (t2 - t1) / 1e9:.1f} GB/s", flush=True)
container.set_kernel_variant(0)

data = datagen.rep_text(48 << 20, period=1_000_003)
enc = np.frombuffer(pyoracle.encode("chameleon", data), dtype=np.uint8)
outb = np.zeros(data.size, dtype=np.uint8)
for cut in (3, 1000, enc.size // 2 + 1):
    for variant in (0, 4):
        print("cut", cut, "variant", variant, flush=True)
        container.set_kernel_variant(variant)
        try:
            m = Chameleon.decode(enc[:-cut].copy(), outb)
            print("  ok", m, flush=True)
        except DecodeError as ex:
            print("  error", ex, flush=True)
container.set_kernel_variant(0)
