"""How the exchange passes (exchange_stages.hip) served a container encode: chunks through the passes, chunks handed back to the in-order
kernel (kernel variant bit 64).  usage: python tools/gpu_stage_audit.py [algo] [bytes] [chunk]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import datagen
from density_amd import _lib, container

algo = sys.argv[1] if len(sys.argv) > 1 else "lion"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
data = datagen.prose(n, seed=0xD1B54A32D192ED03)
cont = np.zeros(container.container_bound(algo, n, chunk), dtype=np.uint8)
container.set_kernel_variant(64)
a = (ctypes.c_uint64 * 2)()
container.encode(algo, data, cont, chunk)
_lib.lib().density_hip_stage_stats(a)
print(f"{algo} {n} bytes, chunks of {chunk}: {a[0]} chunks through the passes, {a[1]} handed back")
