"""Cycle account of the Lion pair decoder's steps (an experiment build with -DLION_PHASES: tools/build_variant.sh LP "-DLION_PHASES"): stream 7 of config 4's
container, the parser wave and the table wave.   python tools/gpu_lion_phases.py [name]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container, _lib
name = sys.argv[1] if len(sys.argv) > 1 else "LP"
algo = "lion"; n = 100_000_000
host = datagen.prose(n, seed=0xD1B54A32D192ED03)
x = torch.from_numpy(host).cuda()
chunk = int(_lib.lib().density_hip_auto_chunk_for(_lib.ALGO_IDS[algo], n))
cap = container.container_bound_slotted(algo, n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
hdr = container.encode_device_slotted(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
L = ctypes.CDLL(os.path.join(ROOT, "probes", "variants", f"lib_{name}.so"))
for sym, (res, args) in _lib.SYMBOLS.items():
    if hasattr(L, sym):
        fn = getattr(L, sym); fn.restype, fn.argtypes = res, args
_lib._lib = L
buf = (ctypes.c_ulonglong * 32)()
for _ in range(3): container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s)
L.density_debug_lion_phases(buf, 1)
R = 4
for _ in range(R): container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s)
L.density_debug_lion_phases(buf, 1)
v = list(buf); steps = v[9] or 1
print("equal:", bool(torch.equal(back, x)), " chunk", chunk, " steps per decode", steps / R)
names = ["PARSER: waiting for a free ring slot", "PARSER: parse + items + key match", "TABLES: waiting for a step", "chain of predicted runs", "row load (drained)",
         "key match + dictionary rounds", "rows' rounds", "repair", "stores"]
print("parser wave, cycles a step:")
for i in (0, 1): print(f"{names[i]:>48}: {v[i] / steps:9.0f}")
print(f"table wave, cycles a step (sum {sum(v[2:9]) / steps:.0f}):")
for i in range(2, 9): print(f"{names[i]:>48}: {v[i] / steps:9.0f}")
print(f"chain rounds a step {v[10] / steps:.2f}; rows' rounds a step {v[11] / steps:.2f}; steps repaired {v[12] / steps:.3f}; quads walked per repair {v[13] / max(v[12], 1):.1f}")
