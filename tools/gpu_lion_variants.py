"""Lion container decode time of experiment builds (probes/variants/lib_<name>.so; no correctness claim) against the tree's library, 100 MB of prose at the
automatic chunk:  python tools/gpu_lion_variants.py [steps] [name ...]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container, _lib
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
names = [a for a in sys.argv[2:] if not a.startswith("--") and not a.isdigit()]
algo = "lion" if "--cheetah" not in sys.argv else "cheetah"
tree = _lib.LIB_PATH
libs = [("tree", tree)] + [(n, os.path.join(ROOT, "probes", "variants", f"lib_{n}.so")) for n in names]
host = datagen.prose(100_000_000, seed=0xD1B54A32D192ED03)
n = host.size
x = torch.from_numpy(host).cuda()
handles = {}
def use(name, path):
    if name not in handles:
        L = ctypes.CDLL(path)
        for sym, (res, args) in _lib.SYMBOLS.items():
            if hasattr(L, sym):
                fn = getattr(L, sym); fn.restype, fn.argtypes = res, args
        handles[name] = L
    _lib._lib = handles[name]
use("tree", tree)
chunk = int(_lib.lib().density_hip_auto_chunk_for(_lib.ALGO_IDS[algo], n)) if "--chunk" not in sys.argv else int(sys.argv[sys.argv.index("--chunk") + 1])
cap = container.container_bound_slotted(algo, n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda"); scratch = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
hdr = container.encode_device_slotted(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
for name, path in libs:
    use(name, path)
    def timed(fn, key):
        fn(); torch.cuda.synchronize(); container.set_profiling(True); container.last_timings()
        for _ in range(steps): fn()
        torch.cuda.synchronize()
        t = sum(ms for nm, ms in container.last_timings() if nm == key) / steps
        container.set_profiling(False); return t
    e = timed(lambda: container.encode_device_slotted(algo, x.data_ptr(), n, scratch.data_ptr(), cap, chunk, stream=s, want_header=False), f"{algo}_encode_chunks")
    back.zero_()
    d = timed(lambda: container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s, sync=False), f"{algo}_decode_chunks")
    print(f"{name:>10}: {algo} chunk {chunk}: encode {e:.3f} ms  decode {d:.3f} ms  decode == input: {bool(torch.equal(back, x))}", flush=True)
