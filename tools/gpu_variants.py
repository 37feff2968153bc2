"""Same-box, same-buffers A/B of experiment builds (tools/build_variant.sh -> probes/variants/lib_*.so) against the tree's library: kernel times of
the headline workload's ENCODE and DECODE separately, HIP events, no correctness claim for the variants (the decode leg of every library
reads the container the tree's library made and is checked; experiment encoders may write garbage — their output goes to a scratch buffer).
    python tools/gpu_variants.py [steps] [name ...]        (names: probes/variants/lib_<name>.so; DENSITY_HIP_TUNE applies to all of them)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container, _lib
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tree_path = _lib.LIB_PATH
libs = [("tree", tree_path)] + [(n, os.path.join(ROOT, "probes", "variants", f"lib_{n}.so")) for n in sys.argv[2:]]
n = int(os.environ.get("DENSITY_AB_BYTES", 1 << 30))                       # (DENSITY_AB_BYTES: another size at its automatic chunk, e.g. 10000000)
chunk = (4 << 20) if n == 1 << 30 else int(_lib.lib().density_hip_auto_chunk_for(0, n))
x = torch.from_numpy(datagen.rep_text(n)).cuda()
cap = container.container_bound_slotted("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda"); scratch = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
handles = {}
def use(name, path):
    if name not in handles:
        import ctypes
        L = ctypes.CDLL(path)                                    # (older builds lack newer symbols: bind what is there)
        for sym, (res, args) in _lib.SYMBOLS.items():
            if hasattr(L, sym):
                fn = getattr(L, sym); fn.restype, fn.argtypes = res, args
        handles[name] = L
    _lib._lib = handles[name]
use("tree", tree_path)
hdr = container.encode_device_slotted("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
def timed(fn, key):
    for _ in range(30): fn()                                     # (the clock ramp: tools/gpu_ramp.py)
    torch.cuda.synchronize(); container.set_profiling(True); container.last_timings()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    t = sum(ms for nm, ms in container.last_timings() if nm == key) / steps
    container.set_profiling(False)
    return t
for rep in range(3):
    for name, path in libs:
        use(name, path)
        e = timed(lambda: container.encode_device_slotted("chameleon", x.data_ptr(), n, scratch.data_ptr(), cap, chunk, stream=s, want_header=False), "chameleon_encode_chunks")
        back.zero_()
        d = timed(lambda: container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s, sync=False), "chameleon_decode_chunks")
        print(f"{name:>12}: encode {e:.4f} ms   decode {d:.4f} ms   decode == input: {bool(torch.equal(back, x))}", flush=True)
