"""How long a head the exchange passes need (exchange_stages.hip StageGeo::kHeadBytes / kHeadCalm): per data kind and chunk size, the chunks through the passes, the chunks
handed back to the in-order kernel (kernel variant bit 64) and the encode time, for the tree's library and experiment builds.   python tools/gpu_head_audit.py ALGO name ..."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
import test_gpu_decode_passes as tp
from density_amd import container, _lib
algo = sys.argv[1]
libs = [("tree", _lib.LIB_PATH)] + [(nm, os.path.join(ROOT, "probes", "variants", f"lib_{nm}.so")) for nm in sys.argv[2:]]
handles = {}
def use(name, path):
    if name not in handles:
        L = ctypes.CDLL(path)
        for sym, (res, args) in _lib.SYMBOLS.items():
            if hasattr(L, sym):
                fn = getattr(L, sym); fn.restype, fn.argtypes = res, args
        handles[name] = L
    _lib._lib = handles[name]
n = 64 << 20
s = torch.cuda.current_stream().cuda_stream
for kind in ["prose", "rep", "mixed", "binaryish", "patchy", "pairs", "vocab30", "random", "zeros"]:
    host = datagen.rep_text(n) if kind == "rep" else tp.make(kind, n, 11)
    x = torch.from_numpy(host).cuda()
    for chunk in (256 << 10, 384 << 10, 1 << 20):
        cap = container.container_bound_slotted(algo, n, chunk)
        cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
        row = []
        for name, path in libs:
            use(name, path)
            container.set_kernel_variant(64)
            a = (ctypes.c_uint64 * 2)(); _lib.lib().density_hip_stage_stats(a); before = (a[0], a[1])
            container.encode_device_slotted(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
            torch.cuda.synchronize(); _lib.lib().density_hip_stage_stats(a)
            container.set_kernel_variant(0)
            for _ in range(3): container.encode_device_slotted(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): container.encode_device_slotted(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
            torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
            row.append(f"{name}: {a[0] - before[0]} / {a[1] - before[1]} back, {ms:.3f} ms")
        print(f"{kind:>10} chunks of {chunk >> 10:4d} KiB   " + "   ".join(row), flush=True)
