"""Same-box A/B of experiment builds of a container DECODE (tools/build_variant.sh -> probes/variants/lib_*.so): config 3 / 4's container made by the tree's library,
decoded by every library named; decode checked.   python tools/gpu_decode_ab.py ALGO name ..."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container, _lib
algo = sys.argv[1]; sys.argv = sys.argv[:1] + sys.argv[2:]
n = int(os.environ.get('AB_BYTES', 100_000_000))
host = datagen.prose(n, seed=0xD1B54A32D192ED03)
x = torch.from_numpy(host).cuda()
tree = _lib.LIB_PATH
chunk = int(os.environ.get('AB_CHUNK', 0)) or int(_lib.lib().density_hip_auto_chunk_for(_lib.ALGO_IDS[algo], n))
cap = container.container_bound_slotted(algo, n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
hdr = container.encode_device_slotted(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
libs = [("tree", tree)] + [(nm, os.path.join(ROOT, "probes", "variants", f"lib_{nm}.so")) for nm in sys.argv[1:]]
handles = {}
def use(name, path):
    if name not in handles:
        L = ctypes.CDLL(path)
        for sym, (res, args) in _lib.SYMBOLS.items():
            if hasattr(L, sym):
                fn = getattr(L, sym); fn.restype, fn.argtypes = res, args
        handles[name] = L
    _lib._lib = handles[name]
for rep in range(2):
    for name, path in libs:
        use(name, path)
        back.zero_(); torch.cuda.synchronize()
        for _ in range(10): container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s, sync=False)
        torch.cuda.synchronize(); container.set_profiling(True); container.last_timings()
        for _ in range(8): container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s, sync=False)
        torch.cuda.synchronize()
        t = sum(ms for nm, ms in container.last_timings() if nm == "" + algo + "_decode_chunks") / 8
        container.set_profiling(False)
        print(f"{name:>12}: decode {t:.4f} ms  equal: {bool(torch.equal(back, x))}  differing bytes: {int((back != x).sum())}", flush=True)
