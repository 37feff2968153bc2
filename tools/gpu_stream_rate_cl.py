"""Device-side rate of `cheetah_encode` / `lion_encode` on ONE long stream (density_hip_stream_encode_device: the exchange passes with the
whole stream as one chunk), and whether the result is the reference's stream: python tools/gpu_stream_rate_cl.py [MiB] [kind]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from density_amd import _lib, BY_NAME
from oracle import pyoracle
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 16
kind = sys.argv[2] if len(sys.argv) > 2 else "prose"
n = (mib << 20) + 1003
data = datagen.by_kind(kind, n, seed=3)
lib = _lib.lib()
d_in = torch.from_numpy(data).cuda()
for algo, code in (("cheetah", 1), ("lion", 2)):
    cap = BY_NAME[algo].safe_encode_buffer_size(n)
    d_out = torch.empty(cap + 64, dtype=torch.uint8, device="cuda")
    size = ctypes.c_size_t(0)
    def enc():
        rc = lib.density_hip_stream_encode_device(code, ctypes.c_void_p(d_in.data_ptr()), n, ctypes.c_void_p(d_out.data_ptr()), d_out.numel(), None, ctypes.byref(size))
        assert rc == 0, rc
    enc(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(3): enc()
    torch.cuda.synchronize(); dt = (time.time() - t) / 3
    got = d_out[:size.value].cpu().numpy().tobytes()
    print(f"{algo} {kind} {n} bytes: stream encode {n / dt / 1e9:.2f} GB/s ({dt * 1e3:.2f} ms), {size.value} bytes, == reference stream: {got == pyoracle.encode(algo, data)}")
