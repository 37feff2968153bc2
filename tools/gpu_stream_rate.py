"""Device-side rates of the reference-shaped stream entry points on ONE long stream (density_hip_stream_{encode,decode}_device), and
whether the results are the reference's stream / the input: python tools/gpu_stream_rate.py [MiB] [kind]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from density_amd import _lib, Chameleon
from oracle import pyoracle
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind = sys.argv[2] if len(sys.argv) > 2 else "rep"
n = mib << 20
data = datagen.by_kind(kind, n, seed=3)
lib = _lib.lib()
d_in = torch.from_numpy(data).cuda()
cap = Chameleon.safe_encode_buffer_size(n)
d_out = torch.empty(cap + 64, dtype=torch.uint8, device="cuda")
size = ctypes.c_size_t(0)
def enc():
    rc = lib.density_hip_stream_encode_device(0, ctypes.c_void_p(d_in.data_ptr()), n, ctypes.c_void_p(d_out.data_ptr()), d_out.numel(), None, ctypes.byref(size))
    assert rc == 0, rc
enc(); torch.cuda.synchronize()
t = time.time()
for _ in range(5): enc()
torch.cuda.synchronize(); dt = (time.time() - t) / 5
got = d_out[:size.value].cpu().numpy().tobytes()
want = pyoracle.encode("chameleon", data)
print(f"{kind} {mib} MiB: stream encode {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.2f} ms), {size.value} bytes, == reference stream: {got == want}")
d_enc = torch.from_numpy(np.frombuffer(want, dtype=np.uint8).copy()).cuda()
d_back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
back = ctypes.c_size_t(0)
def dec():
    rc = lib.density_hip_stream_decode_device(0, ctypes.c_void_p(d_enc.data_ptr()), len(want), ctypes.c_void_p(d_back.data_ptr()), n, None, ctypes.byref(back))
    assert rc == 0, rc
dec(); torch.cuda.synchronize()
t = time.time()
for _ in range(5): dec()
torch.cuda.synchronize(); dt = (time.time() - t) / 5
ok = back.value == n and bool(torch.equal(d_back[:n], d_in))
print(f"{kind} {mib} MiB: stream decode {n / dt / 1e9:.1f} GB/s ({dt * 1e3:.2f} ms), {back.value} bytes, == input: {ok}")
