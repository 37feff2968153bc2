"""One reference stream of BASELINE config 1's size (10,192,446 B of prose) through the strict device entry points: wall time per call (tools; run it under
rocprofv3 --kernel-trace for the launches behind a call)."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from density_amd import _lib, Chameleon
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_192_446
data = datagen.prose(n, seed=0x9E3779B97F4A7C15)
lib = _lib.lib()
x = torch.from_numpy(data).cuda()
d_out = torch.empty(Chameleon.safe_encode_buffer_size(n) + 64, dtype=torch.uint8, device="cuda")
d_back = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
size, back = ctypes.c_size_t(0), ctypes.c_size_t(0)
enc = lambda: lib.density_hip_stream_encode_device(0, ctypes.c_void_p(x.data_ptr()), n, ctypes.c_void_p(d_out.data_ptr()), d_out.numel(), None, ctypes.byref(size))
dec = lambda: lib.density_hip_stream_decode_device(0, ctypes.c_void_p(d_out.data_ptr()), size.value, ctypes.c_void_p(d_back.data_ptr()), n, None, ctypes.byref(back))
for i in range(30): enc(); dec()
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); assert enc() == 0; torch.cuda.synchronize(); t1 = time.perf_counter()
    assert dec() == 0; torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"call {i}: encode {1e3*(t1-t0):.3f} ms, decode {1e3*(t2-t1):.3f} ms, E {size.value}, equal {bool(torch.equal(d_back[:n], x))}")
