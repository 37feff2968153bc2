"""Phase profile (debug build, DENSITY_HIP_PROF=1: work-group 0's cycle accounting and event counts, rotor.hip PhaseClock) of the rotation kernels on one
data kind:    python tools/gpu_phase_prof.py [text|zeros|random|mixed] [MiB] [chunk KiB] [kernel variant]"""
import os, sys
os.environ.setdefault("DENSITY_HIP_PROF", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen, bench
from density_amd import container, _lib
_lib.use_debug_build()
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
mib = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if kind == "text" else 256)
chunk = (int(sys.argv[3]) << 10) if len(sys.argv) > 3 else ((4 << 20) if mib >= 1024 else (1 << 20))
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n = mib << 20
host = datagen.rep_text(n) if kind == "text" else bench.hostile_data(kind, n)
x = torch.from_numpy(host).cuda()
cap = container.container_bound_slotted("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
container.set_kernel_variant(variant)
print(f"== {kind}, {mib} MiB, chunks of {chunk >> 10} KiB, kernel variant {variant}", file=sys.stderr, flush=True)
for _ in range(2):
    hdr = container.encode_device_slotted("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
    assert container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s) == n
