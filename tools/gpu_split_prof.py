"""Phase profile (debug build, DENSITY_HIP_PROF=1: work-group 0's cycle accounting, rotor.hip PhaseClock) of the two rotation encoders on the headline workload:
    DENSITY_HIP_PROF=1 python tools/gpu_split_prof.py"""
import os, sys
os.environ.setdefault("DENSITY_HIP_PROF", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container, _lib
_lib.use_debug_build()
n, chunk = 1 << 30, 4 << 20
x = torch.from_numpy(datagen.rep_text(n)).cuda()
cap = container.container_bound_slotted("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for v in (0, 2048):
    container.set_kernel_variant(v)
    print(f"== kernel variant {v}", file=sys.stderr, flush=True)
    for _ in range(3):
        container.encode_device_slotted("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
