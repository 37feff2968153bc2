"""The three forms of the headline container on one box, same buffers: slotted (no stitch, worst-case slots), paged (no stitch, wire-ready) and packed
(encode + stitch), kernel times by HIP events, bytes of each form:   python tools/gpu_forms.py [steps=10]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n, chunk = 1 << 30, 4 << 20
x = torch.from_numpy(datagen.rep_text(n)).cuda()
cap = max(container.container_bound_paged("chameleon", n, chunk), container.container_bound_slotted("chameleon", n, chunk))
cont = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
forms = {"slotted": container.encode_device_slotted, "paged": container.encode_device_paged, "packed": container.encode_device}
def timed(fn):
    for _ in range(30): fn()
    torch.cuda.synchronize(); container.set_profiling(True); container.last_timings()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    t = {}
    for nm, ms in container.last_timings(): t[nm] = t.get(nm, 0.0) + ms / steps
    container.set_profiling(False)
    return t
for rep in range(2):
    for name, enc in forms.items():
        hdr = enc("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
        te = timed(lambda: enc("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False))
        back.zero_(); torch.cuda.synchronize()
        td = timed(lambda: container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s, sync=False))
        e, d = sum(te.values()), sum(td.values())
        print(f"{name:>8}: encode {e:.4f} ms ({', '.join(f'{k} {v:.4f}' for k, v in te.items())})  decode {d:.4f} ms  round trip {n / (e + d) / 1e6:.1f} GB/s  "
              f"container {hdr.container_len} B (N / that = {n / hdr.container_len:.4f})  == input: {bool(torch.equal(back, x))}", flush=True)
