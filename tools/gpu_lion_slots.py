"""Lion config 4 (100 MB of prose, automatic chunk) decode and encode times by the number of streams IN FLIGHT (DENSITY_HIP_SERIAL_SLOTS, debug build): does
keeping the tables of the running streams inside the 256 MiB Infinity Cache (128 streams x 1.75 MiB) buy more per step than the idle CUs cost?
python tools/gpu_lion_slots.py [chunk]   """
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, datagen
from density_amd import container, _lib
_lib.use_debug_build()
algo = os.environ.get("ALGO", "lion")
n = 100_000_000
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 0
host = datagen.prose(n, seed=0xD1B54A32D192ED03)
x = torch.from_numpy(host).cuda()
cap = container.container_bound_slotted(algo, n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda"); back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for slots in [0, 64, 128, 192, 256, 384, 512]:
    if slots: os.environ["DENSITY_HIP_SERIAL_SLOTS"] = str(slots)
    else: os.environ.pop("DENSITY_HIP_SERIAL_SLOTS", None)
    hdr = container.encode_device_slotted(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
    def step():
        container.encode_device_slotted(algo, x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
        container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s, sync=False)
    for _ in range(2): step()
    torch.cuda.synchronize(); container.set_profiling(True); container.last_timings()
    for _ in range(4): step()
    torch.cuda.synchronize()
    t = {}
    for name, ms in container.last_timings(): t[name] = t.get(name, 0.0) + ms / 4
    container.set_profiling(False)
    ok = bool(torch.equal(back, x))
    e = sum(v for k, v in t.items() if "encode" in k); d = sum(v for k, v in t.items() if "decode" in k)
    print(f"{algo} chunk {hdr.chunk_size} ({hdr.n_chunks} chunks), streams in flight {slots or 'all'}: encode {e:.3f} ms decode {d:.3f} ms -> {n / (e + d) / 1e6:.1f} GB/s, equal {ok}", flush=True)
