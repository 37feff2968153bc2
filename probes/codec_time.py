"""Times container encode and decode kernels on rep-text for a list of DENSITY_HIP_DBG values (experiment driver)."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, datagen
from density_amd import container
n = int(os.environ.get("SIZE", 1 << 30))
chunk = int(os.environ.get("CHUNK", 1 << 20))
host = datagen.rep_text(n)
x = torch.from_numpy(host).cuda()
back = torch.empty(n, dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
cap = container.container_bound("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
os.environ["DENSITY_HIP_DBG"] = "0"
hdr = container.encode_device("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
container.set_profiling(True)
for dbg in (sys.argv[1:] or ["0"]):
    os.environ["DENSITY_HIP_DBG"] = dbg
    container.last_timings()
    for it in range(4):
        if os.environ.get("ENC", "1") == "1" and int(dbg) < 16:
            container.encode_device("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
        container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, stream=s, sync=False)
    torch.cuda.synchronize()
    tm = container.last_timings()
    enc = [ms for nm, ms in tm if nm == "chameleon_encode_chunks"]
    dec = [ms for nm, ms in tm if nm == "chameleon_decode_chunks"]
    print("DBG", dbg, "enc", ["%.3f" % v for v in enc[1:]], "dec", ["%.3f" % v for v in dec[1:]], flush=True)
