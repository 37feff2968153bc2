"""Times only the container encode on rep-text (experiment driver; sweeps DENSITY_HIP_DBG values given in argv)."""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, datagen
from density_amd import container
n = int(os.environ.get("SIZE", 1 << 30))
host = datagen.rep_text(n)
x = torch.from_numpy(host).cuda()
s = torch.cuda.current_stream().cuda_stream
container.set_profiling(True)
for chunk in [int(c) for c in os.environ.get("CHUNKS", str(1 << 20)).split(",")]:
    cap = container.container_bound("chameleon", n, chunk)
    cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
    for dbg in (sys.argv[1:] or ["0"]):
        os.environ["DENSITY_HIP_DBG"] = dbg
        for it in range(5):
            container.encode_device("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s, want_header=False)
        torch.cuda.synchronize()
        tm = container.last_timings()
        enc = [ms for nm, ms in tm if nm == "chameleon_encode_chunks"]
        print("DBG", dbg, "chunk", chunk, "encode ms", ["%.3f" % v for v in enc], flush=True)
