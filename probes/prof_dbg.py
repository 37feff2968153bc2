import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, datagen
from density_amd import container
n, chunk = 1 << 28, 1 << 20
host = datagen.rep_text(n)
x = torch.from_numpy(host).cuda()
back = torch.empty(n, dtype=torch.uint8, device="cuda")
cap = container.container_bound("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
os.environ.pop("DENSITY_HIP_PROF", None)
hdr = container.encode_device("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk)
os.environ["DENSITY_HIP_PROF"] = "1"
for dbg in sys.argv[1:]:
    os.environ["DENSITY_HIP_DBG"] = dbg
    print("DBG", dbg, flush=True)
    try:
        container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr)
    except Exception as ex:
        print("  (decode reported:", str(ex)[:60], ")")
