import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import torch, datagen
from density_amd import container
n, chunk = 1 << 26, 1 << 20
host = datagen.rep_text(n)
x = torch.from_numpy(host).cuda()
back = torch.empty(n, dtype=torch.uint8, device="cuda")
cap = container.container_bound("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
hdr = container.encode_device("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk)
os.environ["DENSITY_HIP_DBG"] = "2048"
try:
    container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr)
    print("no early stop; equal:", bool(torch.equal(back, x)))
except Exception as ex:
    print("EARLY STOP detected:", str(ex)[:80], "equal:", bool(torch.equal(back, x)))
