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
for variant in (0, 2):
    container.set_kernel_variant(variant)
    hdr = container.encode_device("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk)
    hdr = container.encode_device("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk)
    container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr)
    print("variant", variant, "ok", bool(torch.equal(back, x)), flush=True)
