import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
import numpy as np, torch, datagen
from density_amd import container
for kind, chunk, n in (("mixed", 4096, 40*4096+77), ("prose", 4096, 40*4096+77), ("mixed", 256, 40*256+77), ("mixed", 65536, 3*(1<<20)+12345)):
    data = datagen.by_kind(kind, n, seed=chunk)
    x = torch.from_numpy(data).cuda()
    cap = container.container_bound("chameleon", n, chunk)
    cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
    back = torch.zeros(n, dtype=torch.uint8, device="cuda")
    hdr = container.encode_device("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk)
    try:
        container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr, sync=False)
    except Exception as ex:
        print("launch err", ex)
    torch.cuda.synchronize()
    b = back.cpu().numpy()
    bad = np.nonzero(b != data)[0]
    idx = container.block_index(cont[:hdr.container_len].cpu().numpy())
    print(kind, chunk, "bad bytes", bad.size, "first", bad[:3], "chunks with errors", sorted(set((bad // chunk).tolist()))[:10])
    if bad.size:
        c = int(bad[0] // chunk)
        print("  chunk", c, "index entries", list(idx[c*(chunk//256):(c+1)*(chunk//256)]), "first bad offset in chunk", int(bad[0] % chunk))
