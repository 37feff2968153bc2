"""Diagnostic: the paged container at the headline size, several runs: which chunks decode wrong, their directories, and whether their CPU reassembly equals the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from density_amd import container, _lib
from oracle import pyoracle
n, chunk = 1 << 30, 4 << 20
host = datagen.rep_text(n)
x = torch.from_numpy(host).cuda()
cap = container.container_bound_paged("chameleon", n, chunk)
cont = torch.empty(cap, dtype=torch.uint8, device="cuda")
back = torch.zeros(n, dtype=torch.uint8, device="cuda")
ppc = int(_lib.lib().density_hip_paged_pages_per_chunk(chunk))
for run in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    cont.fill_(0xEE)
    hdr = container.encode_device_paged("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk)
    back.zero_(); torch.cuda.synchronize()
    got = container.decode_device(cont.data_ptr(), hdr.container_len, back.data_ptr(), n, header=hdr)
    diff = (back != x).view(-1, chunk).any(dim=1).nonzero().flatten().tolist()
    off = (32 + 4 * hdr.n_chunks + 15) // 16 * 16
    off = (off + (n + 255) // 256 + 15) // 16 * 16
    pages_base = (off + 16 * (ppc + 1) * hdr.n_chunks + 255) // 256 * 256
    npages = (hdr.container_len - pages_base) // 65536
    front = cont[:pages_base].cpu().numpy().tobytes()
    used_pages = sum(int.from_bytes(front[off + 16 * (ppc + 1) * i:off + 16 * (ppc + 1) * i + 4], "little") for i in range(hdr.n_chunks))
    print(f"run {run}: decoded {got}, container {hdr.container_len}, pages {npages}, in directories {used_pages}, chunks that differ: {diff[:10]} ({len(diff)})", flush=True)
    for c in diff[:2]:
        d = off + 16 * (ppc + 1) * c
        k = int.from_bytes(front[d:d + 4], "little")
        ent = [tuple(int.from_bytes(front[d + 16 * (j + 1) + 4 * f:d + 16 * (j + 1) + 4 * f + 4], "little") for f in range(4)) for j in range(k)]
        size = int.from_bytes(front[32 + 4 * c:36 + 4 * c], "little")
        print("  chunk", c, "pages", k, "size", size, "sum used", sum(e[2] for e in ent), "last entries", ent[-3:])
        bad = (back[c * chunk:(c + 1) * chunk] != x[c * chunk:(c + 1) * chunk]).nonzero().flatten()
        print("  first / last differing byte of the chunk:", int(bad[0]), int(bad[-1]), "count", bad.numel(), "-> blocks", int(bad[0]) // 256, int(bad[-1]) // 256)
        # the chunk's stream as a CPU reader sees it, against the oracle
        blob = cont[:hdr.container_len].cpu().numpy()
        parts = [bytes(blob[pages_base + e[0] * 65536:pages_base + e[0] * 65536 + e[2]]) for e in ent]
        stream = b"".join(parts)
        want = pyoracle.encode("chameleon", host[c * chunk:(c + 1) * chunk])
        print("  CPU reassembly == oracle stream:", stream == want, len(stream), len(want))
        if stream != want:
            w = np.frombuffer(want, dtype=np.uint8); g = np.frombuffer(stream, dtype=np.uint8)
            m = min(w.size, g.size); dd = np.nonzero(w[:m] != g[:m])[0]
            print("   first differing stream byte", int(dd[0]) if dd.size else None, "of", m, "; page boundaries at", np.cumsum([e[2] for e in ent]).tolist()[-4:])
        del blob
