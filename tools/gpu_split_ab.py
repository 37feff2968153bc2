"""The two rotation encoders side by side on one box, same buffers (kernel variant 0 and 2048: whichever is not the default is the other one):
kernel times by HIP events on the headline workload (slotted and paged) and on SURVEY.md 8d's data kinds, the output of the one held against the
output of the other (slotted containers byte for byte, paged ones chunk stream by chunk stream) and decoded:    python tools/gpu_split_ab.py [steps=10]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
import bench
from density_amd import container
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
s = torch.cuda.current_stream().cuda_stream

def timed(fn, reps):
    for _ in range(20): fn()
    torch.cuda.synchronize(); container.set_profiling(True); container.last_timings()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    t = {}
    for nm, ms in container.last_timings(): t[nm] = t.get(nm, 0.0) + ms / reps
    container.set_profiling(False)
    return t

def run(label, host, chunk, forms, reps):
    n = host.size
    x = torch.from_numpy(host).cuda()
    cap = max(container.container_bound_paged("chameleon", n, chunk), container.container_bound_slotted("chameleon", n, chunk))
    conts = {v: torch.empty(cap, dtype=torch.uint8, device="cuda") for v in (0, 2048)}
    back = torch.empty(n, dtype=torch.uint8, device="cuda")
    encs = {"slotted": container.encode_device_slotted, "paged": container.encode_device_paged}
    for form in forms:
        enc = encs[form]
        hdrs, t = {}, {}
        for rep in range(int(os.environ.get('AB_REPS', 2))):
            for v in (0, 2048):
                container.set_kernel_variant(v)
                hdrs[v] = enc("chameleon", x.data_ptr(), n, conts[v].data_ptr(), cap, chunk, stream=s)
                t[v] = timed(lambda: enc("chameleon", x.data_ptr(), n, conts[v].data_ptr(), cap, chunk, stream=s, want_header=False), reps)
            container.set_kernel_variant(0)
            back.zero_(); torch.cuda.synchronize()      # (the library's kernels run on ITS stream when the caller's is the null stream: not ordered behind torch's)
            got = container.decode_device(conts[2048].data_ptr(), hdrs[2048].container_len, back.data_ptr(), n, header=hdrs[2048], stream=s)
            ok_dec = bool(got == n and torch.equal(back, x))
            if not ok_dec:
                L = int(hdrs[0].container_len)
                dd = (conts[0][:L] != conts[2048][:L]).nonzero().flatten()
                wrong = (back != x).nonzero().flatten()
                print(f"    decode returned {got} of {n}; {wrong.numel()} output bytes wrong, first {wrong[:4].cpu().tolist()}; container bytes differing from variant 0's: {dd.numel()}, first {dd[:6].cpu().tolist()}", flush=True)
                back.zero_()
                got2 = container.decode_device(conts[2048].data_ptr(), hdrs[2048].container_len, back.data_ptr(), n, header=hdrs[2048], stream=s)
                print(f"    decoded again: {got2}, == input: {bool(torch.equal(back, x))}", flush=True)
            if hdrs[0].flags & container.FLAG_PAGED:
                _, p0 = container.chunk_payloads(conts[0][:hdrs[0].container_len].cpu().numpy()); _, p1 = container.chunk_payloads(conts[2048][:hdrs[2048].container_len].cpu().numpy())
                same = p0 == p1
            else:
                _, p0 = container.chunk_payloads(conts[0][:hdrs[0].container_len].cpu().numpy()); _, p1 = container.chunk_payloads(conts[2048][:hdrs[2048].container_len].cpu().numpy())
                same = hdrs[0].container_len == hdrs[2048].container_len and p0 == p1
            e0, e1 = sum(t[0].values()), sum(t[2048].values())
            print(f"{label:>8} {form:>8}: variant 0 encode {e0:.4f} ms | variant 2048 encode {e1:.4f} ms ({e1 / e0 - 1:+.1%})   same streams: {same}   2048's container decodes to the input: {ok_dec}", flush=True)

n = 1 << 30
quick = "quick" in sys.argv
run("text1G", datagen.rep_text(n), 4 << 20, ("slotted",) if quick else ("slotted", "paged"), steps)
if quick: sys.exit(0)
for kind in ("zeros", "random", "mixed"):
    run(kind, bench.hostile_data(kind, 256 << 20), 1 << 20, ("slotted",), 5)
run("prose10M", datagen.prose(10_192_446, seed=1), 65536, ("slotted",), 10)
