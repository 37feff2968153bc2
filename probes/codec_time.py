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
# clock reference: the one-wavefront kernels on a quarter of the buffer (their time scales with the shader clock like the
# pipelined kernels'), so runs on different boxes / power states can be compared through the ratio
container.set_kernel_variant(1)
nq = n // 4
capq = container.container_bound("chameleon", nq, chunk)
hq = container.encode_device("chameleon", x.data_ptr(), nq, cont.data_ptr(), capq, chunk, stream=s)
container.last_timings()
for it in range(2):
    container.encode_device("chameleon", x.data_ptr(), nq, cont.data_ptr(), capq, chunk, stream=s, want_header=False)
    container.decode_device(cont.data_ptr(), hq.container_len, back.data_ptr(), nq, header=hq, stream=s, sync=False)
torch.cuda.synchronize()
tm = container.last_timings()
ref_e = min(ms for nm, ms in tm if nm == "chameleon_encode_chunks") * 4
ref_d = min(ms for nm, ms in tm if nm == "chameleon_decode_chunks") * 4
print("REF simple kernels (x4 extrapolated): enc %.3f dec %.3f" % (ref_e, ref_d), flush=True)
container.set_kernel_variant(0)
hdr = container.encode_device("chameleon", x.data_ptr(), n, cont.data_ptr(), cap, chunk, stream=s)
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
    print("DBG", dbg, "enc", ["%.3f" % v for v in enc[1:]], "dec", ["%.3f" % v for v in dec[1:]],
          "| vs ref: enc x%.2f dec x%.2f" % (ref_e / min(enc[1:]) if enc[1:] else 0, ref_d / min(dec[1:]) if dec[1:] else 0), flush=True)
