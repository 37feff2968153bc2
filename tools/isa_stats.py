"""Registers, spills and instruction counts of the shipped kernels (from the built library): python tools/isa_stats.py [name-substring] [--dump DIR]"""
import re, sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from check_isa import shipped_code
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "density_amd", "libdensity_hip.so")
want = sys.argv[1] if len(sys.argv) > 1 else "_rot"
funcs, notes = shipped_code(lib)
meta = {}
for blk in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk)
    if name:
        g = lambda k: re.search(rf"\.{k}:\s+(\d+)", blk)
        meta[name.group(1)] = {k: int(g(k).group(1)) for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size") if g(k)}
for name, body in funcs.items():
    if want not in name:
        continue
    kinds = collections.Counter()
    for t in body:
        op = t.split()[0]
        kinds["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"] += 1
    print(name[:120], len(body), dict(kinds), meta.get(name))
    if "--dump" in sys.argv:
        d = sys.argv[sys.argv.index("--dump") + 1]; os.makedirs(d, exist_ok=True)
        open(os.path.join(d, re.sub(r"\W", "_", name)[:100] + ".s"), "w").write("\n".join(body))
