"""Prints the "algorithmic bytes vs measured HBM bytes" table of profiles/README.md for one round from the committed files
(profiles/rNN_pmc_summary.json, rNN_kernel_stats.csv, rNN_bench.json), so that the README's numbers cannot drift from them.
Usage: python probes/traffic_table.py r03"""
import csv, json, re, sys
tag = sys.argv[1]
pmc = json.load(open(f"profiles/{tag}_pmc_summary.json"))["kernels"]
bench = json.load(open(f"profiles/{tag}_bench.json"))
N = bench["config"]["bytes_per_gpu"]
E = bench["roofline"]["algorithmic_bytes_per_launch"] - N
avg = {}
for row in csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")):
    m = re.search(r"(chameleon_encode_rot|chameleon_decode_rot|compact_kernel)", row["Name"])
    if m and int(row["Calls"]) > avg.get(m.group(1), (0, 0))[1]:             # (several instances of a template: the one the timed steps launched)
        avg[m.group(1)] = (float(row["AverageNs"]) / 1e3, int(row["Calls"]))
rows = [("chameleon_encode_rot", "reads N, writes E into pages + directory + block index" if tag >= "r05" else "reads N, writes E into slots + block index", N + E),
        ("chameleon_decode_rot", "reads E + index, writes N", N + E),
        ("compact_kernel", "reads E, writes E; only in density_hip_pack_device" if tag >= "r03" else "reads E, writes E", 2 * E)]
print(f"N = {N:,} input bytes, E = {E:,} container bytes (ratio {N / E:.4f}); peak 8000 GB/s\n")
print("| kernel | algorithmic bytes | measured HBM bytes (corrected) | ratio | rocprof avg (launches) | algorithmic bytes / avg / peak |")
print("|---|---|---|---|---|---|")
for k, what, alg in rows:
    if k not in pmc:
        continue
    hbm = pmc[k]["hbm_bytes_corrected"]
    us, calls = avg.get(k, (None, 0))
    frac = f"{alg / (us * 1e-6) / 8e12:.3f}" if us else "-"
    print(f"| `{k}` ({what}) | {alg:,} | {hbm:,} | {hbm / alg:.3f} | {us:.1f} µs ({calls}) | {frac} |")
