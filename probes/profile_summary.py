"""Turns the output of probes/profile_round.sh into profiles/rNN_{kernel_stats,pmc_fetch,pmc_write}.csv, rNN_pmc_summary.json,
rNN_lds_counters.txt and rNN_phase_profile.txt."""
import csv, json, collections, os, re, shutil, sys, glob
src, tag = sys.argv[1], sys.argv[2]          # e.g. gpurun_out/prof r02
def one(pattern):
    m = glob.glob(pattern, recursive=True)   # (gpurun merges a call's files into what earlier calls left: the newest is this call's)
    assert m, pattern
    return max(m, key=os.path.getmtime)
KERNEL = r"(chameleon_encode_rot|chameleon_decode_rot|chameleon_\w+_chunks_pipe|compact_kernel|layout_\w+_kernel|selftest_kernel|rotor_selftest_kernel|read4|read16|write4|write2)"
def avg(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        m = re.search(KERNEL, row["Kernel_Name"])
        if m:
            acc[m.group(1)].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
stats = one(f"{src}/stats/**/*_kernel_stats.csv")
fetch = one(f"{src}/fetch/**/*_counter_collection.csv")
write = one(f"{src}/write/**/*_counter_collection.csv")
cf = avg(one(f"{src}/calib_fetch/**/*_counter_collection.csv"), "FETCH_SIZE")
cw = avg(one(f"{src}/calib_write/**/*_counter_collection.csv"), "WRITE_SIZE")
GiB_KB = float(1 << 20)
calib = {"read4_KB_per_GiB": cf.get("read4"), "read16_KB_per_GiB": cf.get("read16"), "write4_KB_per_GiB": cw.get("write4"), "write2_KB_per_GiB": cw.get("write2")}
# correction factors = true bytes / counted bytes for each access width (counters are in KB)
k_r4 = GiB_KB / cf["read4"]; k_r16 = GiB_KB / cf["read16"]; k_w4 = GiB_KB / cw["write4"]; k_w2 = GiB_KB / cw["write2"]
f, w = avg(fetch, "FETCH_SIZE"), avg(write, "WRITE_SIZE")
out = {"unit": "bytes per launch",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --no-cpu",
       "calibration": {"raw_KB_for_1GiB": calib, "factor_read4": round(k_r4, 4), "factor_read16": round(k_r16, 4), "factor_write4": round(k_w4, 4), "factor_write2": round(k_w2, 4),
                       "note": "probes/fetch_calib.hip streams exactly 1 GiB per kernel at each access width; factor = true bytes / counter bytes"},
       "workload": "bench.py default: chameleon rep-text 1 GiB, automatic chunk (4 MiB), PAGED container with block index (round 5; rounds 3-4: slotted)", "kernels": {}}
# the kernel generation these counters belong to: bench.py quotes them as `roofline.traffic` only for a library of the same kernels id
try:
    out["kernels_id"] = json.load(open(f"{src}/bench.json")).get("kernels_id")
except Exception:
    out["kernels_id"] = None
for k in sorted(set(f) | set(w)):
    fk, wk = f.get(k, 0.0), w.get(k, 0.0)
    rot = k.startswith("chameleon_") and k.endswith("_rot")
    # reads: the rotation kernels load 4 B/lane (encoder) or 4-8 B/lane at item granularity (decoder); compact / pipelines 16 B/lane
    kr = k_r4 if rot else k_r16
    kw = k_w4
    out["kernels"][k] = {"fetch_KB_raw": round(fk, 3), "write_KB_raw": round(wk, 3), "hbm_bytes_corrected": int(round((kr * fk + kw * wk) * 1024)),
                         "default_path": bool(rot or k in ("compact_kernel",))}
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
def filt(s, d):
    rows = list(csv.reader(open(s)))
    kn = rows[0].index("Kernel_Name")
    csv.writer(open(d, "w")).writerows([rows[0]] + [r for r in rows[1:] if "density::" in r[kn]])
filt(fetch, f"profiles/{tag}_pmc_fetch.csv")
filt(write, f"profiles/{tag}_pmc_write.csv")
# LDS counters: per-kernel averages
lds = one(f"{src}/lds/**/*_counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(lds)):
    m = re.search(KERNEL, row["Kernel_Name"])
    if m:
        acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(f"profiles/{tag}_lds_counters.txt", "w") as fh:
    fh.write("rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu\n")
    fh.write("per-launch averages (SQ_* summed over the chip; GRBM_GUI_ACTIVE summed over the 8 XCDs: the kernel ran GRBM_GUI_ACTIVE / 8 cycles)\n")
    fh.write("calibration (profiles/r01_lds_counters.txt): one 64-lane ds_mskor_rtn_b32 on random slots = 10.1 SQ_LDS_IDX_ACTIVE, 10.05 ATOMIC_RETURN, 6.05 BANK_CONFLICT\n")
    for k, cs in sorted(acc.items()):
        vals = {c: sum(v) / len(v) for c, v in cs.items()}
        fh.write(f"{k}: " + ", ".join(f"{c} {v:.4g}" for c, v in sorted(vals.items())))
        if "SQ_LDS_IDX_ACTIVE" in vals and vals.get("GRBM_GUI_ACTIVE"):
            cyc = vals['GRBM_GUI_ACTIVE'] / 8
            fh.write(f"  -> per CU: LDS array busy {vals['SQ_LDS_IDX_ACTIVE'] / 256 / cyc:.1%} of the kernel's {cyc:.4g} cycles")
        fh.write("\n")
shutil.copy(f"{src}/phase_profile.txt", f"profiles/{tag}_phase_profile.txt")
import os
if os.path.exists(f"{src}/ta_sq.txt"):
    shutil.copy(f"{src}/ta_sq.txt", f"profiles/{tag}_ta_sq_counters.txt")
if os.path.exists(f"{src}/vmem_width.txt"):
    shutil.copy(f"{src}/vmem_width.txt", f"profiles/{tag}_probe_vmem_width.log")
for extra in ("packed", "slotted", "cheetah", "lion"):
    m = glob.glob(f"{src}/stats_{extra}/**/*_kernel_stats.csv", recursive=True)
    if m:
        shutil.copy(max(m, key=os.path.getmtime), f"profiles/{tag}_{extra}_kernel_stats.csv")   # (the newest: gpurun merges a call into what earlier calls left)
shutil.copy(f"{src}/bench.json", f"profiles/{tag}_bench.json")
for row in csv.DictReader(open(stats)):
    if "density::" in row["Name"]:
        print("%-44s calls %3s avg %10.1f us" % (re.sub(r"\(.*", "", row["Name"])[-44:], row["Calls"], float(row["AverageNs"]) / 1e3))
print(json.dumps(out["calibration"]))
print(json.dumps({k: v["hbm_bytes_corrected"] for k, v in out["kernels"].items()}))
