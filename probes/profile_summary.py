"""Turns the output of probes/profile_round.sh into profiles/rNN_{kernel_stats,pmc_fetch,pmc_write}.csv and rNN_pmc_summary.json."""
import csv, json, collections, re, shutil, sys, glob
src, tag = sys.argv[1], sys.argv[2]          # e.g. gpurun_out/prof r01
def one(pattern):
    m = glob.glob(pattern, recursive=True)
    assert len(m) == 1, (pattern, m)
    return m[0]
stats = one(f"{src}/stats/**/*_kernel_stats.csv")
fetch = one(f"{src}/fetch/**/*_counter_collection.csv")
write = one(f"{src}/write/**/*_counter_collection.csv")
def avg(path, counter):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        m = re.search(r"(chameleon_\w+_chunks_pipe|compact_kernel|layout_\w+_kernel|selftest_kernel)", row["Kernel_Name"])
        if m:
            acc[m.group(1)].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
f, w = avg(fetch, "FETCH_SIZE"), avg(write, "WRITE_SIZE")
out = {"unit": "bytes per launch",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 2 --warmup 1 --no-cpu",
       "correction": "FETCH_SIZE x 2 (MI355X_MICROARCH.md HBM: gfx950 tallies the 128-B requests of 16 B/lane streams at 64 B), WRITE_SIZE x 1; both counters are in KB",
       "workload": "bench.py default: chameleon rep-text 1 GiB, chunk 1 MiB, container with block index", "kernels": {}}
for k in sorted(set(f) | set(w)):
    fk, wk = f.get(k, 0.0), w.get(k, 0.0)
    out["kernels"][k] = {"fetch_KB_raw": round(fk, 3), "write_KB_raw": round(wk, 3), "hbm_bytes_corrected": int(round((2 * fk + wk) * 1024))}
json.dump(out, open(f"profiles/{tag}_pmc_summary.json", "w"), indent=1)
shutil.copy(stats, f"profiles/{tag}_kernel_stats.csv")
def filt(s, d):
    rows = list(csv.reader(open(s)))
    csv.writer(open(d, "w")).writerows([rows[0]] + [r for r in rows[1:] if "density::" in r[8]])
filt(fetch, f"profiles/{tag}_pmc_fetch.csv")
filt(write, f"profiles/{tag}_pmc_write.csv")
for row in csv.DictReader(open(stats)):
    if "density::" in row["Name"]:
        print("%-40s calls %3s avg %10.1f us" % (re.sub(r"\(.*", "", row["Name"])[-40:], row["Calls"], float(row["AverageNs"]) / 1e3))
print(json.dumps({k: v["hbm_bytes_corrected"] for k, v in out["kernels"].items()}))
