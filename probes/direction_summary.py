"""Turns probes/profile_directions.sh's passes into profiles/rNN_pmc_directions.json: HBM bytes per encode call and per decode call of configs 3 / 4
(Cheetah / Lion, 100 MB of prose at the automatic chunk), all kernels and fills of a call summed.  A dispatch belongs to the direction its kernel's name
says; fills and copies (table clears, scratch clears) to the direction of the next codec kernel behind them.  FETCH_SIZE counts half the bytes read on gfx950
(profiles/rNN_pmc_summary.json: the calibration of the same round), WRITE_SIZE all of the bytes written; counters are in KB.
usage: python probes/direction_summary.py gpurun_out/prof5d r05"""
import csv, glob, json, os, re, sys
src, tag = sys.argv[1], sys.argv[2]
cal = json.load(open(f"profiles/{tag}_pmc_summary.json"))
kf, kw = cal["calibration"]["factor_read4"], cal["calibration"]["factor_write4"]
ENC = re.compile(r"encode_wave|exchange_stage|stage_emit_records|stage_record_layout|layout_encode|compact_kernel|encode_lane|scan_offsets|compact_bytes")
DEC = re.compile(r"decode_wave|decode_pair|cheetah_parse|cheetah_prepare|cheetah_pass|cheetah_walk|cheetah_finish|layout_decode|decode_lane")
ENC_CALL = re.compile(r"layout_encode_kernel|layout_encode_batch_kernel")
DEC_CALL = re.compile(r"cheetah_finish|lion_decode_wave|lion_decode_pair|cheetah_decode_wave")
out = {"unit": "bytes per call (all kernels and fills of the direction)", "kernels_id": cal.get("kernels_id"),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --algo A --data prose --size 100000000 --settle-ms 0 --steps 2 --warmup 1 --no-cpu --no-sweep --no-extra",
       "factor_fetch": kf, "factor_write": kw, "configs": {}}
for algo in ("cheetah", "lion"):
    per = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        m = glob.glob(f"{src}/{algo}_{ctr}/**/*_counter_collection.csv", recursive=True)
        assert m, (algo, ctr)
        rows = [r for r in csv.DictReader(open(max(m, key=os.path.getmtime))) if r["Counter_Name"] == ctr]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        kinds = ["enc" if ENC.search(r["Kernel_Name"]) else "dec" if DEC.search(r["Kernel_Name"]) else ("fill" if "rocclr" in r["Kernel_Name"] else "other") for r in rows]
        nxt = None
        for i in range(len(rows) - 1, -1, -1):                       # fills take the direction of the next codec kernel
            if kinds[i] in ("enc", "dec"): nxt = kinds[i]
            elif kinds[i] == "fill" and nxt: kinds[i] = nxt
        tot = {"enc": 0.0, "dec": 0.0}
        for r, k in zip(rows, kinds):
            if k in tot: tot[k] += float(r["Counter_Value"])
        calls = {"enc": sum(1 for r in rows if ENC_CALL.search(r["Kernel_Name"])), "dec": sum(1 for r in rows if DEC_CALL.search(r["Kernel_Name"]))}
        per[ctr] = {k: (tot[k] / calls[k] if calls[k] else None, calls[k]) for k in tot}
    cfg = {}
    for k, name in (("enc", "encode"), ("dec", "decode")):
        f, w = per["FETCH_SIZE"][k][0], per["WRITE_SIZE"][k][0]
        cfg[name] = {"fetch_KB_raw": round(f, 1), "write_KB_raw": round(w, 1), "calls_seen": [per["FETCH_SIZE"][k][1], per["WRITE_SIZE"][k][1]],
                     "hbm_bytes_corrected": int(round((kf * f + kw * w) * 1024))}
    out["configs"][algo] = cfg
json.dump(out, open(f"profiles/{tag}_pmc_directions.json", "w"), indent=1)
print(json.dumps(out["configs"], indent=1))
