#!/bin/bash
# usage: tools/gpu_bench_full.sh <tag> [bench args] — the default bench line (what the driver runs), pretty-printed in parts
TAG=${1:-x}; shift; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
( time timeout 900 python bench.py "$@" > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err ) 2>&1 | grep real
python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$TAG/bench.json"))
    print("value", d["value"], "ms/step", d["ms_per_step"], "kernel_ms", d["kernel_ms"], "roofline", d["roofline"]["frac"], "whole", d["whole_path_hbm_frac"], d["container_form"][:20])
    print("packed", d.get("packed_container"))
    print("cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if k!="sample"})
    print("host_api", d.get("host_api"))
    for o in d.get("other_configs",[]): print("other:", o["config"][:40], o.get("value"), "enc", o.get("encode_ms"), "dec", o.get("decode_ms"), "ratio", o.get("compression_ratio"), o.get("kernel_ms"), "cpu", (o.get("cpu_baseline") or {}).get("value"))
    print("sweep", d.get("size_sweep"))
except Exception as ex: print("bench failed", ex, open("gpurun_out/$TAG/bench.err").read()[-1500:])
PY
