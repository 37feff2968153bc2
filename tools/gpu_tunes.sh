#!/bin/bash
# usage: tools/gpu_tunes.sh <tag> "<tune values>" [extra bench args]   — one bench line per DENSITY_HIP_TUNE value (kernel times, value, roofline)
TAG=${1:-x}; TUNES=${2:-0}; shift; shift
mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
for t in $TUNES; do
  DENSITY_HIP_TUNE=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-sweep --no-extra "$@" > gpurun_out/$TAG/bench_t$t.json 2>gpurun_out/$TAG/bench_t$t.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$TAG/bench_t$t.json")); print("tune $t", d["kernel_ms"], d["value"], d["roofline"]["frac"], d["whole_path_hbm_frac"])
except Exception as ex: print("tune $t bench failed", ex, open("gpurun_out/$TAG/bench_t$t.err").read()[-400:])
PY
done
