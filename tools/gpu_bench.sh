#!/bin/bash
# usage: TUNES="0 4 ..." tools/gpu_bench.sh <tag>  — bench only, per tune value
TAG=${1:-x}; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
for t in $TUNES; do
  DENSITY_HIP_TUNE=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/$TAG/bench_t$t.json 2>gpurun_out/$TAG/bench_t$t.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$TAG/bench_t$t.json")); print("tune $t", d["kernel_ms"], d["value"], d["roofline"]["frac"])
except Exception as ex: print("tune $t bench failed", ex, open("gpurun_out/$TAG/bench_t$t.err").read()[-400:])
PY
done
