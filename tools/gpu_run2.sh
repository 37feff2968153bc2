#!/bin/bash
mkdir -p gpurun_out/r2b
export TMPDIR=/tmp
B="timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu"
DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu > gpurun_out/r2b/prof.json 2> gpurun_out/r2b/prof.err; grep "density_hip prof" gpurun_out/r2b/prof.err | tail -36
for t in 0 2; do
  DENSITY_HIP_TUNE=$t $B > gpurun_out/r2b/bench_t$t.json 2>/dev/null; python - <<PY
import json; d=json.load(open("gpurun_out/r2b/bench_t$t.json")); print("tune $t", d["kernel_ms"], d["value"])
PY
done
for c in 262144 524288 2097152 4194304; do
  $B --chunk $c > gpurun_out/r2b/bench_c$c.json 2>/dev/null; python - <<PY
import json; d=json.load(open("gpurun_out/r2b/bench_c$c.json")); print("chunk $c", d["kernel_ms"], d["value"], d["compression_ratio"])
PY
done
