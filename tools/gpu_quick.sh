#!/bin/bash
# usage: tools/gpu_quick.sh <tag> — no full suite: rotor tests for $TEST_TUNES, profile for $PROF_TUNES, bench for $BENCH_TUNES, extra commands in $EXTRA
TAG=${1:-x}; shift
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
for t in $TEST_TUNES; do
  DENSITY_HIP_TUNE=$t timeout 900 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "rotor" > gpurun_out/$TAG/pytest_t$t.log 2>&1; echo "pytest tune $t rc=$?"; tail -2 gpurun_out/$TAG/pytest_t$t.log
done
for t in $PROF_TUNES; do
  DENSITY_HIP_TUNE=$t DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-sweep > gpurun_out/$TAG/prof_t$t.json 2> gpurun_out/$TAG/prof_t$t.err
  echo "== profile tune $t"; grep "density_hip prof" gpurun_out/$TAG/prof_t$t.err | tail -44 | grep -v "  w[2-9]\|  w1[0-5]"
done
for t in $BENCH_TUNES; do
  DENSITY_HIP_TUNE=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-sweep > gpurun_out/$TAG/bench_t$t.json 2>/dev/null
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$TAG/bench_t$t.json")); print("tune $t", d["kernel_ms"], d["value"], d["roofline"]["frac"])
except Exception as ex: print("tune $t bench failed", ex)
PY
done
if [ -n "$EXTRA" ]; then bash -c "$EXTRA"; fi
