#!/bin/bash
# usage: tools/gpu_check.sh <tag> [pytest-args...]   — GPU suite, profile of one step, bench (default + extra variants via $EXTRA_TUNES)
TAG=${1:-x}; shift
mkdir -p gpurun_out/$TAG
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/$TAG/pytest.log
DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu > gpurun_out/$TAG/prof.json 2> gpurun_out/$TAG/prof.err
grep "density_hip prof" gpurun_out/$TAG/prof.err | tail -42 | grep -v "  w[2-9]\|  w1[0-5]"
for t in 0 $EXTRA_TUNES; do
  DENSITY_HIP_TUNE=$t timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/$TAG/bench_t$t.json 2>/dev/null
  python - <<PY
import json; d=json.load(open("gpurun_out/$TAG/bench_t$t.json")); print("tune $t", d["kernel_ms"], d["value"], d["roofline"]["frac"])
PY
done
