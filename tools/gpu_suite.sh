#!/bin/bash
# usage: tools/gpu_suite.sh <tag>  — full GPU suite, then bench for the variants in $VARIANTS (default "0")
TAG=${1:-x}; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/$TAG/pytest.log
for v in ${VARIANTS:-0}; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu --variant $v > gpurun_out/$TAG/bench_v$v.json 2>gpurun_out/$TAG/bench_v$v.err
  python - <<PY
import json
try:
    d=json.load(open("gpurun_out/$TAG/bench_v$v.json")); print("variant $v", d["kernel_ms"], "ms/step", d["ms_per_step"], d["value"], d["roofline"]["frac"], d["whole_path_hbm_frac"]); print("   sweep", d.get("size_sweep"))
except Exception as ex: print("variant $v bench failed", ex, open("gpurun_out/$TAG/bench_v$v.err").read()[-600:])
PY
done
