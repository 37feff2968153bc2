#!/bin/bash
mkdir -p gpurun_out/$1; export TMPDIR=/tmp
for c in 1048576 2097152 4194304 1048576 2097152 4194304; do
  timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --chunk $c > gpurun_out/$1/bench_c$c.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/$1/bench_c$c.json")); print("chunk $c", d["kernel_ms"], d["value"], d["roofline"]["frac"], d["compression_ratio"], d["whole_path_hbm_frac"])
PY
done
