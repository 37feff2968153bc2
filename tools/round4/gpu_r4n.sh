#!/bin/bash
# the reference symbols on host pointers, pipelined: parity, then the rates
T=gpurun_out/r4n; mkdir -p $T
timeout 1200 python -m pytest tests/test_gpu_host_stream_pipeline.py -m gpu -x -q > $T/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 $T/pytest_new.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-sweep --no-extra > $T/bench.json 2> $T/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$T/bench.json")); print(json.dumps(d["host_api"], indent=1))
PY
