#!/bin/bash
T=gpurun_out/r4p; mkdir -p $T
timeout 600 python tools/gpu_host_stream_trace.py > $T/plain.log 2>&1; echo "plain rc=$?"; grep -v "prof\]" $T/plain.log | tail -25
DENSITY_HIP_PROF=1 timeout 600 python tools/gpu_host_stream_trace.py > $T/prof.log 2>&1; echo "prof rc=$?"
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_host_stream_pipeline.py -m gpu -x -q -s > $T/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -5 $T/pytest_new.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-sweep --no-extra > $T/bench.json 2> $T/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$T/bench.json")); print(json.dumps(d["host_api"]["reference_symbols"]))
PY
