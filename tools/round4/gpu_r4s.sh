#!/bin/bash
# pipelined reference symbols, host-driven: parity, the alternating-paths case, the rates in a plain process and inside bench.py
T=gpurun_out/r4s; mkdir -p $T
timeout 600 python -X faulthandler -m pytest tests/test_gpu_host_stream_pipeline.py -m gpu -x -q > $T/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -2 $T/pytest_new.log
timeout 900 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "stream" > $T/pytest_stream.log 2>&1; echo "stream tests rc=$?"; tail -2 $T/pytest_stream.log
timeout 300 python tools/gpu_host_stream_trace.py 2>&1 | grep "^round\|staged" | head -8
timeout 600 python bench.py --steps 5 --warmup 2 --no-sweep --no-extra > $T/bench.json 2> $T/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$T/bench.json"))
h = d["host_api"]
for k in ("reference_symbols", "container", "container_large"): print(k, h[k]["encode_MBps"], h[k]["decode_MBps"])
PY
