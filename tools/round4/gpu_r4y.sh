#!/bin/bash
T=gpurun_out/r4_final; mkdir -p $T
timeout 300 python -m pytest tests/test_gpu_host_stream_pipeline.py -m gpu -x -q > $T/pytest_host_pipeline.log 2>&1; echo "host pipeline tests rc=$?"; tail -1 $T/pytest_host_pipeline.log
timeout 600 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"
python - <<PY
import json
b = json.load(open("$T/bench_full.json")); print(b["value"], b["kernel_ms"], b["roofline"]["frac"], b["roofline"]["traffic"], b["value_packed"])
h = b["host_api"]
for k in ("reference_symbols", "container", "container_large"): print(k, h[k]["encode_MBps"], h[k]["decode_MBps"])
PY
