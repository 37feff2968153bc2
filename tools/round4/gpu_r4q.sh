#!/bin/bash
T=gpurun_out/r4q; mkdir -p $T
for cfg in "0,4 torch" "0,4 torch" "0,512 torch"; do
  set -- $cfg
  timeout 300 python tools/gpu_host_stream_sequence.py $1 $2 > $T/run_$1_$2.log 2>&1; echo "variants $1 $2: rc=$?"; grep -i "fault\|done" $T/run_$1_$2.log | head -3
done
timeout 1200 python -X faulthandler -m pytest tests/test_gpu_host_stream_pipeline.py -m gpu -x -q > $T/pytest_new.log 2>&1; echo "new tests rc=$?"; tail -2 $T/pytest_new.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-sweep --no-extra > $T/bench.json 2> $T/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$T/bench.json"))["host_api"]
for k in ("reference_symbols", "container", "container_large"): print(k, d[k]["encode_MBps"], d[k]["decode_MBps"])
PY
