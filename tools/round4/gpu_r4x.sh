#!/bin/bash
T=gpurun_out/r4x; mkdir -p $T
timeout 600 python tools/gpu_r4w.py > $T/slices.log 2>&1; echo "slices rc=$?"; cat $T/slices.log | grep MiB
for c in 65536 131072 262144; do timeout 200 python bench.py --algo lion --data prose --size 100000000 --chunk $c --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $T/lion_$c.json 2>/dev/null; python -c "
import json; d=json.load(open('$T/lion_$c.json')); print($c, d['value'], d['kernel_ms'], d['compression_ratio'])"; done
