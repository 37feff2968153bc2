#!/bin/bash
T=gpurun_out/r4l; mkdir -p $T; export TMPDIR=/tmp
for c in 65536 131072 262144; do
  timeout 600 python bench.py --algo lion --data prose --size 100000000 --chunk $c --steps 3 --warmup 1 --no-cpu --no-sweep --no-extra > $T/lion_$c.json 2> $T/lion_$c.err
  python -c "
import json; d=json.load(open('gpurun_out/r4l/lion_$c.json')); print($c, d['value'], d['kernel_ms'], d['compression_ratio'])"
done
