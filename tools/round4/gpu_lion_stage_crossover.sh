#!/bin/bash
# Lion encode: exchange passes against the one-wave kernels (four blocks per step since round 4) by chunk count
T=gpurun_out/r4_lion_cross; mkdir -p $T
for cfg in "10000000 65536" "33554432 131072" "100000000 262144" "100000000 524288" "100000000 1048576"; do
  set -- $cfg
  for most in 0 1; do
    if [ $most = 1 ]; then export DENSITY_HIP_STAGE_MOST=1; else unset DENSITY_HIP_STAGE_MOST; fi
    timeout 200 python bench.py --algo lion --data prose --size $1 --chunk $2 --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $T/l_$1_$2_$most.json 2>/dev/null
    python -c "
import json; d=json.load(open('$T/l_$1_$2_$most.json')); print('$1 B, chunk $2, passes', 'off' if $most else 'on ', d['kernel_ms']['lion_encode_chunks'], 'ms encode')"
  done
done
