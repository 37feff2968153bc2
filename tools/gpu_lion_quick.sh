#!/bin/bash
# config 4 (Lion, 100 MB of prose at the automatic chunk): parity suites, then round trip + kernel times, default and the one-wave decoder (variant 32768)
T=gpurun_out/${1:-lionq}; mkdir -p $T; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_cheetah_lion.py tests/test_gpu_shipped_configs.py tests/test_gpu_patchwork.py -x -q 2>&1 | tail -3
timeout 600 python tools/gpu_fuzz_streams.py 150 2>&1 | grep -v amdgpu.ids | grep lion
for v in 0 32768; do
timeout 300 python bench.py --algo lion --data prose --size 100000000 --steps 6 --warmup 2 --no-cpu --no-sweep --no-extra --variant $v > $T/bench_lion_$v.json 2> $T/bench_lion.err
python - <<PY
import json
try:
    d=json.load(open("$T/bench_lion_$v.json")); print("lion variant $v:", d["value"], "MB/s; kernel_ms", d["kernel_ms"], "ratio", d["compression_ratio"])
except Exception as ex: print("failed", ex, open("$T/bench_lion.err").read()[-1500:])
PY
done
