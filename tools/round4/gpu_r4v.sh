#!/bin/bash
# Lion decoder, four records per step: parity, fuzz, config 4 against variant 2048 (two records per step)
T=gpurun_out/r4v; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_cheetah_lion.py tests/test_gpu_shipped_configs.py tests/test_gpu_patchwork.py tests/test_gpu_slotted.py -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
for f in encode streams; do timeout 400 python tools/gpu_fuzz_$f.py > $T/fuzz_$f.log 2>&1; echo "fuzz $f rc=$?"; tail -2 $T/fuzz_$f.log; done
for v in 0 2048; do timeout 300 python bench.py --algo lion --data prose --size 100000000 --variant $v --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $T/bench_lion_$v.json 2> $T/bench_lion_$v.err; echo "variant $v rc=$?"; python -c "
import json; d=json.load(open('$T/bench_lion_$v.json')); print(d['value'], d['kernel_ms'], d['compression_ratio'])"; done
