#!/bin/bash
# Cheetah: records of calm chunks found by window kernels — parity (window parse == one-wave parse == one-wave decoder == oracle), fuzz, config 3
T=gpurun_out/r4u; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_decode_passes.py tests/test_gpu_shipped_configs.py tests/test_gpu_cheetah_lion.py tests/test_gpu_patchwork.py -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 400 python tools/gpu_fuzz_passes.py > $T/fuzz_passes.log 2>&1; echo "fuzz passes rc=$?"; tail -3 $T/fuzz_passes.log
for v in 0 1024; do timeout 300 python bench.py --algo cheetah --data prose --size 100000000 --variant $v --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $T/bench_cheetah_$v.json 2> $T/bench_cheetah_$v.err; echo "variant $v rc=$?"; python -c "
import json; d=json.load(open('$T/bench_cheetah_$v.json')); print(d['value'], d['kernel_ms'], d['compression_ratio'])"; done
