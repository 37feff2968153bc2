#!/bin/bash
# r5s: repetition of the tests that drive the encoder off calm text (aborts, ordered rounds, run-ahead), and the encode fuzzer on fresh seeds
T=gpurun_out/r5s; mkdir -p $T; export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_chameleon.py tests/test_gpu_shipped_configs.py -m gpu -x -q -k "abort or incompressible or hostile or patch or random or mixed" 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $T/repeat.txt
for seed in 11 12 13 14; do timeout 200 python tools/gpu_fuzz_encode.py 60 $seed 2>&1 | grep -v amdgpu.ids | tail -1; done | tee $T/fuzz_seeds.txt
