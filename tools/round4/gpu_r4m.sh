#!/bin/bash
# after the Lion two-records-per-step kernels and the 128 KiB automatic chunk: shipped configurations, Lion / Cheetah parity, fuzzers
T=gpurun_out/r4m; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_shipped_configs.py tests/test_gpu_cheetah_lion.py tests/test_gpu_decode_passes.py tests/test_gpu_slotted.py -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
for f in encode streams passes; do timeout 400 python tools/gpu_fuzz_$f.py > $T/fuzz_$f.log 2>&1; echo "fuzz $f rc=$?"; tail -2 $T/fuzz_$f.log; done
timeout 300 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $T/bench_lion.json 2> $T/bench_lion.err; echo "lion rc=$?"; head -c 400 $T/bench_lion.json
