#!/bin/bash
# after the clean-up (one Lion kernel pair): Lion / Cheetah parity again, then the closing run's profile passes for the new kernels id
T=gpurun_out/r4z; mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_cheetah_lion.py tests/test_gpu_shipped_configs.py tests/test_gpu_patchwork.py tests/test_gpu_slotted.py tests/test_gpu_decode_passes.py -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $T/pytest.log
for f in encode streams passes; do timeout 200 python tools/gpu_fuzz_$f.py > $T/fuzz_$f.log 2>&1; echo "fuzz $f rc=$?"; tail -1 $T/fuzz_$f.log; done
