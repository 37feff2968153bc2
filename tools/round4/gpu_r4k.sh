#!/bin/bash
T=gpurun_out/r4k; mkdir -p $T; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_cheetah_lion.py tests/test_gpu_shipped_configs.py -m gpu -x -q -k "lion" > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 300 python tools/gpu_lion_variants.py 5 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/gpu_lion_variants.py 3 --chunk 131072 2>&1 | grep -v amdgpu.ids
