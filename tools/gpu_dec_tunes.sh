#!/bin/bash
export TMPDIR=/tmp
for t in 64 0 64 0; do echo "tune $t"; DENSITY_HIP_TUNE=$t python tools/gpu_kernel_time.py; done
DENSITY_HIP_TUNE=64 timeout 900 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "rotor and not config2 and not long_stream" 2>&1 | tail -2
