#!/bin/bash
# The round's ONE GPU job script (rewritten per call; git history keeps the versions): gpurun -- 'bash tools/gpu_job.sh'
# r5a: the split encoder (8 chain + 8 emit waves) against the 8-wave one — parity of the split one first, then the A/B
T=gpurun_out/r5a; mkdir -p $T; export TMPDIR=/tmp
timeout 420 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "rotor-alt" 2>&1 | grep -v amdgpu.ids | tail -15 | tee $T/pytest_alt.txt
DENSITY_TEST_VARIANT=2048 timeout 300 python -m pytest tests/test_gpu_paged.py tests/test_gpu_slotted.py tests/test_gpu_shipped_configs.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 | tee $T/pytest_alt2.txt
timeout 400 python tools/gpu_split_ab.py 10 2>&1 | grep -v amdgpu.ids | tee $T/split_ab.txt
