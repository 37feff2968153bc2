#!/bin/bash
# The round's ONE GPU job script (rewritten per call; git history keeps the versions): gpurun -- 'bash tools/gpu_job.sh'
# r5n: same-box A/B against round 4's library with the ordered path apart from the fast path again (1 GiB twice, 100 MB), the data kinds, parity
T=gpurun_out/r5n; mkdir -p $T; export TMPDIR=/tmp
timeout 400 python tools/gpu_variants.py 10 r04 2>&1 | grep -v amdgpu.ids | tee $T/ab_1g.txt
timeout 400 python tools/gpu_variants.py 10 r04 2>&1 | grep -v amdgpu.ids | tee -a $T/ab_1g.txt
DENSITY_AB_BYTES=100000000 timeout 300 python tools/gpu_variants.py 20 r04 2>&1 | grep -v amdgpu.ids | tee $T/ab_100m.txt
timeout 600 python tools/gpu_data_kinds.py 2>&1 | grep -v amdgpu.ids | tee $T/kinds.txt
timeout 900 python -m pytest tests/test_gpu_chameleon.py tests/test_gpu_patchwork.py -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
