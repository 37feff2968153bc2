#!/bin/bash
# The round's ONE GPU job script (rewritten per call; git history keeps the versions): gpurun -- 'bash tools/gpu_job.sh'
# r5r: exact slot-0 filter in the decoder.s rare branch: whole GPU suite, A/B against round 4's library, data kinds
T=gpurun_out/r5r; mkdir -p $T; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 400 python tools/gpu_variants.py 10 r04 2>&1 | grep -v amdgpu.ids | tee $T/ab_1g.txt
timeout 600 python tools/gpu_data_kinds.py 2>&1 | grep -v amdgpu.ids | tee $T/kinds.txt
