#!/bin/bash
# The round's ONE GPU job script (rewritten per call; git history keeps the versions): gpurun -- 'bash tools/gpu_job.sh'
# r5w: the three container forms side by side on the headline workload
T=gpurun_out/r5w; mkdir -p $T; export TMPDIR=/tmp
timeout 600 python tools/gpu_forms.py 10 2>&1 | grep -v amdgpu.ids | tee $T/forms.txt
