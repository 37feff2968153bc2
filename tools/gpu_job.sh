#!/bin/bash
# The round's ONE GPU job script (rewritten per call; git history keeps the versions): gpurun -- 'bash tools/gpu_job.sh'
# r5j: all-raw ordered rounds short-cut: parity of the hostile inputs, then the data kinds' kernel times
T=gpurun_out/r5j; mkdir -p $T; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_shipped_configs.py tests/test_gpu_chameleon.py -m gpu -x -q -k "not rotor-alt and not pipelined and not simple and not hostpipe" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 300 python tools/gpu_data_kinds.py 256 text,zeros,random,mixed 5 2>&1 | grep -v amdgpu.ids | tee $T/data_kinds.txt
