#!/bin/bash
# r5n: run-ahead with the token inside the exchange statement and tight polls: parity, data kinds, D chain
T=gpurun_out/r5n; mkdir -p $T; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_shipped_configs.py tests/test_gpu_chameleon.py -m gpu -x -q -k "not pipelined and not simple and not hostpipe and not alt" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python tools/gpu_data_kinds.py 256 text,zeros,random,mixed 5 2>&1 | grep -v amdgpu.ids | tee $T/data_kinds.txt
timeout 200 python tools/gpu_phase_prof.py random 2>&1 | grep "events: fast\|D chain" | head -2 | tee $T/prof_random.txt
