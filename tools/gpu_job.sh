#!/bin/bash
T=gpurun_out/r5r; mkdir -p $T; export TMPDIR=/tmp
timeout 200 python tools/gpu_phase_prof.py random 2>&1 | grep "events: fast\|D chain\|memo of" | head -3 | tee $T/prof_random.txt
