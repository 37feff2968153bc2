#!/bin/bash
# The round's ONE GPU job script (rewritten per call; git history keeps the versions): gpurun -- 'bash tools/gpu_job.sh'
# r5v: paged container, encoder side: streams == oracle after CPU reassembly
T=gpurun_out/r5v; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_paged.py -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $T/pytest.log
