#!/bin/bash
T=gpurun_out/r4j; mkdir -p $T; export TMPDIR=/tmp
timeout 500 python tools/gpu_fuzz_encode.py 150 41 > $T/fuzz_encode.txt 2>&1; echo "fuzz encode rc=$?"; tail -3 $T/fuzz_encode.txt
timeout 500 python tools/gpu_fuzz_streams.py 150 42 > $T/fuzz_streams.txt 2>&1; echo "fuzz streams rc=$?"; tail -3 $T/fuzz_streams.txt
timeout 500 python tools/gpu_fuzz_passes.py 150 43 > $T/fuzz_passes.txt 2>&1; echo "fuzz passes rc=$?"; tail -3 $T/fuzz_passes.txt
