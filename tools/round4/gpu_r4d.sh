#!/bin/bash
T=gpurun_out/r4d; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "rotor and not beyond_2" > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 600 python tools/gpu_variants.py 20 noprep r03 2>&1 | grep -v amdgpu.ids | tee $T/variants.txt
DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-sweep --no-extra > $T/prof.json 2> $T/prof.err
grep "density_hip prof" $T/prof.err | grep -v "  w[2-9] \|  w1[0-5] " | tail -12
