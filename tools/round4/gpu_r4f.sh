#!/bin/bash
T=gpurun_out/r4f; mkdir -p $T; export TMPDIR=/tmp
./probes/vmem_width 2>&1 | grep -v amdgpu.ids | tee $T/vmem_width.txt
DENSITY_HIP_TUNE=512 timeout 900 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "rotor and not beyond_2" > $T/pytest512.log 2>&1; echo "pytest tune 512 rc=$?"; tail -1 $T/pytest512.log
for t in 0 512 0 512; do echo "== tune $t"; DENSITY_HIP_TUNE=$t timeout 300 python tools/gpu_variants.py 20 2>&1 | grep -v amdgpu.ids | tail -1; done
DENSITY_HIP_TUNE=512 DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-sweep --no-extra > $T/prof_t512.json 2> $T/prof_t512.err
grep "density_hip prof" $T/prof_t512.err | grep -v "  w[2-9] \|  w1[0-5] " | tail -12
