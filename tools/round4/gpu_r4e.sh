#!/bin/bash
T=gpurun_out/r4e; mkdir -p $T; export TMPDIR=/tmp
for t in 160 192 224; do
  DENSITY_HIP_TUNE=$t timeout 600 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "rotor and not full_size and not beyond_2 and not long_stream" > $T/pytest$t.log 2>&1; echo "pytest tune $t rc=$?"; tail -1 $T/pytest$t.log
done
for t in 0 160 192 224; do echo "== tune $t"; DENSITY_HIP_TUNE=$t timeout 300 python tools/gpu_variants.py 20 2>&1 | grep -v amdgpu.ids | tail -1; done
for t in 160 224; do
  DENSITY_HIP_TUNE=$t DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-sweep --no-extra > $T/prof_t$t.json 2> $T/prof_t$t.err
  echo "== prof tune $t"; grep "density_hip prof" $T/prof_t$t.err | grep -v "  w[2-9] \|  w1[0-5] " | tail -5
done
