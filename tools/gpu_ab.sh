#!/bin/bash
# A/B on ONE box, in one process, on the same buffers: the library in the tree against probes/lib_old.so (a build of an earlier state), alternating;
# usage: tools/gpu_ab.sh <tag> [pytest -k expression]   (PROF=1: the phase profile of the tree's library too; EXTRA: more commands)
T=gpurun_out/${1:-ab}; mkdir -p $T; export TMPDIR=/tmp
K=${2:-rotor}
timeout 900 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "$K" > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $T/pytest.log
timeout 600 python tools/gpu_kernel_time.py 20 probes/lib_old.so 2>&1 | grep -v amdgpu.ids | tee $T/ab.txt
if [ -n "$PROF" ]; then
  DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-sweep --no-extra > $T/prof.json 2> $T/prof.err
  grep "density_hip prof" $T/prof.err | grep -v "  w[2-9] \|  w1[0-5] " | tail -16
fi
if [ -n "$EXTRA" ]; then bash -c "$EXTRA"; fi
