#!/bin/bash
# A/B on ONE box: the library in the tree against probes/lib_old.so (a build of the previous commit), alternating;  usage: tools/gpu_ab.sh <tag> [pytest -k expression]
T=gpurun_out/${1:-ab}; mkdir -p $T; export TMPDIR=/tmp
K=${2:-rotor}
timeout 900 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "$K" > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $T/pytest.log
cp density_amd/libdensity_hip.so /tmp/lib_new.so
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then cp probes/lib_old.so density_amd/libdensity_hip.so; else cp /tmp/lib_new.so density_amd/libdensity_hip.so; fi
    echo -n "$v: "; timeout 200 python tools/gpu_kernel_time.py 20 2>&1 | tail -1
  done
done | tee $T/ab.txt
cp /tmp/lib_new.so density_amd/libdensity_hip.so
DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-sweep --no-extra > $T/prof.json 2> $T/prof.err
grep "density_hip prof" $T/prof.err | grep -v "  w[2-9] \|  w1[0-5] " | tail -16
if [ -n "$EXTRA" ]; then bash -c "$EXTRA"; fi
