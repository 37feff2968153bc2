#!/bin/bash
mkdir -p gpurun_out/r3c; export TMPDIR=/tmp
DENSITY_HIP_TUNE=256 timeout 900 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "rotor and not config2 and not long_stream" > gpurun_out/r3c/pytest_t256.log 2>&1; echo "pytest tune 256 rc=$?"; tail -2 gpurun_out/r3c/pytest_t256.log
bash tools/gpu_tunes.sh r3c "0 256 0 256"
for t in 0 256; do
  DENSITY_HIP_TUNE=$t DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-sweep --no-extra > gpurun_out/r3c/prof_t$t.json 2> gpurun_out/r3c/prof_t$t.err
  echo "== profile tune $t"; grep "density_hip prof" gpurun_out/r3c/prof_t$t.err | tail -44 | grep -v "  w[2-9]\|  w1[0-5]" | tail -14
done
