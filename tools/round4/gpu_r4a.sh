#!/bin/bash
# round 4, first GPU call: the new parity tests, the decoder geometries (rare paths rolled: rounds of 16 / 20 on 12 waves, 12 on 16), their times and phase profiles
T=gpurun_out/r4a; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_shipped_configs.py -m gpu -x -q > $T/pytest_shipped.log 2>&1; echo "shipped rc=$?"; tail -3 $T/pytest_shipped.log
for t in 0 64 96 128; do
  DENSITY_HIP_TUNE=$t timeout 600 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "rotor and not full_size and not beyond_2 and not long_stream" > $T/pytest_tune$t.log 2>&1; echo "tune $t pytest rc=$?"; tail -2 $T/pytest_tune$t.log
done
bash tools/gpu_tunes.sh r4a "0 64 96 128 8 72" 2>&1 | grep -v amdgpu.ids
for t in 0 64 96 8; do
  DENSITY_HIP_TUNE=$t DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-sweep --no-extra > $T/prof_t$t.json 2> $T/prof_t$t.err
  echo "== prof tune $t"; grep "density_hip prof" $T/prof_t$t.err | grep -v "  w[2-9] \|  w1[0-5] " | tail -12
done
