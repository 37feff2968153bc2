#!/bin/bash
mkdir -p gpurun_out/r2c
export TMPDIR=/tmp
for t in 0 2 1; do
DENSITY_HIP_TUNE=$t DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu > gpurun_out/r2c/prof$t.json 2> gpurun_out/r2c/prof$t.err; echo "== tune $t"; grep "density_hip prof" gpurun_out/r2c/prof$t.err | tail -42 | grep -v "  w[1-9]"
done
