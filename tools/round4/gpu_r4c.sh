#!/bin/bash
T=gpurun_out/r4c; mkdir -p $T; export TMPDIR=/tmp
./probes/vmem_width 2>&1 | grep -v amdgpu.ids | tee $T/vmem_width.txt
timeout 600 python tools/gpu_variants.py 20 r03 saltold encA encB encC 2>&1 | grep -v amdgpu.ids | tee $T/variants_t0.txt
