#!/bin/bash
# The round's ONE GPU job script (rewritten per call; git history keeps the versions): gpurun -- 'bash tools/gpu_job.sh'
# r5u: probes/host_register_trap.hip: what tells a registration answered from the runtime's pin cache from a clean one
T=gpurun_out/r5u; mkdir -p $T; export TMPDIR=/tmp
cd probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o host_register_trap host_register_trap.hip 2>&1 | tail -3; cd ..
timeout 120 ./probes/host_register_trap 2>&1 | tee $T/trap.txt
timeout 120 ./probes/host_register_trap touch 2>&1 | tail -3 | tee $T/trap_touch.txt
