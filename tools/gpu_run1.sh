#!/bin/bash
# first light of the wave-rotation kernels: diagnostics, full GPU suite, bench (default / late token / old pipelines)
mkdir -p gpurun_out/r2a
export TMPDIR=/tmp
timeout 600 python tools/gpu_first_light.py > gpurun_out/r2a/first_light.log 2>&1; echo "first_light rc=$?"
tail -40 gpurun_out/r2a/first_light.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 gpurun_out/r2a/pytest.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/r2a/bench_rot.json 2> gpurun_out/r2a/bench_rot.err; echo "bench rc=$?"; cat gpurun_out/r2a/bench_rot.json; tail -3 gpurun_out/r2a/bench_rot.err
DENSITY_HIP_TUNE=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu > gpurun_out/r2a/bench_rot_late.json 2> gpurun_out/r2a/bench_rot_late.err; echo "bench late rc=$?"; cat gpurun_out/r2a/bench_rot_late.json
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu --variant 4 > gpurun_out/r2a/bench_pipe.json 2> gpurun_out/r2a/bench_pipe.err; echo "bench pipe rc=$?"; cat gpurun_out/r2a/bench_pipe.json
