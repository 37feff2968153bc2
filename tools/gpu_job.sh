#!/bin/bash
# round 6's closing run: GPU suite, smoke, the driver's bench line, the rocprofv3 passes behind profiles/r06_*, the bench harness, the three container forms, the
# fuzzers.  Afterwards, here: python probes/profile_summary.py gpurun_out/prof6 r06 ; python probes/direction_summary.py gpurun_out/prof6d r06 ; cp the rest into
# profiles/ (see profiles/README.md).  Before it, here: hipcc --offload-arch=gfx950 -O3 probes/fetch_calib.hip -o probes/fetch_calib (and vmem_width), python -m density_amd.build --debug
T=gpurun_out/r6_final; mkdir -p $T; export TMPDIR=/tmp
date +%s > $T/t0
timeout 1200 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 200 python __graft_entry__.py smoke > $T/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $T/smoke.log
echo "elapsed $(( $(date +%s) - $(cat $T/t0) )) s"
bash probes/profile_round.sh gpurun_out/prof6 > $T/profile_round.log 2>&1; echo "profile rc=$?"; tail -3 $T/profile_round.log
bash probes/profile_directions.sh gpurun_out/prof6d > $T/profile_directions.log 2>&1; echo "directions rc=$?"
echo "elapsed $(( $(date +%s) - $(cat $T/t0) )) s"
timeout 900 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"; head -c 300 $T/bench_full.json; echo
timeout 400 python benches/density.py > $T/benches_density.txt 2>&1; echo "harness rc=$?"; tail -24 $T/benches_density.txt
timeout 200 python tools/gpu_forms.py 10 2>&1 | grep -v amdgpu.ids > $T/forms.txt; tail -6 $T/forms.txt
for f in encode streams passes tail paged; do timeout 200 python tools/gpu_fuzz_$f.py > $T/fuzz_$f.log 2>&1; echo "fuzz $f rc=$?"; tail -2 $T/fuzz_$f.log; done
echo "elapsed $(( $(date +%s) - $(cat $T/t0) )) s"
