#!/bin/bash
# round 6, session 2 baseline: the whole GPU suite, smoke, the driver's bench line, the bench harness (reference symbols on one stream)
T=gpurun_out/r6d; mkdir -p $T; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 200 python __graft_entry__.py smoke > $T/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $T/smoke.log
timeout 900 python bench.py > $T/bench_default.json 2> $T/bench_default.err; echo "bench rc=$?"; head -c 400 $T/bench_default.json; echo
timeout 400 python benches/density.py > $T/benches_density.txt 2>&1; echo "harness rc=$?"; tail -24 $T/benches_density.txt
