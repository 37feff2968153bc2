#!/bin/bash
# round 6: whole GPU suite + smoke + the driver's bench line on the tree with the fused / dense dictionary pass and pinned operands
T=gpurun_out/r6f; mkdir -p $T; export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 200 python __graft_entry__.py smoke > $T/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $T/smoke.log
timeout 900 python bench.py > $T/bench_default.json 2> $T/bench_default.err; echo "bench rc=$?"; head -c 300 $T/bench_default.json; echo
