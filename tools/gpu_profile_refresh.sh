#!/bin/bash
# after a change to a kernel source: the parity tests of what changed (arguments), then the profile passes and the bench line again — so that
# profiles/rNN_pmc_summary.json carries the library's kernels id and bench.py quotes the counter traffic
T=gpurun_out/r4_final; mkdir -p $T; export TMPDIR=/tmp
if [ $# -gt 0 ]; then timeout 900 python -m pytest "$@" -m gpu -x -q > $T/pytest_refresh.log 2>&1; echo "pytest rc=$?"; tail -2 $T/pytest_refresh.log; fi
bash probes/profile_round.sh gpurun_out/prof4 > $T/profile_round.log 2>&1; echo "profile rc=$?"
