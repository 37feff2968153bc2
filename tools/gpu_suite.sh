#!/bin/bash
# usage: tools/gpu_suite.sh <tag> [pytest args]  — full GPU suite (or the tests named), log under gpurun_out/<tag>/
TAG=${1:-x}; shift; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q --durations=15 "$@" > gpurun_out/$TAG/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/$TAG/pytest.log
