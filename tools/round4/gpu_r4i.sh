#!/bin/bash
T=gpurun_out/r4i; mkdir -p $T; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $T/pytest.log
timeout 900 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r4i/bench_full.json')); print(d['value'], d['kernel_ms'], d['roofline']); print(d['size_sweep'])"
