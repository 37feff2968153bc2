#!/bin/bash
# r5u: the bench line of the final tree (counter traffic of the headline kernels and of configs 3 / 4 quoted)
T=gpurun_out/r5u; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"; head -c 200 $T/bench_full.json; echo
