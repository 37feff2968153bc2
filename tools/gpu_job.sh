#!/bin/bash
# r5w: the bench line of the final tree
T=gpurun_out/r5w; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"; head -c 160 $T/bench_full.json; echo
