#!/bin/bash
# r5p: the bench line again, as the driver runs it (all-cores rows: fastest of five + median)
T=gpurun_out/r5p; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"; head -c 200 $T/bench_full.json; echo
