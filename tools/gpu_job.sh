#!/bin/bash
# r5zz: the bench line of the final tree and library
T=gpurun_out/r5zz; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"; head -c 160 $T/bench_full.json; echo
