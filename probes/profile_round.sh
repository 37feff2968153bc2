#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate counter runs.
# Usage on the GPU box (from the repo root):  bash probes/profile_round.sh gpurun_out/prof
set -e
OUT=${1:-gpurun_out/prof}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$OUT
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- python bench.py --steps 5 --warmup 1 --no-cpu > $R/$OUT/bench_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/fetch -- python bench.py --steps 2 --warmup 1 --no-cpu > $R/$OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/write -- python bench.py --steps 2 --warmup 1 --no-cpu > $R/$OUT/bench_write.log 2>&1
find $R/$OUT -name "*_kernel_stats.csv" -o -name "*_counter_collection.csv"
