#!/bin/bash
# Counter traffic of configs 3 / 4 per DIRECTION (the review's item: `roofline.traffic` of the Cheetah / Lion entries was null): FETCH_SIZE and WRITE_SIZE in
# separate passes over the bench's own legs.  Usage on the GPU box (from the repo root):  bash probes/profile_directions.sh gpurun_out/prof5d ; then, here,
# python probes/direction_summary.py gpurun_out/prof5d r05
OUT=${1:-gpurun_out/prof5d}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$OUT
cd $R
for algo in cheetah lion; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/$OUT/${algo}_$ctr -- python bench.py --algo $algo --data prose --size 100000000 --settle-ms 0 --steps 2 --warmup 1 --no-cpu --no-sweep --no-extra > $R/$OUT/${algo}_$ctr.log 2>&1
    echo "$algo $ctr rc=$?"
  done
done
find $R/$OUT -name "*_counter_collection.csv"
