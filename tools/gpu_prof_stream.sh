#!/bin/bash
# per-kernel times of the reference-shaped stream calls on one 1 GiB stream: bash tools/gpu_prof_stream.sh <out dir under gpurun_out>
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-prof_stream}; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $GRAFT_REPO_ROOT/tools/gpu_stream_rate.py 1024 rep > $OUT/run.log 2>&1
tail -2 $OUT/run.log
python - <<PY
import csv, glob
f = glob.glob("$OUT/stats/**/*_kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    print(r["Name"][:80].ljust(80), r["Calls"].rjust(5), f'{float(r["AverageNs"]) / 1e3:10.1f} us')
PY
