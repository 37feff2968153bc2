#!/bin/bash
# Cheetah / Lion config-3/4 stand-in at several chunk sizes + per-kernel times of the decode passes
T=gpurun_out/${1:-r3g}; mkdir -p $T; export TMPDIR=/tmp
for a in ${ALGOS:-cheetah}; do for c in ${CHUNKS:-1048576 262144 65536}; do
  timeout 600 python bench.py --algo $a --data prose --size 100000000 --chunk $c --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $T/bench_${a}_$c.json 2> $T/bench_${a}_$c.err
  python - <<PY
import json
try:
    d=json.load(open("$T/bench_${a}_$c.json")); print("$a chunk $c:", d["value"], "MB/s; kernel_ms", d["kernel_ms"], "ratio", d["compression_ratio"])
except Exception as ex: print("$a $c failed", ex, open("$T/bench_${a}_$c.err").read()[-800:])
PY
done; done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$T/prof -- python $OLDPWD/bench.py --algo cheetah --data prose --size 100000000 --chunk 1048576 --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > /dev/null 2>&1
cd $OLDPWD; f=$(find $T/prof -name "*kernel_stats.csv" | head -1); cp $f $T/kernel_stats.csv 2>/dev/null; python - <<PY
import csv
try:
    rows=list(csv.DictReader(open("$T/kernel_stats.csv")))
    for r in rows[:16]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
except Exception as ex: print("no stats", ex)
PY
