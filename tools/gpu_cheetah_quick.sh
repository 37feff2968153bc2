#!/bin/bash
# config 3 (Cheetah, 100 MB of prose at the automatic chunk): round trip + kernel times, and the decode-pass parity tests
T=gpurun_out/${1:-quick}; mkdir -p $T; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_decode_passes.py tests/test_gpu_cheetah_lion.py tests/test_gpu_shipped_configs.py tests/test_gpu_patchwork.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --algo cheetah --data prose --size 100000000 --steps 8 --warmup 2 --no-cpu --no-sweep --no-extra > $T/bench_cheetah.json 2> $T/bench_cheetah.err
python - <<PY
import json
try:
    d=json.load(open("$T/bench_cheetah.json")); print("cheetah:", d["value"], "MB/s; kernel_ms", d["kernel_ms"], "ratio", d["compression_ratio"])
except Exception as ex: print("failed", ex, open("$T/bench_cheetah.err").read()[-1500:])
PY
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$T/prof -- python $OLDPWD/bench.py --algo cheetah --data prose --size 100000000 --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > /dev/null 2>&1
cd $OLDPWD; f=$(find $T/prof -name "*kernel_stats.csv" | head -1); cp $f $T/kernel_stats.csv 2>/dev/null; python - <<PY
import csv
try:
    rows=list(csv.DictReader(open("$T/kernel_stats.csv")))
    for r in rows[:14]: print(r["Name"][:80], r["Calls"], r["AverageNs"])
except Exception as ex: print("no stats", ex)
PY
