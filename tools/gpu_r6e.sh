#!/bin/bash
# round 6: the fused dictionary pass (order + cells by quarters, next trip's loads in flight) — parity, then config 3's kernel times
T=gpurun_out/r6e; mkdir -p $T; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_decode_passes.py tests/test_gpu_cheetah_lion.py tests/test_gpu_shipped_configs.py -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $T/pytest.log
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
    for r in rows[:16]: print(r["Name"][:80], r["Calls"], r["AverageNs"])
except Exception as ex: print("no stats", ex)
PY
