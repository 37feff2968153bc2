#!/bin/bash
# round 6, third GPU call: the grouped walk (128 / 256 quads at a time) against 64 and the chain; paged + multi-rank container tests
T=gpurun_out/r6c; mkdir -p $T; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_paged.py tests/test_gpu_decode_passes.py tests/test_gpu_cheetah_lion.py tests/test_gpu_shipped_configs.py -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $T/pytest.log
for v in 0 8192 16384 4096; do
  timeout 300 python bench.py --algo cheetah --data prose --size 100000000 --steps 8 --warmup 2 --no-cpu --no-sweep --no-extra --variant $v > $T/bench_cheetah_v$v.json 2> $T/bench_cheetah_v$v.err
  python - <<PY
import json
try:
    d=json.load(open("$T/bench_cheetah_v$v.json")); print("cheetah variant $v:", d["value"], "MB/s; kernel_ms", d["kernel_ms"], "ratio", d["compression_ratio"])
except Exception as ex: print("failed", ex, open("$T/bench_cheetah_v$v.err").read()[-800:])
PY
done
for v in 0 8192 16384; do DENSITY_TEST_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_decode_passes.py -q -x 2>&1 | tail -1; done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$T/prof -- python $OLDPWD/bench.py --algo cheetah --data prose --size 100000000 --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > /dev/null 2>&1
cd $OLDPWD; f=$(find $T/prof -name "*kernel_stats.csv" | head -1); cp $f $T/kernel_stats.csv 2>/dev/null; python - <<PY
import csv
try:
    rows=list(csv.DictReader(open("$T/kernel_stats.csv")))
    for r in rows[:16]: print(r["Name"][:80], r["Calls"], r["AverageNs"])
except Exception as ex: print("no stats", ex)
PY
