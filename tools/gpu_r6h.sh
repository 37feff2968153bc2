#!/bin/bash
# round 6: the walk by a team of four waves — parity on every walk variant, then config 3
T=gpurun_out/r6h; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode_passes.py tests/test_gpu_cheetah_lion.py tests/test_gpu_shipped_configs.py tests/test_gpu_patchwork.py -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
for v in 8192 16384 4096; do DENSITY_TEST_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_decode_passes.py -q -x 2>&1 | tail -1; done
timeout 300 python tools/gpu_fuzz_passes.py > $T/fuzz_passes.log 2>&1; echo "fuzz rc=$?"; tail -2 $T/fuzz_passes.log
for v in 0 16384; do
timeout 300 python bench.py --algo cheetah --data prose --size 100000000 --steps 8 --warmup 2 --no-cpu --no-sweep --no-extra --variant $v > $T/bench_cheetah_$v.json 2> $T/bench_cheetah.err
python - <<PY
import json
try:
    d=json.load(open("$T/bench_cheetah_$v.json")); print("cheetah variant $v:", d["value"], "MB/s; kernel_ms", d["kernel_ms"], "ratio", d["compression_ratio"])
except Exception as ex: print("failed", ex, open("$T/bench_cheetah.err").read()[-1500:])
PY
done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$T/prof -- python $OLDPWD/bench.py --algo cheetah --data prose --size 100000000 --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > /dev/null 2>&1
cd $OLDPWD; f=$(find $T/prof -name "*kernel_stats.csv" | head -1); cp $f $T/kernel_stats.csv 2>/dev/null; python - <<PY
import csv
rows=list(csv.DictReader(open("$T/kernel_stats.csv")))
for r in rows[:8]: print(r["Name"][:80], r["Calls"], r["AverageNs"])
PY
timeout 300 python benches/density.py 2>&1 | grep -A8 "cheetah" | grep -A1 "stream decompr"
