#!/bin/bash
T=gpurun_out/r4h; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_decode_passes.py tests/test_gpu_shipped_configs.py -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $T/pytest.log
timeout 600 python bench.py --algo cheetah --data prose --size 100000000 --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $T/bench_cheetah.json 2> $T/bench_cheetah.err
python -c "
import json; d=json.load(open('gpurun_out/r4h/bench_cheetah.json')); print(d['value'], d['kernel_ms'], d['compression_ratio'])"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$T/stats_cheetah -- bash -c "cd $OLDPWD && python bench.py --algo cheetah --data prose --size 100000000 --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra" > $OLDPWD/$T/prof.log 2>&1; cd $OLDPWD
python - <<'PY'
import csv, glob, re
for f in glob.glob("gpurun_out/r4h/stats_cheetah/**/*_kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "density::" in row["Name"]:
            print("%-40s calls %4s avg %9.1f us" % (re.sub(r"\(.*", "", row["Name"])[-40:], row["Calls"], float(row["AverageNs"]) / 1e3))
PY
