#!/bin/bash
# the Cheetah / Lion lines of tools/gpu_round_end.sh alone
T=gpurun_out/r2_final; mkdir -p $T; export TMPDIR=/tmp
for a in cheetah lion; do
  timeout 900 python bench.py --algo $a --data prose --size 100000000 --chunk 1048576 --steps 3 --warmup 1 --no-sweep > $T/bench_${a}_1M.json 2> $T/bench_${a}_1M.err
  timeout 900 python bench.py --algo $a --data prose --size 100000000 --chunk 65536 --steps 3 --warmup 1 --no-cpu --no-sweep > $T/bench_${a}_64K.json 2> $T/bench_${a}_64K.err
  timeout 600 python bench.py --algo $a --steps 2 --warmup 1 --no-cpu --no-sweep > $T/bench_${a}_1G_auto.json 2> $T/bench_${a}_1G_auto.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2_final/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["kernel_ms"], d["compression_ratio"])
    except Exception as ex: print(f, "failed", ex)
PY
