#!/bin/bash
# the bench lines of tools/gpu_round_end.sh alone (after a bench.py fix)
T=gpurun_out/r2_final; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"; head -c 300 $T/bench_full.json; echo
for a in cheetah lion; do
  timeout 600 python bench.py --algo $a --steps 2 --warmup 1 --no-cpu --no-sweep > $T/bench_${a}_1G_auto.json 2> $T/bench_${a}_1G_auto.err; echo "$a 1 GiB auto rc=$?"
done
timeout 900 python bench.py --algo chameleon --data prose --size 100000000 --steps 10 --warmup 3 --no-cpu --no-sweep > $T/bench_chameleon_prose100M.json 2>/dev/null
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2_final/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["kernel_ms"], d["compression_ratio"], d["roofline"]["frac"])
    except Exception as ex: print(f, "failed", ex)
PY
