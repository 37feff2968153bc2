#!/bin/bash
# everything the round's documents quote, in one call: GPU suite, smoke, the default bench line (what the driver runs), Cheetah / Lion lines,
# the bench harness, stream rates, probes;  usage: tools/gpu_round_end.sh <tag>
T=gpurun_out/${1:-r3_final}; mkdir -p $T; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 300 python __graft_entry__.py smoke > $T/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $T/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"; head -c 400 $T/bench_full.json; echo
for a in cheetah lion; do
  timeout 900 python bench.py --algo $a --data prose --size 100000000 --chunk 1048576 --steps 3 --warmup 1 --no-sweep --no-extra > $T/bench_${a}_1M.json 2> $T/bench_${a}_1M.err; echo "$a 1M rc=$?"
  timeout 900 python bench.py --algo $a --data prose --size 100000000 --steps 3 --warmup 1 --no-sweep --no-extra > $T/bench_${a}_auto.json 2> $T/bench_${a}_auto.err; echo "$a auto rc=$?"
  timeout 600 python bench.py --algo $a --steps 2 --warmup 1 --no-cpu --no-sweep --no-extra > $T/bench_${a}_1G_auto.json 2> $T/bench_${a}_1G_auto.err; echo "$a 1 GiB auto rc=$?"
done
timeout 900 python benches/density.py > $T/benches_density.txt 2>&1; echo "harness rc=$?"; tail -30 $T/benches_density.txt
for p in issue_rate_all lds_chase; do timeout 120 ./probes/$p > $T/probe_$p.log 2>&1; done
for k in rep random; do timeout 200 python tools/gpu_stream_rate.py 1024 $k 2>&1 | tail -2; done > $T/stream_rate.txt; cat $T/stream_rate.txt
timeout 300 python tools/gpu_stream_rate_cl.py 16 prose > $T/stream_rate_cheetah_lion.txt 2>&1; tail -6 $T/stream_rate_cheetah_lion.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$T/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["kernel_ms"], d["compression_ratio"], d["roofline"]["frac"])
    except Exception as ex: print(f, "failed", ex)
PY
