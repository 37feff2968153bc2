#!/bin/bash
# round 4's closing run: GPU suite, smoke, the driver's bench line, the rocprofv3 passes behind profiles/r04_*, the bench harness, the fall-back kernel families, the fuzzers
T=gpurun_out/r4_final; mkdir -p $T; export TMPDIR=/tmp
date +%s > $T/t0
timeout 900 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 200 python __graft_entry__.py smoke > $T/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $T/smoke.log
timeout 600 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"; head -c 300 $T/bench_full.json; echo
echo "elapsed $(( $(date +%s) - $(cat $T/t0) )) s"
bash probes/profile_round.sh gpurun_out/prof4 > $T/profile_round.log 2>&1; echo "profile rc=$?"; tail -3 $T/profile_round.log
echo "elapsed $(( $(date +%s) - $(cat $T/t0) )) s"
timeout 400 python benches/density.py > $T/benches_density.txt 2>&1; echo "harness rc=$?"; tail -24 $T/benches_density.txt
for v in 4 1; do timeout 200 python bench.py --variant $v --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $T/bench_variant$v.json 2> $T/bench_variant$v.err; echo "variant $v rc=$?"; done
for f in encode streams passes; do timeout 200 python tools/gpu_fuzz_$f.py > $T/fuzz_$f.log 2>&1; echo "fuzz $f rc=$?"; tail -2 $T/fuzz_$f.log; done
echo "elapsed $(( $(date +%s) - $(cat $T/t0) )) s"
python - <<PY
import json, glob
for f in sorted(glob.glob("$T/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["kernel_ms"], d["compression_ratio"], d["roofline"]["frac"], d["whole_path_hbm_frac"])
    except Exception as ex: print(f, "failed", ex)
PY
