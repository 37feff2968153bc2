#!/bin/bash
# round 4's closing run: GPU suite, smoke, the driver's bench line, the rocprofv3 passes behind profiles/r04_*, the fall-back kernel families, the bench harness
T=gpurun_out/r4_final; mkdir -p $T; export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -x -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $T/pytest.log
timeout 300 python __graft_entry__.py smoke > $T/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $T/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > $T/bench_full.json 2> $T/bench_full.err; echo "bench rc=$?"; head -c 300 $T/bench_full.json; echo
bash probes/profile_round.sh gpurun_out/prof4 > $T/profile_round.log 2>&1; echo "profile rc=$?"; tail -3 $T/profile_round.log
for v in 4 1; do timeout 400 python bench.py --variant $v --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $T/bench_variant$v.json 2> $T/bench_variant$v.err; echo "variant $v rc=$?"; done
timeout 900 python benches/density.py > $T/benches_density.txt 2>&1; echo "harness rc=$?"; tail -24 $T/benches_density.txt
python - <<PY
import json, glob
for f in sorted(glob.glob("$T/bench_*.json")):
    try:
        d = json.load(open(f)); print(f.split("/")[-1], d["value"], d["kernel_ms"], d["compression_ratio"], d["roofline"]["frac"], d["whole_path_hbm_frac"])
    except Exception as ex: print(f, "failed", ex)
PY
