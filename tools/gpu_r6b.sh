#!/bin/bash
# round 6, second GPU call: tests again (no -x), the rotor hop probe with EXEC-masked exchanges, Lion by streams in flight, Lion's instruction counters
T=gpurun_out/r6b; mkdir -p $T; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_paged.py tests/test_gpu_decode_passes.py tests/test_gpu_cheetah_lion.py tests/test_gpu_shipped_configs.py -q > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $T/pytest.log
(cd probes && timeout 120 ./rotor_hop) > $T/probe_rotor_hop.log 2>&1; echo "probe rc=$?"; cat $T/probe_rotor_hop.log
timeout 600 python tools/gpu_lion_slots.py 2>&1 | grep -v amdgpu.ids > $T/lion_slots.txt; cat $T/lion_slots.txt
R=$(pwd); cd /tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_SMEM" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  d=$R/$T/pmc_$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -- python $R/bench.py --algo lion --data prose --size 100000000 --settle-ms 0 --steps 2 --warmup 1 --no-cpu --no-sweep --no-extra > $d.log 2>&1
  echo "pmc rc=$?"
  python - <<PY
import csv, glob, collections
for f in glob.glob("$d/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, v in acc.items():
        if "lion" in k or "fill" in k:
            print(k, {c: round(x / max(1, cnt[(k, c)]), 0) for c, x in v.items()}, "launches", max(cnt[(k, c)] for c in v))
PY
done
