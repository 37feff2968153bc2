#!/bin/bash
# round 4, second GPU call: same-box A/B (round-3 library, old salt, encoder diagnostics), decoder geometries x naps, PMC counters of the codec kernels
T=gpurun_out/r4b; mkdir -p $T; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "rotor and not full_size and not beyond_2 and not long_stream" > $T/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $T/pytest.log
DENSITY_HIP_TUNE=64 timeout 600 python -m pytest tests/test_gpu_chameleon.py -m gpu -x -q -k "rotor and not full_size and not beyond_2 and not long_stream" > $T/pytest64.log 2>&1; echo "pytest tune 64 rc=$?"; tail -2 $T/pytest64.log
echo "== default geometry: tree vs r03 vs old salt vs encoder diagnostics (A no item stores, B no emit, C no loads)"
timeout 600 python tools/gpu_variants.py 20 r03 saltold encA encB encC 2>&1 | grep -v amdgpu.ids | tee $T/variants_t0.txt
for t in 64 128; do for nap in 5,3 4,2 6,4 3,1; do
  echo "== tune $t naps $nap"; DENSITY_HIP_TUNE=$t DENSITY_HIP_NAP=$nap timeout 300 python tools/gpu_variants.py 20 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $T/naps.txt
done; done
for nap in 4,2 6,4 3,1; do echo "== tune 0 naps $nap"; DENSITY_HIP_NAP=$nap timeout 300 python tools/gpu_variants.py 20 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $T/naps.txt; done
for t in 0 64; do
  DENSITY_HIP_TUNE=$t DENSITY_HIP_PROF=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-cpu --no-sweep --no-extra > $T/prof_t$t.json 2> $T/prof_t$t.err
  echo "== prof tune $t"; grep "density_hip prof" $T/prof_t$t.err | grep -v "  w[2-9] \|  w1[0-5] " | tail -12
done
cd /tmp
rocprofv3 -L > $OLDPWD/$T/counters.txt 2>&1
B="python bench.py --no-cpu --no-sweep --no-extra --settle-ms 0 --steps 2 --warmup 1"
cd $OLDPWD
for set in "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_INSTS_SMEM"; do
  n=$(echo $set | cut -c1-12 | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OLDPWD/$T/pmc_$n -- bash -c "cd $OLDPWD && $B" > $OLDPWD/$T/pmc_$n.log 2>&1)
  python - "$T/pmc_$n" <<'PY'
import csv, glob, sys, collections, re
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        m = re.search(r"(chameleon_encode_rot|chameleon_decode_rot)", row["Kernel_Name"])
        if m: acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
PY
done
