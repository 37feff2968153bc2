#!/bin/bash
# rocprofv3 passes behind profiles/rNN_*: kernel-trace stats, then FETCH_SIZE and WRITE_SIZE in separate counter runs (never combined
# with other trace domains), a counter calibration on known byte counts, and the LDS counters of the codec kernels.
# Usage on the GPU box (from the repo root):  bash probes/profile_round.sh gpurun_out/prof
OUT=${1:-gpurun_out/prof}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$OUT
cd $R
B="python bench.py --no-cpu --no-sweep --no-extra"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats -- $B --steps 5 --warmup 1 > $R/$OUT/bench_stats.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/fetch -- $B --settle-ms 0 --steps 2 --warmup 1 > $R/$OUT/bench_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/write -- $B --settle-ms 0 --steps 2 --warmup 1 > $R/$OUT/bench_write.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/calib_fetch -- ./probes/fetch_calib > $R/$OUT/calib_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$OUT/calib_write -- ./probes/fetch_calib > $R/$OUT/calib_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/lds -- $B --settle-ms 0 --steps 2 --warmup 1 > $R/$OUT/bench_lds.log 2>&1
# round 4: what the waves and the memory pipeline of a CU do meanwhile — texture addresser / L1 busy, instruction mix, wait states (three passes: 8 SQ slots each)
: > $R/$OUT/ta_sq.txt
for set in "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_SMEM"; do
  n=$(echo $set | cut -c1-12 | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/$OUT/pmc_$n -- $B --settle-ms 0 --steps 2 --warmup 1 > $R/$OUT/pmc_$n.log 2>&1
  echo "rocprofv3 --pmc $set --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu   (per-launch averages, summed over the chip)" >> $R/$OUT/ta_sq.txt
  python - "$R/$OUT/pmc_$n" >> $R/$OUT/ta_sq.txt <<'PY'
import csv, glob, sys, collections, re
for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        m = re.search(r"(chameleon_encode_rot|chameleon_decode_rot)", row["Kernel_Name"])
        if m: acc[m.group(1)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        print("  " + k + ": " + ", ".join(f"{c} {sum(v) / len(v):.4g}" for c, v in sorted(cs.items())))
PY
done
timeout 120 ./probes/vmem_width > $R/$OUT/vmem_width.txt 2>/dev/null
# (the cycle accounting is compiled into the debug build only: tools/gpu_phase_prof.py loads it)
: > $R/$OUT/phase_profile.txt
for k in text random mixed; do DENSITY_HIP_PROF=1 timeout 300 python tools/gpu_phase_prof.py $k 2>&1 | grep -v amdgpu.ids >> $R/$OUT/phase_profile.txt; done
timeout 600 python bench.py --steps 10 --warmup 3 --no-extra > $R/$OUT/bench.json 2>/dev/null
# the packed container (with the stitch pass) and Cheetah at its automatic chunk, kernel by kernel
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_packed -- $B --packed --steps 5 --warmup 1 > $R/$OUT/bench_stats_packed.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_slotted -- $B --slotted --steps 5 --warmup 1 > $R/$OUT/bench_stats_slotted.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_cheetah -- python bench.py --algo cheetah --data prose --size 100000000 --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $R/$OUT/bench_stats_cheetah.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/stats_lion -- python bench.py --algo lion --data prose --size 100000000 --steps 5 --warmup 2 --no-cpu --no-sweep --no-extra > $R/$OUT/bench_stats_lion.log 2>&1
find $R/$OUT -name "*_kernel_stats.csv" -o -name "*_counter_collection.csv"
